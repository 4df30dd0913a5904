"""tfhe-rs 0.5 wire format (serde + bincode 1.x) through the C ABI's fhe_wire_* entry points: thin ctypes
marshalling only (include/fhestr.h, csrc/wire_format.cpp)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import FheError, Params, _check, _ptr, _u64, lib


class _Meta(C.Structure):
    _fields_ = [("degree", C.c_uint64), ("noise_level", C.c_uint64), ("message_modulus", C.c_uint64),
                ("carry_modulus", C.c_uint64), ("pbs_order", C.c_uint32)]


@dataclass
class ShortintMeta:
    """shortint::Ciphertext's metadata fields (shortint/ciphertext/mod.rs:261-270)."""
    degree: int
    noise_level: int = 1            # NoiseLevel::NOMINAL
    message_modulus: int = 4
    carry_modulus: int = 4
    pbs_order: int = 0              # PBSOrder::KeyswitchBootstrap


def _sigs():
    L = lib()
    if getattr(L, "_wire_ready", False):
        return L
    vp, sz = C.c_void_p, C.c_size_t
    PP = C.POINTER(type(Params(1, 1, 1, 1, 1, 1, 1, 1, 1, 0.0, 0.0).c()))
    szp = C.POINTER(sz)
    for name, args in (
            ("fhe_wire_write_lwe_ciphertext", [vp, sz, vp, sz, szp]),
            ("fhe_wire_read_lwe_ciphertext", [vp, sz, vp, sz, szp, szp]),
            ("fhe_wire_write_keyswitch_key", [PP, vp, vp, sz, szp]),
            ("fhe_wire_read_keyswitch_key", [PP, vp, sz, vp, szp]),
            ("fhe_wire_write_bootstrap_key", [PP, vp, vp, sz, szp]),
            ("fhe_wire_read_bootstrap_key", [PP, vp, sz, vp, szp]),
            ("fhe_aes128_encrypt_block", [vp, vp, vp]),
            ("fhe_seeded_mask_words", [vp, vp, sz]),
            ("fhe_seeded_decompress_keyswitch_key", [PP, vp, vp, vp]),
            ("fhe_seeded_decompress_bootstrap_key", [PP, vp, vp, vp]),
            ("fhe_seeded_split_keyswitch_key", [PP, vp, vp]),
            ("fhe_seeded_split_bootstrap_key", [PP, vp, vp]),
            ("fhe_wire_write_seeded_keyswitch_key", [PP, vp, vp, vp, sz, szp]),
            ("fhe_wire_read_seeded_keyswitch_key", [PP, vp, sz, vp, vp, szp]),
            ("fhe_wire_write_seeded_bootstrap_key", [PP, vp, vp, vp, sz, szp]),
            ("fhe_wire_read_seeded_bootstrap_key", [PP, vp, sz, vp, vp, szp]),
            ("fhe_wire_write_compressed_server_key", [PP, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, sz, szp]),
            ("fhe_wire_read_compressed_server_key", [PP, vp, sz, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), szp]),
            ("fhe_seeded_decompress_lwe_batch", [C.c_uint32, vp, vp, C.c_uint32, vp]),
            ("fhe_wire_write_compressed_ciphertext", [C.c_uint64, sz, vp, C.POINTER(_Meta), vp, sz, szp]),
            ("fhe_wire_read_compressed_ciphertext", [vp, sz, C.POINTER(C.c_uint64), szp, vp, C.POINTER(_Meta), szp]),
            ("fhe_wire_write_radix_ciphertext", [vp, sz, vp, sz, vp, sz, szp]),
            ("fhe_wire_read_radix_ciphertext", [vp, sz, vp, sz, sz, vp, szp, szp]),
            ("fhe_wire_write_compressed_radix_ciphertext", [vp, vp, sz, vp, sz, vp, sz, szp]),
            ("fhe_wire_read_compressed_radix_ciphertext", [vp, sz, vp, vp, szp, sz, vp, szp, szp]),
            ("fhe_wire_write_multi_bit_bootstrap_key", [PP, vp, vp, sz, szp]),
            ("fhe_wire_read_multi_bit_bootstrap_key", [PP, vp, sz, vp, szp]),
            ("fhe_wire_write_shortint_ciphertext", [vp, sz, C.POINTER(_Meta), C.c_int, vp, sz, szp]),
            ("fhe_wire_read_shortint_ciphertext", [vp, sz, C.c_int, C.c_uint64, vp, sz, szp, C.POINTER(_Meta), szp])):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    L._wire_ready = True
    return L


def _write(call) -> bytes:
    n = C.c_size_t()
    _check(call(None, 0, C.byref(n)))
    buf = (C.c_uint8 * n.value)()
    _check(call(buf, n.value, C.byref(n)))
    return bytes(buf)


def _in(data: bytes):
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")


def write_lwe_ciphertext(ct) -> bytes:
    ct = _u64(ct)
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_lwe_ciphertext(_ptr(ct), ct.size, out, cap, n))


def read_lwe_ciphertext(data: bytes, max_words: int = 1 << 20):
    """-> (ciphertext words, bytes consumed)"""
    ct = np.zeros(max_words, dtype=np.uint64)
    size, used = C.c_size_t(), C.c_size_t()
    _check(_sigs().fhe_wire_read_lwe_ciphertext(_in(data), len(data), _ptr(ct), ct.size, C.byref(size), C.byref(used)))
    return ct[:size.value].copy(), used.value


def write_keyswitch_key(params: Params, ksk) -> bytes:
    ksk = _u64(ksk)
    if ksk.size != params.ksk_len:
        raise FheError("key size mismatch")
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_keyswitch_key(C.byref(params.c()), _ptr(ksk), out, cap, n))


def read_keyswitch_key(params: Params, data: bytes) -> np.ndarray:
    ksk = np.zeros(params.ksk_len, dtype=np.uint64)
    used = C.c_size_t()
    _check(_sigs().fhe_wire_read_keyswitch_key(C.byref(params.c()), _in(data), len(data), _ptr(ksk), C.byref(used)))
    return ksk


def write_bootstrap_key(params: Params, bsk) -> bytes:
    bsk = _u64(bsk)
    if bsk.size != params.bsk_len:
        raise FheError("key size mismatch")
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_bootstrap_key(C.byref(params.c()), _ptr(bsk), out, cap, n))


def read_bootstrap_key(params: Params, data: bytes) -> np.ndarray:
    bsk = np.zeros(params.bsk_len, dtype=np.uint64)
    used = C.c_size_t()
    _check(_sigs().fhe_wire_read_bootstrap_key(C.byref(params.c()), _in(data), len(data), _ptr(bsk), C.byref(used)))
    return bsk


def write_shortint_ciphertext(ct, meta: ShortintMeta, safe: bool = False) -> bytes:
    ct = _u64(ct)
    m = _Meta(meta.degree, meta.noise_level, meta.message_modulus, meta.carry_modulus, meta.pbs_order)
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_shortint_ciphertext(_ptr(ct), ct.size, C.byref(m), int(safe),
                                                                                 out, cap, n))


def read_shortint_ciphertext(data: bytes, safe: bool = False, size_limit: int = 0, max_words: int = 1 << 20):
    """-> (ciphertext words, ShortintMeta, bytes consumed)"""
    ct = np.zeros(max_words, dtype=np.uint64)
    size, used, m = C.c_size_t(), C.c_size_t(), _Meta()
    _check(_sigs().fhe_wire_read_shortint_ciphertext(_in(data), len(data), int(safe), size_limit, _ptr(ct), ct.size,
                                                     C.byref(size), C.byref(m), C.byref(used)))
    return ct[:size.value].copy(), ShortintMeta(m.degree, m.noise_level, m.message_modulus, m.carry_modulus, m.pbs_order), used.value


# ---- seeded ("compressed") server keys: shortint/server_key/compressed.rs; csrc/seeded_keys.cpp -------------------

def _seed16(seed) -> "C.Array":
    """128-bit compression seed: 16 bytes, or an int (the reference's Seed(u128), native = little endian)."""
    b = seed.to_bytes(16, "little") if isinstance(seed, int) else bytes(seed)
    if len(b) != 16:
        raise FheError("compression seed must be 16 bytes")
    return (C.c_uint8 * 16).from_buffer_copy(b)


def ksk_bodies_len(params: Params) -> int:
    return params.k * params.N * params.ks_level


def bsk_bodies_len(params: Params) -> int:
    return params.n_ggsw * params.pbs_level * (params.k + 1) * params.N


def aes128_encrypt_block(key: bytes, block: bytes) -> bytes:
    out = (C.c_uint8 * 16)()
    _check(_sigs().fhe_aes128_encrypt_block(_in(key), _in(block), out))
    return bytes(out)


def seeded_mask_words(seed, count: int) -> np.ndarray:
    """The first `count` u64 a MaskRandomGenerator::new(Seed(seed)) draws."""
    out = np.zeros(count, dtype=np.uint64)
    _check(_sigs().fhe_seeded_mask_words(_seed16(seed), _ptr(out), count))
    return out


def decompress_keyswitch_key(params: Params, seed, bodies) -> np.ndarray:
    bodies = _u64(bodies)
    if bodies.size != ksk_bodies_len(params):
        raise FheError("seeded keyswitch key: body count mismatch")
    ksk = np.zeros(params.ksk_len, dtype=np.uint64)
    _check(_sigs().fhe_seeded_decompress_keyswitch_key(C.byref(params.c()), _seed16(seed), _ptr(bodies), _ptr(ksk)))
    return ksk


def decompress_bootstrap_key(params: Params, seed, bodies) -> np.ndarray:
    bodies = _u64(bodies)
    if bodies.size != bsk_bodies_len(params):
        raise FheError("seeded bootstrap key: body count mismatch")
    bsk = np.zeros(params.bsk_len, dtype=np.uint64)
    _check(_sigs().fhe_seeded_decompress_bootstrap_key(C.byref(params.c()), _seed16(seed), _ptr(bodies), _ptr(bsk)))
    return bsk


def split_keyswitch_key(params: Params, ksk) -> np.ndarray:
    ksk = _u64(ksk)
    bodies = np.zeros(ksk_bodies_len(params), dtype=np.uint64)
    _check(_sigs().fhe_seeded_split_keyswitch_key(C.byref(params.c()), _ptr(ksk), _ptr(bodies)))
    return bodies


def split_bootstrap_key(params: Params, bsk) -> np.ndarray:
    bsk = _u64(bsk)
    bodies = np.zeros(bsk_bodies_len(params), dtype=np.uint64)
    _check(_sigs().fhe_seeded_split_bootstrap_key(C.byref(params.c()), _ptr(bsk), _ptr(bodies)))
    return bodies


def write_seeded_keyswitch_key(params: Params, seed, bodies) -> bytes:
    bodies, sd = _u64(bodies), _seed16(seed)
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_seeded_keyswitch_key(C.byref(params.c()), sd, _ptr(bodies), out, cap, n))


def read_seeded_keyswitch_key(params: Params, data: bytes):
    """-> (seed bytes, bodies)"""
    bodies, sd, used = np.zeros(ksk_bodies_len(params), dtype=np.uint64), (C.c_uint8 * 16)(), C.c_size_t()
    _check(_sigs().fhe_wire_read_seeded_keyswitch_key(C.byref(params.c()), _in(data), len(data), sd, _ptr(bodies), C.byref(used)))
    return bytes(sd), bodies


def write_seeded_bootstrap_key(params: Params, seed, bodies) -> bytes:
    bodies, sd = _u64(bodies), _seed16(seed)
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_seeded_bootstrap_key(C.byref(params.c()), sd, _ptr(bodies), out, cap, n))


def read_seeded_bootstrap_key(params: Params, data: bytes):
    """-> (seed bytes, bodies); the multi-bit container when the parameter set has a grouping factor"""
    bodies, sd, used = np.zeros(bsk_bodies_len(params), dtype=np.uint64), (C.c_uint8 * 16)(), C.c_size_t()
    _check(_sigs().fhe_wire_read_seeded_bootstrap_key(C.byref(params.c()), _in(data), len(data), sd, _ptr(bodies), C.byref(used)))
    return bytes(sd), bodies


def write_multi_bit_bootstrap_key(params: Params, bsk) -> bytes:
    bsk = _u64(bsk)
    if bsk.size != params.bsk_len:
        raise FheError("key size mismatch")
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_multi_bit_bootstrap_key(C.byref(params.c()), _ptr(bsk), out, cap, n))


def read_multi_bit_bootstrap_key(params: Params, data: bytes) -> np.ndarray:
    bsk, used = np.zeros(params.bsk_len, dtype=np.uint64), C.c_size_t()
    _check(_sigs().fhe_wire_read_multi_bit_bootstrap_key(C.byref(params.c()), _in(data), len(data), _ptr(bsk), C.byref(used)))
    return bsk


def write_compressed_server_key(params: Params, ksk_seed, ksk_bodies, bsk_seed, bsk_bodies, max_degree: int | None = None,
                                pbs_order: int = 0) -> bytes:
    """shortint CompressedServerKey (shortint/server_key/compressed.rs:44-55)."""
    kb, bb, ks, bs = _u64(ksk_bodies), _u64(bsk_bodies), _seed16(ksk_seed), _seed16(bsk_seed)
    deg = params.msg_mod * params.carry_mod - 1 if max_degree is None else max_degree
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_compressed_server_key(
        C.byref(params.c()), ks, _ptr(kb), bs, _ptr(bb), deg, pbs_order, out, cap, n))


def read_compressed_server_key(params: Params, data: bytes) -> dict:
    """-> {ksk_seed, ksk_bodies, bsk_seed, bsk_bodies, max_degree, pbs_order, consumed}: feed the first four to
    Engine.load_seeded_keys."""
    kb = np.zeros(ksk_bodies_len(params), dtype=np.uint64)
    bb = np.zeros(bsk_bodies_len(params), dtype=np.uint64)
    ks, bs = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
    deg, order, used = C.c_uint64(), C.c_uint32(), C.c_size_t()
    _check(_sigs().fhe_wire_read_compressed_server_key(C.byref(params.c()), _in(data), len(data), ks, _ptr(kb), bs, _ptr(bb),
                                                       C.byref(deg), C.byref(order), C.byref(used)))
    return {"ksk_seed": bytes(ks), "ksk_bodies": kb, "bsk_seed": bytes(bs), "bsk_bodies": bb, "max_degree": deg.value,
            "pbs_order": order.value, "consumed": used.value}


def decompress_lwe_batch(lwe_dim: int, seeds, bodies) -> np.ndarray:
    """Host-side expansion of seeded LWE ciphertexts (one seed each): (count, lwe_dim + 1)."""
    seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8).reshape(-1, 16))
    bodies = _u64(bodies)
    out = np.zeros((bodies.size, lwe_dim + 1), dtype=np.uint64)
    _check(_sigs().fhe_seeded_decompress_lwe_batch(lwe_dim, seeds.ctypes.data_as(C.c_void_p), _ptr(bodies), bodies.size, _ptr(out)))
    return out


def write_compressed_ciphertext(body: int, lwe_size: int, seed, meta: ShortintMeta) -> bytes:
    """shortint CompressedCiphertext (shortint/ciphertext/mod.rs:471-478)."""
    m, sd = _Meta(meta.degree, meta.noise_level, meta.message_modulus, meta.carry_modulus, meta.pbs_order), _seed16(seed)
    return _write(lambda out, cap, n: _sigs().fhe_wire_write_compressed_ciphertext(int(body), lwe_size, sd, C.byref(m), out, cap, n))


def read_compressed_ciphertext(data: bytes):
    """-> (body, lwe_size, seed bytes, ShortintMeta, bytes consumed)"""
    body, size, used, sd, m = C.c_uint64(), C.c_size_t(), C.c_size_t(), (C.c_uint8 * 16)(), _Meta()
    _check(_sigs().fhe_wire_read_compressed_ciphertext(_in(data), len(data), C.byref(body), C.byref(size), sd, C.byref(m), C.byref(used)))
    return body.value, size.value, bytes(sd), ShortintMeta(m.degree, m.noise_level, m.message_modulus, m.carry_modulus, m.pbs_order), used.value


# ---- integer RadixCiphertext / CompressedRadixCiphertext (integer/ciphertext/mod.rs:18-21,30,45) --------------------

def _metas(metas, n):
    arr = (_Meta * max(1, n))()
    for i, m in enumerate(metas):
        arr[i] = _Meta(m.degree, m.noise_level, m.message_modulus, m.carry_modulus, m.pbs_order)
    return arr


def _metas_out(arr, n):
    return [ShortintMeta(m.degree, m.noise_level, m.message_modulus, m.carry_modulus, m.pbs_order) for m in arr[:n]]


def write_radix_ciphertext(cts, metas) -> bytes:
    """cts: (n_blocks, lwe_size), least significant block first; one ShortintMeta per block."""
    cts = np.ascontiguousarray(np.asarray(cts, dtype=np.uint64))
    n, size = cts.shape
    m = _metas(metas, n)
    return _write(lambda out, cap, w: _sigs().fhe_wire_write_radix_ciphertext(_ptr(cts), size, m, n, out, cap, w))


def read_radix_ciphertext(data: bytes, lwe_size: int, max_blocks: int = 4096):
    """-> (cts (n_blocks, lwe_size), [ShortintMeta], bytes consumed)"""
    cts = np.zeros((max_blocks, lwe_size), dtype=np.uint64)
    m, n, used = (_Meta * max_blocks)(), C.c_size_t(), C.c_size_t()
    _check(_sigs().fhe_wire_read_radix_ciphertext(_in(data), len(data), _ptr(cts), lwe_size, max_blocks, m, C.byref(n), C.byref(used)))
    return cts[:n.value].copy(), _metas_out(m, n.value), used.value


def write_compressed_radix_ciphertext(bodies, seeds, lwe_size: int, metas) -> bytes:
    bodies = _u64(bodies)
    seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint8).reshape(-1, 16))
    n = bodies.size
    m = _metas(metas, n)
    return _write(lambda out, cap, w: _sigs().fhe_wire_write_compressed_radix_ciphertext(
        _ptr(bodies), seeds.ctypes.data_as(C.c_void_p), lwe_size, m, n, out, cap, w))


def read_compressed_radix_ciphertext(data: bytes, max_blocks: int = 4096):
    """-> (bodies, seeds (n, 16), lwe_size, [ShortintMeta], bytes consumed): feed bodies / seeds to Engine.expand_seeded_lwe"""
    bodies, seeds = np.zeros(max_blocks, dtype=np.uint64), np.zeros((max_blocks, 16), dtype=np.uint8)
    m, n, used, size = (_Meta * max_blocks)(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    _check(_sigs().fhe_wire_read_compressed_radix_ciphertext(_in(data), len(data), _ptr(bodies), seeds.ctypes.data_as(C.c_void_p),
                                                             C.byref(size), max_blocks, m, C.byref(n), C.byref(used)))
    return bodies[:n.value].copy(), seeds[:n.value].copy(), size.value, _metas_out(m, n.value), used.value
