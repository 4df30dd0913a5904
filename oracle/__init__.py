"""ctypes front-end of the CPU ORACLE (oracle/tfhe_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (fhe-string-bounty_amd/) never imports this module.

The C file restates the tfhe-rs 0.5.0 hot path of the reference (citations inside); this wrapper
only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False, arch: str | None = None) -> str:
    """Compile liboracle.so with gcc (no-op when it is already there)."""
    if force or not os.path.exists(_LIB_PATH):
        cmd = ["make", "-C", _HERE, "-B" if force else "-s"]
        if arch:
            cmd.append(f"ARCH={arch}")
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)   # keep the caller's stdout clean (bench.py prints one JSON line)
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("k", C.c_uint32), ("N", C.c_uint32),
        ("pbs_base_log", C.c_uint32), ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32), ("ks_level", C.c_uint32),
        ("msg_mod", C.c_uint32), ("carry_mod", C.c_uint32),
        ("lwe_std", C.c_double), ("glwe_std", C.c_double),
    ]


@dataclass(frozen=True)
class Params:
    """shortint ClassicPBSParameters (reference: shortint/parameters/mod.rs:61-76)."""
    n: int
    k: int
    N: int
    pbs_base_log: int
    pbs_level: int
    ks_base_log: int
    ks_level: int
    msg_mod: int
    carry_mod: int
    lwe_std: float
    glwe_std: float
    name: str = ""

    @property
    def big_dim(self) -> int:
        return self.k * self.N

    @property
    def big_size(self) -> int:
        return self.k * self.N + 1

    @property
    def small_size(self) -> int:
        return self.n + 1

    @property
    def glwe_len(self) -> int:
        return (self.k + 1) * self.N

    @property
    def delta(self) -> int:
        return (1 << 63) // (self.msg_mod * self.carry_mod)

    def c(self) -> _Params:
        return _Params(self.n, self.k, self.N, self.pbs_base_log, self.pbs_level, self.ks_base_log,
                       self.ks_level, self.msg_mod, self.carry_mod, self.lwe_std, self.glwe_std)


# reference: shortint/parameters/mod.rs:703-717, :658-672, :613-627, :1063-1077
PARAM_MESSAGE_2_CARRY_2_KS_PBS = Params(742, 1, 2048, 23, 1, 3, 5, 4, 4,
                                        0.000007069849454709433, 0.00000000000000029403601535432533,
                                        "PARAM_MESSAGE_2_CARRY_2_KS_PBS")
PARAM_MESSAGE_2_CARRY_1_KS_PBS = Params(742, 2, 1024, 23, 1, 4, 3, 4, 2,
                                        0.000007069849454709433, 0.00000000000000029403601535432533,
                                        "PARAM_MESSAGE_2_CARRY_1_KS_PBS")
PARAM_MESSAGE_1_CARRY_1_KS_PBS = Params(684, 3, 512, 18, 1, 4, 3, 2, 2,
                                        0.00002043784477291318, 0.0000000000034525330484572114,
                                        "PARAM_MESSAGE_1_CARRY_1_KS_PBS")
PARAM_MESSAGE_4_CARRY_4_KS_PBS = Params(996, 1, 32768, 15, 2, 3, 7, 16, 16,
                                        0.00000006767666038309478, 0.0000000000000000002168404344971009,
                                        "PARAM_MESSAGE_4_CARRY_4_KS_PBS")
# Tiny sets for fast tests (NOT secure; noise small enough that decryption is always right).
TOY_K1 = Params(16, 1, 256, 10, 2, 4, 4, 4, 4, 1e-12, 1e-15, "TOY_K1_N256_L2")
TOY_K2 = Params(12, 2, 128, 12, 1, 3, 5, 2, 2, 1e-12, 1e-15, "TOY_K2_N128_L1")
# large-polynomial shapes (PARAM_MESSAGE_3_CARRY_3 / PARAM_MESSAGE_4_CARRY_4 geometry, tiny n)
TOY_N8192 = Params(8, 1, 8192, 15, 2, 3, 6, 8, 8, 1e-13, 1e-17, "TOY_N8192_L2")
TOY_N32768 = Params(4, 1, 32768, 15, 2, 3, 7, 16, 16, 1e-13, 1e-17, "TOY_N32768_L2")
# remaining (N, k, level, base) shapes of the reference's *_KS_PBS parameter table, tiny n
TOY_SHAPES = [
    Params(6, 5, 256, 15, 1, 5, 2, 2, 1, 1e-12, 1e-15, "TOY_N256_K5"),        # 1_CARRY_0
    Params(6, 2, 512, 8, 2, 4, 3, 4, 1, 1e-12, 1e-15, "TOY_N512_K2_L2"),      # 2_CARRY_0
    Params(6, 3, 512, 18, 1, 4, 3, 2, 2, 1e-12, 1e-15, "TOY_N512_K3"),        # 1_CARRY_1
    Params(6, 2, 1024, 23, 1, 4, 3, 4, 2, 1e-12, 1e-16, "TOY_N1024_K2"),      # 2_CARRY_1
    Params(6, 1, 4096, 22, 1, 3, 6, 4, 8, 1e-13, 1e-17, "TOY_N4096_L1"),      # 2_CARRY_3
    Params(6, 1, 4096, 15, 2, 3, 6, 2, 16, 1e-13, 1e-17, "TOY_N4096_L2"),     # 1_CARRY_4
    Params(6, 1, 8192, 22, 1, 3, 6, 32, 2, 1e-13, 1e-17, "TOY_N8192_L1"),     # 5_CARRY_1
    Params(4, 1, 16384, 15, 2, 3, 6, 8, 16, 1e-13, 1e-17, "TOY_N16384_L2"),   # 3_CARRY_4
    Params(4, 1, 16384, 11, 3, 3, 6, 2, 64, 1e-13, 1e-17, "TOY_N16384_L3"),   # 1_CARRY_6
    Params(3, 1, 32768, 11, 3, 3, 7, 8, 32, 1e-13, 1e-17, "TOY_N32768_L3"),   # 3_CARRY_5
]

# multi-bit PBS parameter sets (shortint/parameters/multi_bit.rs:115-135): same fields + grouping factor
PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS = Params(818, 1, 2048, 22, 1, 5, 3, 4, 4,
                                                          0.000002226459789930014, 0.0000000000000003152931493498455,
                                                          "PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS")
PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS = Params(888, 1, 2048, 21, 1, 7, 2, 4, 4,
                                                          0.0000006125031601933181, 0.0000000000000003152931493498455,
                                                          "PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS")
# shortint/parameters/multi_bit.rs:96-113, 154-171, 134-152, 192-209
PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_2_KS_PBS = Params(764, 3, 512, 18, 1, 6, 2, 2, 2,
                                                          0.000006025673585415336, 0.0000000000039666089171633006,
                                                          "PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_2_KS_PBS")
PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS = Params(765, 3, 512, 18, 1, 6, 2, 2, 2,
                                                          0.000005915594083804978, 0.0000000000039666089171633006,
                                                          "PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS")
PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_2_KS_PBS = Params(922, 1, 8192, 14, 2, 4, 4, 8, 8,
                                                          0.0000003272369292345697, 0.0000000000000000002168404344971009,
                                                          "PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_2_KS_PBS")
PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_3_KS_PBS = Params(972, 1, 8192, 14, 2, 6, 3, 8, 8,
                                                          0.00000013016688349592805, 0.0000000000000000002168404344971009,
                                                          "PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_3_KS_PBS")
# the other multi-bit shapes with tiny n (two-kernel path of the engine)
TOY_MULTI_BIT_N256_G3 = Params(15, 1, 256, 10, 2, 4, 4, 4, 4, 1e-12, 1e-15, "TOY_MULTI_BIT_N256_G3")
TOY_MULTI_BIT_N128_K2 = Params(12, 2, 128, 12, 1, 3, 5, 2, 2, 1e-12, 1e-15, "TOY_MULTI_BIT_N128_K2_G2")
TOY_MULTI_BIT_N512_K3_G3 = Params(9, 3, 512, 18, 1, 6, 2, 2, 2, 1e-12, 1e-15, "TOY_MULTI_BIT_N512_K3_G3")
TOY_MULTI_BIT_N8192 = Params(8, 1, 8192, 14, 2, 4, 4, 8, 8, 1e-13, 1e-17, "TOY_MULTI_BIT_N8192_G2")
TOY_MULTI_BIT_N8192_G3 = Params(9, 1, 8192, 14, 2, 6, 3, 8, 8, 1e-13, 1e-17, "TOY_MULTI_BIT_N8192_G3")
TOY_MULTI_BIT_N2048 = Params(12, 1, 2048, 22, 1, 5, 3, 4, 4, 1e-13, 1e-17, "TOY_MULTI_BIT_N2048_G2")
TOY_MULTI_BIT_N2048_G3 = Params(12, 1, 2048, 21, 1, 7, 2, 4, 4, 1e-13, 1e-17, "TOY_MULTI_BIT_N2048_G3")
TOY_MULTI_BIT_N256 = Params(16, 1, 256, 10, 2, 4, 4, 4, 4, 1e-12, 1e-15, "TOY_MULTI_BIT_N256_G2")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER(_Params)
    u64, u32, sz, dbl, vp, i32 = C.c_uint64, C.c_uint32, C.c_size_t, C.c_double, C.c_void_p, C.c_int

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("orc_closest_representable", u64, u64, u32, u32, u32)
    sig("orc_decompose", None, u64, u32, u32, u32, _u64p)
    sig("orc_modulus_switch", u64, u64, u32)
    sig("orc_monomial_div", None, _u64p, _u64p, u32, u64, u32)
    sig("orc_monomial_mul", None, _u64p, _u64p, u32, u64, u32)
    sig("orc_monomial_mul_and_subtract", None, _u64p, _u64p, u32, u64, u32)
    sig("orc_slice_sub_scalar_mul", None, _u64p, _u64p, u64, sz, u32)
    sig("orc_from_torus", u64, dbl)
    sig("orc_f64_to_i64", C.c_int64, dbl)
    sig("orc_keyswitch", None, P, _u64p, _u64p, _u64p)
    sig("orc_sample_extract", None, P, _u64p, _u64p)
    sig("orc_fft_new", vp, u32)
    sig("orc_fft_free", None, vp)
    sig("orc_fft_forward_as_integer", None, vp, _f64p, _u64p)
    sig("orc_fft_forward_as_torus", None, vp, _f64p, _u64p)
    sig("orc_fft_add_backward_as_torus", None, vp, _u64p, _f64p)
    sig("orc_bsk_to_fourier", None, P, _u64p, _f64p)
    sig("orc_add_external_product_fft", None, P, vp, _u64p, _f64p, _u64p)
    sig("orc_add_external_product_exact", None, P, _u64p, _u64p, _u64p)
    sig("orc_blind_rotate_fft", None, P, vp, _f64p, _u64p, _u64p)
    sig("orc_blind_rotate_exact", None, P, _u64p, _u64p, _u64p)
    sig("orc_pbs_fft", None, P, vp, _f64p, _u64p, _u64p, _u64p)
    sig("orc_pbs_exact", None, P, _u64p, _u64p, _u64p, _u64p)
    sig("orc_ks_pbs_batch", None, P, _u64p, vp, vp, i32, _u64p, vp, _u64p, _u64p, sz, i32)
    sig("orc_fill_accumulator", u64, P, _u64p, _u64p)
    sig("orc_trivial_pbs_body", u64, P, u64, _u64p)
    sig("orc_gen_binary_key", None, C.c_char_p, u64, _u64p, sz)
    sig("orc_lwe_encrypt", None, _u64p, sz, u64, dbl, vp, _u64p)
    sig("orc_lwe_decrypt", u64, _u64p, sz, _u64p)
    sig("orc_gen_ksk", None, P, _u64p, _u64p, C.c_char_p, _u64p)
    sig("orc_gen_bsk", None, P, _u64p, _u64p, C.c_char_p, _u64p, i32)
    sig("orc_encode", u64, P, u64)
    sig("orc_decode", u64, P, u64)
    sig("orc_multi_bit_key_bits", None, _u64p, u32, u32, _u64p)
    sig("orc_multi_bit_pbs_fft", None, P, u32, vp, _f64p, _u64p, _u64p, _u64p)
    sig("orc_multi_bit_pbs_exact", None, P, u32, _u64p, _u64p, _u64p, _u64p)
    sig("orc_rng_init", None, vp, C.c_char_p, u64)
    sig("orc_chacha20_block", None, C.c_char_p, u64, u64, vp)
    sig("orc_rng_next", u64, vp)
    _lib = L
    return L


def _a(x, dtype=np.uint64):
    return np.ascontiguousarray(x, dtype=dtype)


# --------------------------------------------------------------------- integer helpers
def closest_representable(x, base_log, level, bits=64):
    return int(lib().orc_closest_representable(int(x), base_log, level, bits))


def decompose(x, base_log, level, bits=64):
    out = np.zeros(level, dtype=np.uint64)
    lib().orc_decompose(int(x), base_log, level, bits, out)
    return out


def modulus_switch(x, log2N):
    return int(lib().orc_modulus_switch(int(x), log2N))


def monomial_div(poly, degree, bits=64):
    poly = _a(poly)
    out = np.zeros_like(poly)
    lib().orc_monomial_div(out, poly, len(poly), degree, bits)
    return out


def monomial_mul(poly, degree, bits=64):
    poly = _a(poly)
    out = np.zeros_like(poly)
    lib().orc_monomial_mul(out, poly, len(poly), degree, bits)
    return out


def monomial_mul_and_subtract(poly, degree, bits=64):
    poly = _a(poly)
    out = np.zeros_like(poly)
    lib().orc_monomial_mul_and_subtract(out, poly, len(poly), degree, bits)
    return out


def slice_sub_scalar_mul(out, inp, scalar, bits=64):
    out = _a(out).copy()
    lib().orc_slice_sub_scalar_mul(out, _a(inp), int(scalar), len(out), bits)
    return out


def from_torus(x: float) -> int:
    return int(lib().orc_from_torus(float(x)))


def f64_to_i64(x: float) -> int:
    return int(lib().orc_f64_to_i64(float(x)))


# --------------------------------------------------------------------- client side (harness)
def seed_bytes(seed) -> bytes:
    """256-bit seed (ChaCha20 key): 32 bytes, or an int for tests (little endian, zero extended)."""
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 32
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


def chacha20_block(key: bytes, counter: int, stream: int) -> np.ndarray:
    out = np.zeros(16, dtype=np.uint32)
    lib().orc_chacha20_block(seed_bytes(key), counter, stream, out.ctypes.data_as(C.c_void_p))
    return out


class Rng:
    """Sequential reader of one ChaCha20 stream of the harness generator (orc_rng)."""

    def __init__(self, seed, stream: int = 0):
        self.buf = (C.c_uint64 * 16)()          # >= sizeof(orc_rng) = 32 + 16 + 64 + 4 (+ padding)
        lib().orc_rng_init(C.addressof(self.buf), seed_bytes(seed), stream)

    def next(self) -> int:
        return int(lib().orc_rng_next(C.addressof(self.buf)))

    @property
    def ptr(self):
        return C.addressof(self.buf)


class ClientKey:
    """Secret keys of one parameter set, generated from a seed (binary keys).

    reference: shortint/engine/client_side.rs:13-56 (big LWE key == flattened GLWE key)."""

    def __init__(self, params: Params, seed: int):
        self.params = params
        self.seed = seed
        self.glwe_sk = np.zeros(params.k * params.N, dtype=np.uint64)
        self.small_sk = np.zeros(params.n, dtype=np.uint64)
        lib().orc_gen_binary_key(seed_bytes(seed), 1, self.glwe_sk, self.glwe_sk.size)
        lib().orc_gen_binary_key(seed_bytes(seed), 2, self.small_sk, self.small_sk.size)
        self.big_sk = self.glwe_sk  # client_side.rs:29
        self._rng = Rng(seed, 3)

    def encrypt_plaintext(self, pt: int, rng: Rng | None = None) -> np.ndarray:
        p = self.params
        ct = np.zeros(p.big_size, dtype=np.uint64)
        lib().orc_lwe_encrypt(self.big_sk, p.big_dim, int(pt) & (2**64 - 1), p.glwe_std,
                              (rng or self._rng).ptr, ct)
        return ct

    def encrypt(self, msg: int, rng: Rng | None = None) -> np.ndarray:
        """message + carry space encryption (client_side.rs:58-86 with m < msg*carry)."""
        return self.encrypt_plaintext(int(lib().orc_encode(C.byref(self.params.c()), int(msg))), rng)

    def encrypt_many(self, msgs, rng: Rng | None = None) -> np.ndarray:
        return np.stack([self.encrypt(int(m), rng) for m in msgs])

    def decrypt_plaintext(self, ct) -> int:
        return int(lib().orc_lwe_decrypt(self.big_sk, self.params.big_dim, _a(ct)))

    def decrypt_small_plaintext(self, ct) -> int:
        return int(lib().orc_lwe_decrypt(self.small_sk, self.params.n, _a(ct)))

    def decrypt_message_and_carry(self, ct) -> int:
        return int(lib().orc_decode(C.byref(self.params.c()), self.decrypt_plaintext(ct)))

    def decrypt(self, ct) -> int:
        return self.decrypt_message_and_carry(ct) % self.params.msg_mod

    def decrypt_many(self, cts) -> np.ndarray:
        return np.array([self.decrypt_message_and_carry(c) for c in cts], dtype=np.int64)


class ServerKey:
    """Oracle twin of shortint::ServerKey: KSK + standard BSK + Fourier BSK + the operators."""

    def __init__(self, ck: ClientKey, threads: int | None = None, fourier: bool = True):
        p = ck.params
        self.params = p
        self.threads = threads or min(8, os.cpu_count() or 1)
        pc = p.c()
        self.ksk = np.zeros(p.big_dim * p.ks_level * p.small_size, dtype=np.uint64)
        lib().orc_gen_ksk(C.byref(pc), ck.big_sk, ck.small_sk, seed_bytes(ck.seed), self.ksk)
        self.bsk = np.zeros(p.n * p.pbs_level * (p.k + 1) ** 2 * p.N, dtype=np.uint64)
        lib().orc_gen_bsk(C.byref(pc), ck.small_sk, ck.glwe_sk, seed_bytes(ck.seed), self.bsk, self.threads)
        self.fbsk = None
        if fourier:
            self.fbsk = np.zeros(self.bsk.size, dtype=np.float64)
            lib().orc_bsk_to_fourier(C.byref(pc), self.bsk, self.fbsk)

    @classmethod
    def from_keys(cls, params: Params, bsk, ksk, threads: int | None = None, fourier: bool = True):
        """Oracle operators over server keys made elsewhere (standard-domain layouts of SURVEY.md 8(a)), e.g. the keys the
        GPU engine generated and exported: the checker then runs on exactly the key material the device kernels use."""
        self = cls.__new__(cls)
        self.params = params
        self.threads = threads or min(8, os.cpu_count() or 1)
        self.ksk = _a(ksk).reshape(-1)
        self.bsk = _a(bsk).reshape(-1)
        assert self.ksk.size == params.big_dim * params.ks_level * params.small_size
        assert self.bsk.size == params.n * params.pbs_level * (params.k + 1) ** 2 * params.N
        self.fbsk = None
        if fourier:
            self.fbsk = np.zeros(self.bsk.size, dtype=np.float64)
            lib().orc_bsk_to_fourier(C.byref(params.c()), self.bsk, self.fbsk)
        return self

    # shortint/server_key/mod.rs:383-399
    def generate_lookup_table(self, f):
        p = self.params
        table = np.array([int(f(i)) for i in range(p.msg_mod * p.carry_mod)], dtype=np.uint64)
        lut = np.zeros(p.glwe_len, dtype=np.uint64)
        degree = int(lib().orc_fill_accumulator(C.byref(p.c()), table, lut))
        return lut, degree

    # shortint/server_key/bivariate_pbs.rs:71-96,125-130
    def generate_lookup_table_bivariate(self, f, factor=None):
        p = self.params
        factor = factor or p.msg_mod
        return self.generate_lookup_table(
            lambda x: f((x // factor) % p.msg_mod, (x % factor) % p.msg_mod))

    def keyswitch(self, ct_big):
        p = self.params
        out = np.zeros(p.small_size, dtype=np.uint64)
        lib().orc_keyswitch(C.byref(p.c()), self.ksk, _a(ct_big), out)
        return out

    def pbs(self, ct_small, lut, exact=False):
        p = self.params
        out = np.zeros(p.big_size, dtype=np.uint64)
        if exact:
            lib().orc_pbs_exact(C.byref(p.c()), self.bsk, _a(ct_small), _a(lut), out)
        else:
            f = lib().orc_fft_new(p.N)
            lib().orc_pbs_fft(C.byref(p.c()), f, self.fbsk, _a(ct_small), _a(lut), out)
            lib().orc_fft_free(f)
        return out

    def apply_lookup_table_batch(self, cts, luts, lut_idx=None, exact=False, threads=None):
        """KS -> PBS for every row of `cts` (shortint/server_key/mod.rs:783-857)."""
        p = self.params
        cts = _a(cts).reshape(-1, p.big_size)
        luts = _a(luts).reshape(-1, p.glwe_len)
        out = np.zeros_like(cts)
        idx = None
        if lut_idx is not None:
            idx_arr = _a(lut_idx, np.uint32)
            assert idx_arr.size == cts.shape[0] and int(idx_arr.max(initial=0)) < luts.shape[0]
            idx = idx_arr.ctypes.data_as(C.c_void_p)
        fb = self.fbsk.ctypes.data_as(C.c_void_p) if self.fbsk is not None else None
        lib().orc_ks_pbs_batch(C.byref(p.c()), self.ksk, fb, self.bsk.ctypes.data_as(C.c_void_p),
                               1 if exact else 0, cts, idx, luts, out, cts.shape[0],
                               threads or self.threads)
        return out

    def apply_lookup_table(self, ct, lut, exact=False):
        return self.apply_lookup_table_batch(np.asarray(ct)[None, :], lut, exact=exact, threads=1)[0]

    def trivial_pbs_body(self, body: int, lut) -> int:
        return int(lib().orc_trivial_pbs_body(C.byref(self.params.c()), int(body), _a(lut)))


class MultiBitServerKey:
    """Oracle twin of a shortint ServerKey holding a multi-bit bootstrapping key
    (ShortintBootstrappingKey::MultiBit, shortint/server_key/mod.rs:829-851): KSK + the n/g * 2^g
    GGSWs of lwe_multi_bit_bootstrap_key_generation.rs:87-173, standard and Fourier."""

    def __init__(self, ck: ClientKey, grouping: int = 2, threads: int | None = None):
        p = ck.params
        assert p.n % grouping == 0
        self.params, self.grouping = p, grouping
        self.threads = threads or min(8, os.cpu_count() or 1)
        self.n_ggsw = p.n // grouping * (1 << grouping)
        self.key_bits = np.zeros(self.n_ggsw, dtype=np.uint64)
        lib().orc_multi_bit_key_bits(ck.small_sk, p.n, grouping, self.key_bits)
        self.ksk = np.zeros(p.big_dim * p.ks_level * p.small_size, dtype=np.uint64)
        lib().orc_gen_ksk(C.byref(p.c()), ck.big_sk, ck.small_sk, seed_bytes(ck.seed), self.ksk)
        # the multi-bit key is a list of n_ggsw constant GGSWs: the classic generator on that list
        self._pk = dataclasses.replace(p, n=self.n_ggsw)
        self.bsk = np.zeros(self.n_ggsw * p.pbs_level * (p.k + 1) ** 2 * p.N, dtype=np.uint64)
        lib().orc_gen_bsk(C.byref(self._pk.c()), self.key_bits, ck.glwe_sk, seed_bytes(ck.seed), self.bsk, self.threads)
        self.fbsk = np.zeros(self.bsk.size, dtype=np.float64)
        lib().orc_bsk_to_fourier(C.byref(self._pk.c()), self.bsk, self.fbsk)
        self._fft = lib().orc_fft_new(p.N)

    generate_lookup_table = ServerKey.generate_lookup_table

    def keyswitch(self, ct):
        out = np.zeros(self.params.small_size, dtype=np.uint64)
        lib().orc_keyswitch(C.byref(self.params.c()), self.ksk, _a(ct), out)
        return out

    def pbs(self, ct_small, lut, exact=False):
        out = np.zeros(self.params.big_size, dtype=np.uint64)
        if exact:
            lib().orc_multi_bit_pbs_exact(C.byref(self.params.c()), self.grouping, self.bsk, _a(ct_small), _a(lut), out)
        else:
            lib().orc_multi_bit_pbs_fft(C.byref(self.params.c()), self.grouping, self._fft, self.fbsk,
                                        _a(ct_small), _a(lut), out)
        return out

    def apply_lookup_table(self, ct, lut, exact=False):
        return self.pbs(self.keyswitch(ct), lut, exact=exact)

    def apply_lookup_table_batch(self, cts, lut, exact=False):
        return np.stack([self.apply_lookup_table(c, lut, exact=exact) for c in _a(cts).reshape(-1, self.params.big_size)])


def fft_roundtrip_product(N: int, torus_poly, int_poly):
    """forward_as_torus(a) * forward_as_integer(b) -> add_backward_as_torus into zeros."""
    f = lib().orc_fft_new(N)
    fa = np.zeros(N, dtype=np.float64)
    fb = np.zeros(N, dtype=np.float64)
    lib().orc_fft_forward_as_torus(f, fa, _a(torus_poly))
    lib().orc_fft_forward_as_integer(f, fb, _a(int_poly))
    za = fa.view(np.complex128) * fb.view(np.complex128)
    out = np.zeros(N, dtype=np.uint64)
    lib().orc_fft_add_backward_as_torus(f, out, np.ascontiguousarray(za).view(np.float64))
    lib().orc_fft_free(f)
    return out


def negacyclic_schoolbook(a, b):
    """Exact negacyclic product mod (X^N+1, 2^64) with numpy wrapping arithmetic."""
    a = _a(a)
    b = _a(b)
    N = len(a)
    full = np.zeros(2 * N, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(N):
            if a[i]:
                full[i:i + N] += a[i] * b
        return full[:N] - full[N:]
