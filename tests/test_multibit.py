"""Multi-bit programmable bootstrap (SURVEY.md section 8(f) rank 4;
core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs, shortint/parameters/multi_bit.rs).

CPU: the oracle's restatement decrypts correctly on its f64 and exact-integer paths (the reference's own
doctest asserts no more: :170-178 of that file) and the product's client generates the oracle's keys.
GPU: blind_rotate_multibit_kernel vs the oracle on identical keys and inputs."""
import dataclasses

import numpy as np
import pytest

import oracle as O
from conftest import torus_distance

G = 2
TOY = O.TOY_MULTI_BIT_N2048
TOY_G3 = O.TOY_MULTI_BIT_N2048_G3
REAL = O.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS
REAL_G3 = O.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS     # shortint/parameters/multi_bit.rs:173-190
_KEYS = {}


def _g(p):
    return 3 if p.name.endswith("G3") or "GROUP_3" in p.name else 2


def _fp(p):
    import fhestr
    return fhestr.Params(p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level,
                         p.msg_mod, p.carry_mod, p.lwe_std, p.glwe_std, p.name, _g(p))


def _keys(p, seed=0x4D420001):
    if p.name not in _KEYS:
        ck = O.ClientKey(p, seed)
        _KEYS[p.name] = (ck, O.MultiBitServerKey(ck, _g(p)))
    return _KEYS[p.name]


def test_key_bits_follow_the_reference_selector_order():
    # lwe_multi_bit_bootstrap_key_generation.rs:401-427: GGSW 0 encrypts (1-s1)(1-s2), 1: (1-s1) s2,
    # 2: s1 (1-s2), 3: s1 s2
    sk = np.array([0, 0, 0, 1, 1, 0, 1, 1], dtype=np.uint64)
    bits = np.zeros(16, dtype=np.uint64)
    O.lib().orc_multi_bit_key_bits(sk, 8, 2, bits)
    assert bits.reshape(4, 4).tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]


def test_key_bits_grouping_factor_3():
    # selector bit g-1-b <-> key bit b of the group: GGSW `sel` encrypts prod_b (s_b if that bit is set else 1 - s_b)
    sk = np.array([1, 0, 1], dtype=np.uint64)
    bits = np.zeros(8, dtype=np.uint64)
    O.lib().orc_multi_bit_key_bits(sk, 3, 3, bits)
    assert bits.tolist() == [0, 0, 0, 0, 0, 1, 0, 0]          # only selector 0b101


@pytest.mark.parametrize("p", [O.TOY_MULTI_BIT_N256, TOY, TOY_G3, O.TOY_MULTI_BIT_N128_K2, O.TOY_MULTI_BIT_N512_K3_G3],
                         ids=lambda p: p.name)
def test_oracle_multi_bit_pbs_decrypts(p):
    ck, sk = _keys(p)
    M = p.msg_mod * p.carry_mod
    f = lambda x: (5 * x + 2) % M
    lut, _ = sk.generate_lookup_table(f)
    cts = ck.encrypt_many(range(M))
    out = sk.apply_lookup_table_batch(cts, lut)
    assert np.array_equal(ck.decrypt_many(out), [f(m) for m in range(M)])
    if p.N <= 256:      # exact-integer twin (schoolbook products) agrees at decrypt level and in phase
        exact = sk.apply_lookup_table_batch(cts, lut, exact=True)
        assert np.array_equal(ck.decrypt_many(exact), [f(m) for m in range(M)])
        d = max(torus_distance(ck.decrypt_plaintext(a), ck.decrypt_plaintext(b)) for a, b in zip(out, exact))
        assert d < 2.0 ** 50


def test_product_client_generates_the_oracle_multi_bit_keys():
    import fhestr
    for toy in (TOY, TOY_G3):
        g = _g(toy)
        ck, sk = _keys(toy)
        pck = fhestr.ClientKey(_fp(toy), ck.seed)
        bsk, ksk = pck.gen_server_keys(2)
        assert bsk.size == sk.bsk.size == toy.n // g * (1 << g) * 4 * toy.N
        assert np.array_equal(bsk, sk.bsk) and np.array_equal(ksk, sk.ksk)


# every multi-bit parameter shape of shortint/parameters/multi_bit.rs: N = 2048 (fused kernel / prepared GGSWs),
# N = 512 k = 3 and N = 8192 two levels (generic two-kernel path), plus toy shapes of the same kernels
OTHER_SHAPES = [O.TOY_MULTI_BIT_N256, O.TOY_MULTI_BIT_N256_G3, O.TOY_MULTI_BIT_N128_K2, O.TOY_MULTI_BIT_N512_K3_G3,
                O.TOY_MULTI_BIT_N8192, O.TOY_MULTI_BIT_N8192_G3, O.PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS]


@pytest.mark.gpu
@pytest.mark.parametrize("p", [TOY, REAL, TOY_G3, REAL_G3] + OTHER_SHAPES, ids=lambda p: p.name)
def test_gpu_multi_bit_pbs_matches_oracle(p):
    import fhestr
    ck, sk = _keys(p)
    eng = fhestr.Engine(_fp(p), 0)
    try:
        eng.load_keys(sk.bsk, sk.ksk)
        M = p.msg_mod * p.carry_mod
        f = lambda x: (7 * x + 3) % M
        lut, _ = sk.generate_lookup_table(f)
        lut_id = eng.upload_lut(lut)
        msgs = np.arange(2 * M) % M
        cts = ck.encrypt_many(msgs, O.Rng(0x4D42, 7))
        assert np.array_equal(eng.keyswitch(cts), np.stack([sk.keyswitch(c) for c in cts]))     # bit-exact
        got = eng.apply_lookup_table(cts, np.full(len(cts), lut_id, dtype=np.uint32))
        assert np.array_equal(ck.decrypt_many(got), [f(int(m)) for m in msgs])
        # 32 LWEs took the two-kernel path (GGSWs of every group prepared on the whole GPU, then n/G plain
        # external products); the fused kernel does the same operations in the same order
        eng.set_multibit_combine_max(0)
        fused = eng.apply_lookup_table(cts, np.full(len(cts), lut_id, dtype=np.uint32))
        assert np.array_equal(got, fused)
        eng.set_multibit_combine_max(64)
        # phase of the GPU result against the oracle's f64 path on the first few LWEs
        want = sk.apply_lookup_table_batch(cts[:4], lut)
        logN = p.N.bit_length() - 1
        tol = 2.0 ** (64 - (52 - p.pbs_base_log - logN)) * np.sqrt(p.n * 2 * (p.k + 1)) * np.sqrt(p.k * p.N / 2) \
            + 2.0 ** (64 - p.pbs_base_log) * np.sqrt(p.n * (p.k * p.N / 2 + 1))
        for a, b in zip(got[:4], want):
            assert torus_distance(ck.decrypt_plaintext(a), ck.decrypt_plaintext(b)) < tol
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_multi_bit_device_keygen_and_string_eq():
    """Device-side generation of the multi-bit key is bit-identical to the oracle's, and the string layer
    runs unchanged on top of a multi-bit engine."""
    import fhestr
    for p in (TOY, TOY_G3):
        _device_keygen_and_string_eq(p)


def _device_keygen_and_string_eq(p):
    import fhestr
    ck, sk = _keys(p)
    eng = fhestr.Engine(_fp(p), 0)
    try:
        bsk, ksk = eng.generate_keys(ck.glwe_sk, ck.small_sk, ck.seed, export=True)
        assert np.array_equal(bsk, sk.bsk) and np.array_equal(ksk, sk.ksk)
        ops = fhestr.FheStringOps(eng)
        fp = _fp(p)
        enc = lambda s: ck.encrypt_many(fhestr.string_to_blocks(fp, s, 8))
        dec = lambda ct: ck.decrypt_many(np.asarray(ct).reshape(-1, p.big_size))
        assert int(dec(ops.eq(enc(b"multibit"), enc(b"multibit")))[0]) == 1
        assert int(dec(ops.eq(enc(b"multibit"), enc(b"multibat")))[0]) == 0
        assert fhestr.blocks_to_string(fp, dec(ops.to_upper(enc(b"multiBit")))) == b"MULTIBIT"
    finally:
        eng.close()
