"""Functional pins of the oracle's f64 PBS against the reference's own test expectations
(decrypt-level, as in fft64/crypto/tests.rs:5-13, algorithms/test/lwe_programmable_bootstrapping.rs:70-166,
shortint/server_key/tests/shortint.rs:366-560) and FFT tolerances (fft64/math/fft/tests.rs)."""
import numpy as np
import pytest

import oracle as O
from conftest import keyset, torus_distance


@pytest.mark.parametrize("N", [64, 256, 2048])
def test_fft_roundtrip_within_2_pow_14(N):
    # fft/tests.rs:9-80 : forward_as_torus -> add_backward: error < 2^(64-50)
    rng = np.random.default_rng(N)
    poly = rng.integers(0, 2**64, size=N, dtype=np.uint64)
    one = np.zeros(N, dtype=np.uint64)
    one[0] = 1
    out = O.fft_roundtrip_product(N, poly, one)
    assert torus_distance(out, poly).max() < 2.0**14


@pytest.mark.parametrize("N", [64, 256, 2048])
def test_fft_product_vs_schoolbook(N):
    # fft/tests.rs:82-222 : torus poly x 16-bit integer poly; threshold 2^(64-(52-16-log2 N))
    rng = np.random.default_rng(N + 1)
    a = rng.integers(0, 2**64, size=N, dtype=np.uint64)
    b = rng.integers(0, 2**16, size=N, dtype=np.uint64)
    got = O.fft_roundtrip_product(N, a, b)
    want = O.negacyclic_schoolbook(b, a)
    assert torus_distance(got, want).max() <= 2.0 ** (64 - (52 - 16 - int(np.log2(N))))


@pytest.mark.parametrize("params", [O.TOY_K1, O.TOY_K2], ids=lambda p: p.name)
def test_pbs_all_messages_fft_and_exact(params):
    # algorithms/test/lwe_programmable_bootstrapping.rs:70-166 : every message, LUT = identity-like
    ks = keyset(params)
    M = params.msg_mod * params.carry_mod
    for f in (lambda x: x, lambda x: (2 * x) % M, lambda x: (x // params.msg_mod) % params.carry_mod,
              lambda x: x % params.msg_mod):   # identity, double, carry_extract, message_extract
        lut, deg = ks.sk.generate_lookup_table(f)
        assert deg == max(f(i) for i in range(M))
        cts = ks.ck.encrypt_many(range(M))
        for exact in (False, True):
            out = ks.sk.apply_lookup_table_batch(cts, lut, exact=exact)
            assert ks.ck.decrypt_many(out).tolist() == [f(m) for m in range(M)]


def test_bivariate_lut_2xy_mod4(toy_k1):
    # shortint.rs `shortint_bivariate_programmable_bootstrap`-style pin: f(x, y) = (2*x*y) % 4
    p = toy_k1.params
    lut, _ = toy_k1.sk.generate_lookup_table_bivariate(lambda x, y: (2 * x * y) % p.msg_mod)
    for x in range(p.msg_mod):
        for y in range(p.msg_mod):
            # unchecked_apply_lookup_table_bivariate: lhs * factor + rhs (bivariate_pbs.rs:167-182)
            with np.errstate(over="ignore"):
                ct = toy_k1.ck.encrypt(x) * np.uint64(p.msg_mod) + toy_k1.ck.encrypt(y)
            out = toy_k1.sk.apply_lookup_table(ct, lut)
            assert toy_k1.ck.decrypt(out) == (2 * x * y) % p.msg_mod


def test_trivial_pbs_shortcut_matches_real_pbs(toy_k1):
    # shortint/server_key/mod.rs:763-781 : clear lookup for trivial ciphertexts
    p = toy_k1.params
    M = p.msg_mod * p.carry_mod
    lut, _ = toy_k1.sk.generate_lookup_table(lambda x: (7 * x + 2) % M)
    for m in range(2 * M):  # includes padding-bit-set values
        body = (m * p.delta) % 2**64
        got = toy_k1.sk.trivial_pbs_body(body, lut)
        ct = np.zeros(p.big_size, dtype=np.uint64)
        ct[-1] = body
        real = toy_k1.sk.apply_lookup_table(ct, lut)
        assert toy_k1.ck.decrypt_plaintext(real) - got in (0,) or \
            torus_distance([toy_k1.ck.decrypt_plaintext(real)], [got]).max() < p.delta / 4


@pytest.mark.slow
def test_pbs_p22_doctest_2_times_3(p22):
    # algorithms/lwe_programmable_bootstrapping.rs:973-1015 : 4-bit message 3, LUT x -> 2x gives 6
    lut, _ = p22.sk.generate_lookup_table(lambda x: 2 * x)
    ct = p22.ck.encrypt(3)
    out = p22.sk.apply_lookup_table(ct, lut)
    pt = p22.ck.decrypt_plaintext(out)
    assert O.closest_representable(pt, 5, 1) // p22.params.delta == 6
    assert p22.ck.decrypt_message_and_carry(out) == 6
