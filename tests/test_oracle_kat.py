"""Pin the CPU oracle to the reference's own deterministic known-answer vectors (its doctests and
unit tests; there are no committed ciphertext fixtures -- SURVEY.md 8(c), F7).  Citations are
file:line under /root/reference/tfhe/src."""
import math
import struct

import numpy as np
import pytest

import oracle as O


def test_closest_representable_u32_doctest():
    # core_crypto/commons/math/decomposition/decomposer.rs:94-95
    assert O.closest_representable(1_340_987_234, 4, 3, bits=32) == 1_341_128_704


def test_decompose_terms_in_range_u32_doctest():
    # decomposer.rs:134-142 : every term in [-B/2, B/2), three terms
    digits = O.decompose(1_340_987_234, 4, 3, bits=32)
    assert len(digits) == 3
    for d in digits:
        signed = int(d) - (1 << 32) if int(d) >= (1 << 31) else int(d)
        assert -8 <= signed < 8
    assert len(O.decompose(1, 4, 3, bits=32)) == 3


@pytest.mark.parametrize("val", [1_340_987_234, 0, 1, 2**32 - 1, 0x80000000, 0x7FFFFFFF, 123456789])
def test_recompose_equals_closest_u32(val):
    # decomposer.rs:166-169 : sum_i term_i * q / B^i == closest_representable(val)
    b, l = 4, 3
    digits = O.decompose(val, b, l, bits=32)           # level l first
    total = 0
    for it, d in enumerate(digits):
        level = l - it
        total += int(d) << (32 - b * level)
    assert total % 2**32 == O.closest_representable(val, b, l, bits=32)


def test_recompose_equals_closest_u64_random():
    rng = np.random.default_rng(5)
    for b, l in [(23, 1), (15, 2), (3, 5), (4, 3), (3, 7), (10, 2)]:
        for x in rng.integers(0, 2**64, size=200, dtype=np.uint64):
            digits = O.decompose(int(x), b, l)
            total = sum(int(d) << (64 - b * (l - it)) for it, d in enumerate(digits)) % 2**64
            assert total == O.closest_representable(int(x), b, l)
            for d in digits:  # signed digit in [-B/2, B/2]
                s = int(d) - 2**64 if int(d) >= 2**63 else int(d)
                assert -(1 << (b - 1)) <= s <= (1 << (b - 1))


def test_monomial_div_u8_doctest():
    # core_crypto/algorithms/polynomial_algorithms.rs:310-313
    assert O.monomial_div([1, 2, 3], 2, bits=8).tolist() == [3, 255, 254]


def test_monomial_mul_u8_doctest():
    # polynomial_algorithms.rs:370-373
    assert O.monomial_mul([1, 2, 3], 2, bits=8).tolist() == [254, 253, 1]


def test_monomial_mul_and_subtract_is_mul_minus_identity():
    rng = np.random.default_rng(6)
    N = 64
    p = rng.integers(0, 2**64, size=N, dtype=np.uint64)
    for d in [0, 1, 17, N - 1, N, N + 5, 2 * N - 1, 2 * N]:
        with np.errstate(over="ignore"):
            want = O.monomial_mul(p, d) - p
        assert np.array_equal(O.monomial_mul_and_subtract(p, d), want)
        # div is the inverse of mul
        assert np.array_equal(O.monomial_div(O.monomial_mul(p, d), d), p)


def test_slice_sub_scalar_mul_u8_doctest():
    # core_crypto/algorithms/slice_algorithms.rs:358-362
    got = O.slice_sub_scalar_mul([1, 2, 3, 4, 5, 6], [255, 255, 255, 1, 2, 3], 4, bits=8)
    assert got.tolist() == [5, 6, 7, 0, 253, 250]


def test_f64_to_i64_table():
    # core_crypto/fft_impl/fft64/math/fft/tests.rs:244-301 (values where `x as i64` is specified)
    for x in [0.0, -0.0, 37.1242161, -37.1242161, 0.1, -0.1, 1.0, -1.0, 0.9, -0.9, 2.0, -2.0,
              1e-310, -1e-310, 2.0**62, -(2.0**62), 1.1 * 2.0**62, -1.1 * 2.0**62, -(2.0**63)]:
        bits = struct.unpack("<Q", struct.pack("<d", x))[0]
        mant = (bits & 0xFFFFFFFFFFFFF) | 0x10000000000000
        bexp = (bits >> 52) & 0x7FF
        sign = bits >> 63
        rs = 1086 - bexp
        v = ((mant << 11) & (2**64 - 1)) >> rs if rs < 64 else 0
        v = v if sign == 0 else -v
        v = 0 if bexp == 0 else v
        assert O.f64_to_i64(x) == v == int(x)


def test_from_torus_convention():
    # core_crypto/commons/math/torus/mod.rs:72-78
    assert O.from_torus(0.0) == 0
    assert O.from_torus(0.25) == 1 << 62
    assert O.from_torus(-0.25) == (2**64 - (1 << 62))
    assert O.from_torus(1.25) == 1 << 62          # integer part discarded
    assert O.from_torus(2.0**-64) == 1


def test_modulus_switch_range():
    # core_crypto/fft_impl/common.rs:20-43 : result in [0, 2N], 2N reachable
    logN = 11
    assert O.modulus_switch(0, logN) == 0
    assert O.modulus_switch(2**64 - 1, logN) == 2 * 2048
    assert O.modulus_switch(1 << 63, logN) == 2048
    rng = np.random.default_rng(7)
    for x in rng.integers(0, 2**64, size=100, dtype=np.uint64):
        want = int(round(int(x) * (2 * 2048) / 2**64))  # exact: python ints / correctly rounded
        got = O.modulus_switch(int(x), logN)
        assert abs(got - int(x) * 4096 / 2**64) <= 0.5 + 1e-9 and 0 <= got <= 4096


def test_fill_accumulator_structure():
    # shortint/engine/mod.rs:72-128
    p = O.PARAM_MESSAGE_2_CARRY_2_KS_PBS
    table = np.arange(16, dtype=np.uint64)[::-1].copy()
    lut = np.zeros(p.glwe_len, dtype=np.uint64)
    import ctypes as C
    deg = O.lib().orc_fill_accumulator(C.byref(p.c()), table, lut)
    assert deg == 15
    assert not lut[: p.N].any()                         # mask polynomial zero
    body = lut[p.N:]
    box = p.N // 16
    # after rotate_left(box/2): first half box holds f(0)*delta, last half box holds -f(0)*delta
    assert all(int(v) == 15 * p.delta for v in body[: box // 2])
    assert all(int(v) == (-(15 * p.delta)) % 2**64 for v in body[-(box // 2):])
    for i in range(1, 16):
        seg = body[i * box - box // 2: (i + 1) * box - box // 2]
        assert all(int(v) == int(table[i]) * p.delta for v in seg)


def test_keyswitch_doctest_message_survives():
    # core_crypto/algorithms/lwe_keyswitch.rs:28-94 : 742 -> 2048 in the doctest (either direction is
    # the same algorithm); msg 3<<60 must survive, rounded on the 4 MSBs.
    p = O.PARAM_MESSAGE_2_CARRY_2_KS_PBS
    ck = O.ClientKey(p, 11)
    import ctypes as C
    ksk = np.zeros(p.big_dim * p.ks_level * p.small_size, dtype=np.uint64)
    O.lib().orc_gen_ksk(C.byref(p.c()), ck.big_sk, ck.small_sk, O.seed_bytes(11), ksk)
    ct = ck.encrypt_plaintext(3 << 60)
    out = np.zeros(p.small_size, dtype=np.uint64)
    O.lib().orc_keyswitch(C.byref(p.c()), ksk, ct, out)
    dec = ck.decrypt_small_plaintext(out)
    assert O.closest_representable(dec, 4, 1) >> 60 == 3


def test_chacha20_block_rfc8439_vector():
    """The harness / client generator's block function against RFC 8439 section 2.3.2 (key 00..1f, counter 1,
    nonce 00:00:00:09:00:00:00:4a:00:00:00:00 = words 13, 14, 15), in the oracle and in libfhestr's twin."""
    import fhestr
    key = bytes(range(32))
    counter, stream = 1 | (0x09000000 << 32), 0x4A000000
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert O.chacha20_block(key, counter, stream).tolist() == want
    assert fhestr.chacha20_block(key, counter, stream).tolist() == want


def test_client_and_oracle_draw_the_same_stream():
    """Same seed, same stream ids: the product's client keys equal the oracle's; different seeds differ; an
    OS-entropy seed gives yet another key."""
    import fhestr
    from conftest import to_fhestr_params
    p = O.TOY_K1
    P = to_fhestr_params(p)
    a = fhestr.ClientKey(P, 0x1234).secret_keys()
    b = O.ClientKey(p, 0x1234)
    assert np.array_equal(a[0], b.glwe_sk) and np.array_equal(a[1], b.small_sk)
    assert not np.array_equal(fhestr.ClientKey(P, 0x1235).secret_keys()[0], a[0])
    s1, s2 = fhestr.random_seed(), fhestr.random_seed()
    assert len(s1) == 32 and s1 != s2
    assert not np.array_equal(fhestr.ClientKey(P, s1).secret_keys()[0], fhestr.ClientKey(P, s2).secret_keys()[0])
