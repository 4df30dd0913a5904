"""An exact-integer programmable bootstrap in numpy, fast enough for N = 32768 (test infrastructure).

The C oracle's exact path (oracle/tfhe_oracle.c: orc_pbs_exact) multiplies polynomials the schoolbook way, N^2 per
product: seconds per PBS at N = 32768.  Here a negacyclic product  digits (|d| < 2^15)  x  key words (u64)  mod
(X^N + 1, 2^64) is taken limb by limb: the key in eight 8-bit limbs, each limb's product an f64 FFT whose values stay
below N * 2^15 * 2^8 = 2^38 -- the transform's rounding error is then < 2^-10, so rounding to the nearest integer is
EXACT -- and the limbs recombine in wrapping 64-bit integers.  Same algorithm as bootstrap.rs:242-331 / ggsw.rs:477-598
with the f64 external product replaced by exact arithmetic (last paragraph); pinned bit for bit
against orc_pbs_exact on small N (tests/test_exact_pbs.py)."""
import numpy as np

U64 = np.uint64


def _twist(N):
    return np.exp(1j * np.pi * np.arange(N) / N)


def negacyclic_mul_exact(digits, key, tw=None):
    """sum_j digits[j] * key[i - j] with X^N = -1, mod 2^64.  digits: int64 (|d| < 2^15), key: uint64."""
    N = len(key)
    tw = _twist(N) if tw is None else tw
    D = np.fft.fft(digits.astype(np.float64) * tw)
    out = np.zeros(N, dtype=U64)
    with np.errstate(over="ignore"):
        for t in range(8):
            limb = ((key >> U64(8 * t)) & U64(0xFF)).astype(np.float64)
            prod = np.fft.ifft(D * np.fft.fft(limb * tw)) * np.conj(tw)
            r = np.rint(prod.real)
            assert np.abs(prod.real - r).max() < 0.05 and np.abs(prod.imag).max() < 0.05      # exactness margin
            out += r.astype(np.int64).astype(U64) << U64(8 * t)
    return out


def decompose(x, base_log, level):
    """Signed digits of closest_representable(x), level `level` first (decomposer.rs:98-118, iter.rs:101-127): list of int64 arrays."""
    rep = base_log * level
    t = x >> U64(63 - rep)
    state = ((t + U64(1)) >> U64(1)) & U64((1 << rep) - 1)
    mask = U64((1 << base_log) - 1)
    digits = []
    for _ in range(level):
        res = state & mask
        state = state >> U64(base_log)
        carry = (((res - U64(1)) | state) & res) >> U64(base_log - 1)
        state = state + carry
        digits.append(res.astype(np.int64) - (carry.astype(np.int64) << base_log))
    return digits


def modulus_switch(x, logN):
    return int(((int(x) >> (64 - logN - 2)) + 1) >> 1)


def monomial_mul(poly, d, N):
    """poly * X^d, d in [0, 2N] (polynomial_algorithms.rs:425-490 / 315-354 via X^-d = X^(2N-d))."""
    d %= 2 * N
    rem, odd = d % N, (d // N) & 1
    out = np.roll(poly, rem)
    with np.errstate(over="ignore"):
        out[:rem] = U64(0) - out[:rem]
        if odd:
            out = U64(0) - out
    return out


def pbs_exact(params, bsk, ct_small, lut):
    """params: oracle Params; bsk: standard-domain key [n][level, level 1 first][k+1][k+1][N]; lut: [(k+1) N]."""
    n, k, N, bl, L = params.n, params.k, params.N, params.pbs_base_log, params.pbs_level
    logN = N.bit_length() - 1
    K1 = k + 1
    bsk = np.asarray(bsk, dtype=U64).reshape(n, L, K1, K1, N)
    tw = _twist(N)
    acc = np.asarray(lut, dtype=U64).reshape(K1, N).copy()
    b = modulus_switch(ct_small[n], logN)
    acc = np.stack([monomial_mul(p, 2 * N - b, N) for p in acc])          # X^{-ms(body)}
    with np.errstate(over="ignore"):
        for i in range(n):
            if int(ct_small[i]) == 0:
                continue                                                    # bootstrap.rs:281
            d = modulus_switch(ct_small[i], logN)
            ct1 = np.stack([monomial_mul(p, d, N) - p for p in acc])
            digs = [decompose(ct1[r], bl, L) for r in range(K1)]           # [row][it], it = 0 is level L
            for it in range(L):
                lvl = L - 1 - it                                            # index into "level 1 first"
                for r in range(K1):
                    for col in range(K1):
                        acc[col] += negacyclic_mul_exact(digs[r][it], bsk[i, lvl, r, col], tw)
    out = np.zeros(k * N + 1, dtype=U64)                                    # sample extraction at degree 0
    with np.errstate(over="ignore"):
        for p in range(k):
            out[p * N] = acc[p][0]
            out[p * N + 1: (p + 1) * N] = U64(0) - acc[p][:0:-1]
    out[k * N] = acc[k][0]
    return out
