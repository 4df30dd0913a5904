// det_math.h -- floating-point helpers of the noise sampler written with IEEE +,-,*,/ and sqrt only,
// so that host (client.cpp) and device (keygen_kernels.hip.h) key generation produce identical bits
// (and identical to oracle/tfhe_oracle.c, which restates the same formulas).  Translation units that
// include this must not contract a*b+c into fma for these functions: the library is built with
// -ffp-contract=off and the device header re-asserts it with a pragma.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define FHE_HD __host__ __device__ inline
#else
#define FHE_HD inline
#endif

namespace fhe {

// ln(x) for normal positive x: x = m 2^e, m in [sqrt(1/2), sqrt 2), ln m = 2 atanh((m-1)/(m+1)) as
// an odd series through f^23 (|f| <= 0.1716: truncation < 2^-60).  Stands in for f64::ln of the
// reference's sampler (core_crypto/commons/math/random/gaussian.rs:33).
FHE_HD double det_log(double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double f = (m - 1.0) / (m + 1.0);
    const double f2 = f * f;
    double p = 1.0 / 23.0;
    p = p * f2 + 1.0 / 21.0;
    p = p * f2 + 1.0 / 19.0;
    p = p * f2 + 1.0 / 17.0;
    p = p * f2 + 1.0 / 15.0;
    p = p * f2 + 1.0 / 13.0;
    p = p * f2 + 1.0 / 11.0;
    p = p * f2 + 1.0 / 9.0;
    p = p * f2 + 1.0 / 7.0;
    p = p * f2 + 1.0 / 5.0;
    p = p * f2 + 1.0 / 3.0;
    const double series = 2.0 * f + 2.0 * f * (f2 * p);
    return (double)e * 0.6931471803691238 + (series + (double)e * 1.9082149292705877e-10);
}

// f64::round (ties away from zero) without libm; exact.
FHE_HD double round_half_away(double x) {
    if (x >= 4503599627370496.0 || x <= -4503599627370496.0) return x;   // |x| >= 2^52: already integral
    const double t = (double)(int64_t)x;
    const double d = x - t;
    if (d >= 0.5) return t + 1.0;
    if (d <= -0.5) return t - 1.0;
    return t;
}

// 256-bit seed = ChaCha20 key.  Everything random on the client side and in server-key generation --
// secret keys, masks, noise -- is ChaCha20 keystream under this key, one independent stream per
// (purpose, row) through the 64-bit nonce.  Keystream is public-safe: the masks published in every
// ciphertext and key row are outputs of a PRF and reveal neither the seed nor any other part of any
// stream (the noise, the secret keys).  Stands in for the reference's AES-128-CTR generator with forked
// byte ranges (concrete-csprng/src/generators, core_crypto/commons/generators/encryption.rs); the seed
// comes from the OS (fhe_random_seed) unless a test fixes it.
struct Seed256 {
    uint32_t w[8];
};

FHE_HD Seed256 seed_from_bytes(const uint8_t b[32]) {     // little-endian words, as RFC 8439 loads a key
    Seed256 s;
    for (int i = 0; i < 8; i++)
        s.w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return s;
}

// The ChaCha20 block function (D. J. Bernstein, 20 rounds; RFC 8439 section 2.3 with the original 64-bit
// counter / 64-bit nonce split of words 12..15).
FHE_HD void chacha20_block(const Seed256& key, uint64_t counter, uint64_t stream, uint32_t out[16]) {
    uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t in[16];
    for (int i = 0; i < 16; i++) in[i] = x[i];
#define FHE_ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define FHE_QR(a, b, c, d)                                  \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = FHE_ROTL32(x[d], 16); \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = FHE_ROTL32(x[b], 12); \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = FHE_ROTL32(x[d], 8);  \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = FHE_ROTL32(x[b], 7);
    for (int r = 0; r < 10; r++) {
        FHE_QR(0, 4, 8, 12) FHE_QR(1, 5, 9, 13) FHE_QR(2, 6, 10, 14) FHE_QR(3, 7, 11, 15)
        FHE_QR(0, 5, 10, 15) FHE_QR(1, 6, 11, 12) FHE_QR(2, 7, 8, 13) FHE_QR(3, 4, 9, 14)
    }
#undef FHE_QR
#undef FHE_ROTL32
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

// Sequential reader of one stream: 64-bit words, little endian, 8 per block.
struct Rng {
    Seed256 key;
    uint64_t stream, counter;
    uint32_t buf[16];
    int pos;
    FHE_HD Rng(const Seed256& seed, uint64_t stream_id) : key(seed), stream(stream_id), counter(0), pos(16) {}
    FHE_HD uint64_t next() {
        if (pos >= 16) {
            chacha20_block(key, counter++, stream, buf);
            pos = 0;
        }
        const uint64_t v = (uint64_t)buf[pos] | ((uint64_t)buf[pos + 1] << 32);
        pos += 2;
        return v;
    }
};

// core_crypto/commons/math/torus/mod.rs:72-78 for |x| small (noise samples)
FHE_HD uint64_t from_torus_exact(double x) {
    double fr = x - round_half_away(x);
    fr = round_half_away(fr * 18446744073709551616.0);
    if (fr >= 9223372036854775808.0) return (uint64_t)INT64_MAX;
    if (fr <= -9223372036854775808.0) return (uint64_t)INT64_MIN;
    return (uint64_t)(int64_t)fr;
}

// core_crypto/commons/math/random/gaussian.rs:17-47,85-97: Marsaglia polar method on two signed
// 64-bit draws, first sample of the pair, as a torus element
FHE_HD uint64_t gaussian_torus(Rng& r, double std_dev) {
    for (;;) {
        const double u = (double)(int64_t)r.next() * 1.0842021724855044e-19;   // 2^-63
        const double v = (double)(int64_t)r.next() * 1.0842021724855044e-19;
        const double s = u * u + v * v;
        if (s > 0.0 && s < 1.0) {
            const double cst = std_dev * __builtin_sqrt(-2.0 * det_log(s) / s);
            return from_torus_exact(u * cst);
        }
    }
}

}  // namespace fhe
