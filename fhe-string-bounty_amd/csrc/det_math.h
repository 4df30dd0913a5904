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

// xoshiro256** seeded through splitmix64, one independent stream per (seed, stream id)
struct Rng {
    uint64_t s[4];
    FHE_HD static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    FHE_HD Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
        for (int i = 0; i < 4; i++) s[i] = splitmix(x);
    }
    FHE_HD static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    FHE_HD uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
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
