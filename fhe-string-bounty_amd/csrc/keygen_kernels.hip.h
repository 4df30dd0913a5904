// keygen_kernels.hip.h -- server-key generation on the device:
//   ksk_gen_kernel   allocate_and_generate_new_lwe_keyswitch_key
//                    (core_crypto/algorithms/lwe_keyswitch_key_generation.rs:65-130)
//   bsk_gen_kernel   par_allocate_and_generate_new_lwe_bootstrap_key
//                    (lwe_bootstrap_key_generation.rs:76-135,237-300; ggsw_encryption.rs:72-151,300-331;
//                     glwe_encryption.rs:17-60)
// Both write the reference's standard-domain layouts straight into HBM; the engine then runs its
// usual conversions (ksk_pack_kernel / bsk_convert_kernel) without the keys ever visiting the host.
// Randomness and noise are det_math.h's (one ChaCha20 stream per key row, polar Gaussian with a
// libm-free logarithm), drawn in the same order as client.cpp and oracle/tfhe_oracle.c, so the
// generated keys are bit-identical to the CPU ones for the same (secret keys, seed) -- the parity
// test compares them word for word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)
#include "det_math.h"

namespace fhe {

struct KeygenArgs {
    const uint64_t* glwe_sk;    // [k*N] 0/1 (also the big LWE key)
    const uint64_t* small_sk;   // [n]   0/1
    const uint64_t* bsk_bits;   // [n_ggsw] plaintext bit of every GGSW (classic: = small_sk; multi-bit:
                                // products of the group's key bits, engine.h multi_bit_key_bit)
    uint64_t* ksk;              // [k*N][ks_level][n+1]
    uint64_t* bsk;              // [n][pbs_level][k+1][k+1][N]
    Seed256 seed;               // ChaCha20 key (det_math.h)
    uint32_t n, k, N;
    uint32_t pbs_base_log, pbs_level, ks_base_log, ks_level;
    double lwe_std, glwe_std;
};

constexpr uint64_t KSK_STREAM = 0x4B534B0000000000ull;   // + input key index
constexpr uint64_t BSK_STREAM = 0x42534B0000000000ull;   // + small key index

// One thread per input key coefficient i: ks_level LWE encryptions of s_i * q / beta^level under
// the small key (level ks_level first), all from stream KSK_STREAM + i.
__global__ void __launch_bounds__(64) ksk_gen_kernel(KeygenArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t in_dim = a.k * a.N;
    if (i >= in_dim) return;
    Rng r(a.seed, KSK_STREAM + i);
    const uint64_t bit = a.glwe_sk[i];
    const size_t osz = (size_t)a.n + 1;
    for (uint32_t it = 0; it < a.ks_level; it++) {
        const uint32_t level = a.ks_level - it;
        const uint64_t pt = bit << (64 - a.ks_base_log * level);
        uint64_t* ct = a.ksk + ((size_t)i * a.ks_level + it) * osz;
        uint64_t acc = 0;
        for (uint32_t j = 0; j < a.n; j++) {        // lwe_encryption.rs:61-110
            const uint64_t m = r.next();
            ct[j] = m;
            acc += m * a.small_sk[j];
        }
        ct[a.n] = acc + gaussian_torus(r, a.lwe_std) + pt;
    }
}

// One workgroup per GGSW i of the key (classic: small-key coefficient i).  For every GLWE row of the GGSW:
// thread 0 draws the mask and the noise in stream order (the only sequential part), then all threads
// add the plaintext term and  sum_q A_q * S_q  (binary key: signed shifted adds, negacyclic).
__global__ void __launch_bounds__(256) bsk_gen_kernel(KeygenArgs a) {
    const uint32_t i = blockIdx.x;
    const uint32_t N = a.N, k = a.k, k1 = k + 1, L = a.pbs_level;
    const size_t glwe_len = (size_t)k1 * N, ggsw_len = (size_t)L * k1 * glwe_len;
    uint64_t* ggsw = a.bsk + (size_t)i * ggsw_len;
    const uint64_t m = a.bsk_bits[i];
    Rng r(a.seed, BSK_STREAM + i);          // only thread 0's copy advances
    for (uint32_t li = 0; li < L; li++) {
        const uint64_t factor = (0 - m) * (1ull << (64 - a.pbs_base_log * (li + 1)));
        for (uint32_t row = 0; row < k1; row++) {
            uint64_t* glwe = ggsw + ((size_t)li * k1 + row) * glwe_len;
            uint64_t* body = glwe + (size_t)k * N;
            if (threadIdx.x == 0) {
                for (size_t j = 0; j < (size_t)k * N; j++) glwe[j] = r.next();
                for (uint32_t j = 0; j < N; j++) body[j] = gaussian_torus(r, a.glwe_std);
            }
            __syncthreads();
            for (uint32_t c = threadIdx.x; c < N; c += blockDim.x) {
                uint64_t v = body[c];
                // ggsw_encryption.rs:300-331: row < k carries -m * S_row * q/beta^level, the last row
                // +m * q/beta^level on the constant coefficient
                if (row < k) v += a.glwe_sk[(size_t)row * N + c] * factor;
                else if (c == 0) v += 0 - factor;
                for (uint32_t q = 0; q < k; q++) {
                    const uint64_t* A = glwe + (size_t)q * N;
                    const uint64_t* S = a.glwe_sk + (size_t)q * N;
                    for (uint32_t t = 0; t < N; t++) {
                        if (!S[t]) continue;                       // workgroup-uniform
                        v += c >= t ? A[c - t] : 0 - A[c + N - t];
                    }
                }
                body[c] = v;
            }
            __syncthreads();
        }
    }
}

}  // namespace fhe
