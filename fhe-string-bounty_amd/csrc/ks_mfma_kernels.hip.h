// ks_mfma_kernels.hip.h -- the LWE keyswitch as an exact 8-bit integer matrix product on the matrix cores.
//
// Replaces keyswitch_lwe_ciphertext (core_crypto/algorithms/lwe_keyswitch.rs:96-170) for whole batches:
//     out[b] = (0, ..., 0, body_b) - sum_{i, lv} digit(b, i, lv) * KSK[i][lv][:]        (mod 2^64)
// The signed digits of the reference's decomposer (decomposer.rs:98-152, iter.rs:37-127) are at most
// 2^(base_log - 1) <= 64 in magnitude: they are int8 as they stand.  Every 64-bit key word is rewritten ONCE, at key
// load, in balanced base 256:  K = sum_t s_t 2^(8 t)  (mod 2^64),  s_t in [-128, 127]  -- an exact representation, the
// carry out of the top digit is the reduction mod 2^64.  Then
//     sum_r d_r K_r = sum_t 2^(8 t) ( sum_r d_r s_t(r) )            (mod 2^64)
// and the inner sums are int8 x int8 -> int32 dot products: v_mfma_i32_32x32x32_i8 (|d s| <= 2^13, rows per workgroup
// <= 2^15: no int32 overflow -- the host clamps the K chunk, ks_mfma_max_steps).  Nothing is rounded anywhere: the result is bit-identical to the reference's loop
// (tests/test_gpu_parity.py::test_keyswitch_bit_exact, golden fixtures).
//
// Shapes: M = batch (32-row tiles, one wave each), N = output columns x 8 digit planes (a workgroup owns 32 columns,
// every wave carries all 8 planes of its 32 rows x 32 columns, so the planes recombine in registers), K = kN * level key
// rows, walked in steps of 32 k-slots.  The MFMA pairs element j of lane half h of A with the same (h, j) of B, so the
// assignment of key rows to k-slots is ours: slot group (step, h) holds floor(16 / level) whole mask elements with all
// their levels (digits of one element are produced together), the remaining slots are zero digits.
//   key   [column group][step][plane][lane][16] int8   exactly the B fragments, 8 KB per (group, step): one coalesced
//                                                       16-byte load per thread, staged through LDS for the waves
//   digits[row tile][step][lane][16] int8              exactly the A fragments, written by ks_decompose_kernel
// K is split over workgroups (grid.z) so that the launch fills the GPU; partial sums meet in 64-bit integer atomics
// (wrapping addition is associative and commutative: deterministic, bit-exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe {

typedef int ksm_v4i __attribute__((ext_vector_type(4)));
typedef int ksm_v16i __attribute__((ext_vector_type(16)));

struct KsMfmaGeom {
    uint32_t in_dim, out_size, level, base_log;
    uint32_t epg;          // mask elements per 16-slot group = 16 / level
    uint32_t steps;        // k-steps = ceil(in_dim / (2 * epg))
    uint32_t col_groups;   // ceil(out_size / 32)
};

// A 16-slot group holds floor(16 / level) whole mask elements: at most 16 levels.
inline bool ks_mfma_supported(uint32_t level) { return level >= 1 && level <= 16; }
// K-steps one workgroup may accumulate in int32: 32 slots per step, |digit * key digit| <= 2^(base_log - 1) * 128
// => 32 * steps * 2^(base_log + 6) < 2^31.
inline uint32_t ks_mfma_max_steps(uint32_t base_log) { return (1u << (20 - base_log)) - 1u; }

__host__ __device__ inline KsMfmaGeom ks_mfma_geom(uint32_t in_dim, uint32_t out_size, uint32_t level, uint32_t base_log) {
    KsMfmaGeom g;
    g.in_dim = in_dim; g.out_size = out_size; g.level = level; g.base_log = base_log;
    g.epg = 16 / level;
    g.steps = (in_dim + 2 * g.epg - 1) / (2 * g.epg);
    g.col_groups = (out_size + 31) / 32;
    return g;
}

// Key words -> balanced base-256 digit planes in B-fragment order.  One thread per (column group, step, lane, plane-octet).
__global__ void __launch_bounds__(64) ksk_repack_mfma_kernel(const uint64_t* __restrict__ ksk, int8_t* __restrict__ out, KsMfmaGeom g) {
    const uint32_t cg = blockIdx.x, step = blockIdx.y, lane = threadIdx.x;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t col = cg * 32 + c;
    int8_t frag[8][16];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int j = 0; j < 16; j++) frag[t][j] = 0;
    for (uint32_t j = 0; j < g.epg * g.level; j++) {
        const uint32_t i = (step * 2 + h) * g.epg + j / g.level, lv = j % g.level;
        uint64_t k = 0;
        if (i < g.in_dim && col < g.out_size) k = ksk[((size_t)i * g.level + lv) * g.out_size + col];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int8_t s = (int8_t)(uint8_t)(k & 0xff);        // balanced digit: low byte read as signed
            frag[t][j] = s;
            k = (k - (uint64_t)(int64_t)s) >> 8;                  // exact: k - s is a multiple of 256 (mod 2^64)
        }
    }
    int8_t* dst = out + (((size_t)cg * g.steps + step) * 8) * 1024 + (size_t)lane * 16;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        ksm_v4i w;
        __builtin_memcpy(&w, frag[t], 16);
        *reinterpret_cast<ksm_v4i*>(dst + (size_t)t * 1024) = w;
    }
}

struct KsDecomposeArgs {
    const uint64_t* lwe_in;     // [batch][in_dim + 1]
    int8_t* digits;             // [row tile][step][64][16], pad slots zeroed at allocation and never written
    KsMfmaGeom g;
    uint32_t batch;
};

// One thread per (sample, 16-slot group): the reference's signed decomposition (level L first, iter.rs:101-127) of the
// group's floor(16 / level) mask elements, stored as ONE 16-byte A-fragment element (pad slots written as zero digits).
__global__ void __launch_bounds__(256) ks_decompose_kernel(KsDecomposeArgs a) {
    const uint32_t grp = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (grp >= 2 * a.g.steps) return;
    const uint32_t L = a.g.level, bl = a.g.base_log, rep = bl * L;
    const uint64_t mask = (1ull << bl) - 1;
    const uint64_t* src = a.lwe_in + (size_t)b * (a.g.in_dim + 1);
    int8_t frag[16];
#pragma unroll
    for (int j = 0; j < 16; j++) frag[j] = 0;
    for (uint32_t e = 0; e < a.g.epg; e++) {
        const uint32_t i = grp * a.g.epg + e;
        const uint64_t x = i < a.g.in_dim ? src[i] : 0;
        const uint64_t t = x >> (63 - rep);
        uint64_t state = ((t + 1) >> 1) & ((1ull << rep) - 1);
        for (uint32_t lv = 0; lv < L; lv++) {
            uint64_t res = state & mask;
            state >>= bl;
            uint64_t carry = ((res - 1ull) | state) & res;
            carry >>= bl - 1;
            state += carry;
            const int8_t d = (int8_t)((int32_t)(uint32_t)res - (int32_t)((uint32_t)carry << bl));
            // frag[e * L + lv] without a dynamically indexed private array
#pragma unroll
            for (int j = 0; j < 16; j++) frag[j] = (uint32_t)j == e * L + lv ? d : frag[j];
        }
    }
    const uint32_t step = grp >> 1, h = grp & 1;
    ksm_v4i w;
    __builtin_memcpy(&w, frag, 16);
    *reinterpret_cast<ksm_v4i*>(a.digits + (((size_t)(b >> 5) * a.g.steps + step) * 64 + (h * 32 + (b & 31))) * 16) = w;
}

struct KsMfmaArgs {
    const uint64_t* lwe_in;     // bodies are added here
    const int8_t* key;          // ksk_repack_mfma_kernel output
    const int8_t* digits;       // ks_decompose_kernel output
    uint64_t* lwe_out;          // [batch][out_size], zero-filled before the launch
    KsMfmaGeom g;
    uint32_t batch, row_tiles, steps_per_chunk;
};

// MT waves per workgroup = MT row tiles (32 samples each); grid (column groups, ceil(row tiles / MT), K chunks).
template <int MT>
__global__ void __launch_bounds__(64 * MT) keyswitch_mfma_kernel(KsMfmaArgs a) {
    constexpr int NT = 64 * MT;
    __shared__ __align__(16) int8_t bbuf[2][8 * 1024];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cg = blockIdx.x, rt = blockIdx.y * MT + wave;
    const uint32_t s0 = blockIdx.z * a.steps_per_chunk;
    const uint32_t s1 = min(s0 + a.steps_per_chunk, a.g.steps);
    const bool active = rt < a.row_tiles;                   // waves past the batch only help staging the key
    const int8_t* kbase = a.key + ((size_t)cg * a.g.steps) * 8 * 1024;
    ksm_v16i acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0;

    constexpr int PER_THREAD = 8 * 1024 / 16 / NT;          // 16-byte pieces of a key tile per thread
    constexpr int DEPTH = PER_THREAD >= 4 ? 2 : 4;          // key tiles in flight per workgroup (registers) behind the LDS pair
    ksm_v4i stage[DEPTH][PER_THREAD];
    ksm_v4i afrag[DEPTH];                                   // this wave's digit fragments, requested as far ahead as the key
    const int8_t* abase = a.digits + ((size_t)(active ? rt : 0) * a.g.steps * 64 + lane) * 16;
    auto fetch = [&](uint32_t step, int slot) {
        const ksm_v4i* src = reinterpret_cast<const ksm_v4i*>(kbase + (size_t)step * 8 * 1024);
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) stage[slot][q] = src[q * NT + tid];
        afrag[slot] = *reinterpret_cast<const ksm_v4i*>(abase + (size_t)step * 1024);
    };
    auto deposit = [&](int buf, int slot) {
        ksm_v4i* dstv = reinterpret_cast<ksm_v4i*>(bbuf[buf]);
#pragma unroll
        for (int q = 0; q < PER_THREAD; q++) dstv[q * NT + tid] = stage[slot][q];
    };
    // software pipeline: tile s+DEPTH is requested while tile s is multiplied; tile s+1 moves registers -> LDS behind it
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (s0 + d < s1) fetch(s0 + d, d);
    if (s0 < s1) deposit(0, 0);
    __syncthreads();
    int cur = 0;
    // the loop is unrolled by DEPTH so that the register slots are compile-time
    for (uint32_t base = s0; base < s1; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t step = base + d;
            if (step >= s1) break;
            const ksm_v4i av = afrag[d];
            if (step + DEPTH < s1) fetch(step + DEPTH, d);           // slot d: key tile deposited one step ago, digits just taken
            const ksm_v4i* bl = reinterpret_cast<const ksm_v4i*>(bbuf[cur]);
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bl[t * 64 + lane], acc[t], 0, 0, 0);
            if (step + 1 < s1) deposit(cur ^ 1, (d + 1) % DEPTH);
            __syncthreads();
            cur ^= 1;
        }
    }
    if (!active) return;
    // C layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const uint32_t col = cg * 32 + (lane & 31);
    if (col >= a.g.out_size) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t b = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (b >= a.batch) continue;
        uint64_t p = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) p += (uint64_t)(int64_t)acc[t][r] << (8 * t);
        uint64_t v = 0 - p;
        if (blockIdx.z == 0 && col == a.g.out_size - 1) v += a.lwe_in[(size_t)b * (a.g.in_dim + 1) + a.g.in_dim];   // body (:146)
        atomicAdd(reinterpret_cast<unsigned long long*>(a.lwe_out + (size_t)b * a.g.out_size + col), (unsigned long long)v);
    }
}

}  // namespace fhe
