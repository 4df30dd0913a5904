// lwe_kernels.hip.h -- batched LWE keyswitch and LWE linear combinations for gfx950.
//
//   keyswitch_dot4_kernel   out[b] = (0,...,0,body_b) - sum_i sum_lv digit(b,i,lv) * KSK[i][lv][:]
//                      Replaces keyswitch_lwe_ciphertext (core_crypto/algorithms/lwe_keyswitch.rs:96-170),
//                      SignedDecomposer::decompose (commons/math/decomposition/decomposer.rs:98-152,
//                      iter.rs:37-127) and slice_wrapping_sub_scalar_mul_assign
//                      (algorithms/slice_algorithms.rs:363-399).
//   lincomb_kernel     out[j] = sum_t coeff * in[src_t] (+ constant on the body): the shortint
//                      unchecked_add / unchecked_scalar_mul / unchecked_scalar_add / bivariate packing
//                      (shortint/server_key/add.rs:520-524, scalar_mul.rs:206-208, scalar_add.rs:211-218,
//                      bivariate_pbs.rs:167-182) applied to whole batches.
//
// Keyswitch mapping (L2-streaming integer work; the matrix-core form is ks_mfma_kernels.hip.h): a workgroup owns a tile
// of KS_COLS output columns x S samples x KS_IC input coefficients; the signed digits of the tile are produced once into
// LDS (biased to unsigned bytes) and broadcast-read, the bias is removed exactly with (B/2) * sum(KSK rows) (a property
// of the key alone, summed at load time), partial sums over input chunks are combined with 64-bit integer atomics:
// wrapping addition is associative and commutative, hence the result is deterministic and bit-exact.  (The first
// generation of this kernel -- two v_mad_u64_u32 per element on the 64-bit key layout -- was retired in round 4.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe {

constexpr int KS_COLS = 256;  // threads per workgroup == output columns per tile
constexpr int KS_IC = 64;     // input coefficients per tile

// ------------------------------------------------------------------------------------------------
// Byte planes.  A 64-bit multiply-accumulate issues two v_mad_u64_u32 per (sample, KSK element) and is bound by
// that instruction.  Exact reformulation with 8-bit dot products:
//   KSK[r][col] = sum_t byte_t(r, col) << 8t,      digit' in [0, 2^b] fits a byte as well, so
//   sum_r KSK[r][col] * d'[r] = sum_t ( sum_r byte_t(r, col) * d'[r] ) << 8t   (mod 2^64)
// and the inner sums over 4 consecutive rows are one v_dot4_u32_u8 each (u32 accumulators cannot
// overflow: 255 * 128 * rows-per-tile < 2^32).  Two dot4 per element instead of two 64-bit mads;
// every step is integer ring arithmetic, so the result stays bit-identical to the reference loop.
// The key is repacked once at load time:  packed[r/4][t][col] = bytes t of rows r..r+3 (u32).
__global__ void __launch_bounds__(256) ksk_pack_kernel(const uint64_t* __restrict__ ksk, uint32_t* __restrict__ packed,
                                                       uint32_t rows, uint32_t out_size) {
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r4 = blockIdx.y;
    if (col >= out_size) return;
    uint64_t v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = (r4 * 4 + q) < rows ? ksk[(size_t)(r4 * 4 + q) * out_size + col] : 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) w |= (uint32_t)((v[q] >> (8 * t)) & 0xff) << (8 * q);
        packed[((size_t)r4 * 8 + t) * out_size + col] = w;
    }
}

// The bias removal needs (beta/2) * sum over the tile's rows of KSK[row][col]: a property of the key alone, summed once
// at load time instead of eight more dot products per four rows in every launch.
__global__ void __launch_bounds__(256) ksk_rowsum_kernel(const uint64_t* __restrict__ ksk, uint64_t* __restrict__ rowsum,
                                                         uint32_t rows, uint32_t out_size, uint32_t rows_per_tile) {
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    if (col >= out_size) return;
    const uint32_t r0 = blockIdx.y * rows_per_tile;
    uint64_t sum = 0;
    for (uint32_t r = r0; r < r0 + rows_per_tile && r < rows; r++) sum += ksk[(size_t)r * out_size + col];
    rowsum[(size_t)blockIdx.y * out_size + col] = sum;
}

constexpr int KSD_S = 8;      // samples per tile of the byte-plane kernel (64 u32 accumulators per thread)

struct KeyswitchPackedArgs {
    const uint64_t* lwe_in;    // [B][in_dim+1]
    const uint32_t* packed;    // [in_dim*level/4][8][out_size]
    const uint64_t* rowsum;    // [in_dim / KS_IC][out_size]: sum of the tile's key rows (ksk_rowsum_kernel)
    uint64_t* lwe_out;         // [B][out_size], zero-filled before launch
    uint32_t in_dim, out_size, base_log, level, batch;
};

// KSD_S_ = 8 / MINW = 2: the stand-alone kernel.  KSD_S_ = 4 / MINW = 7 (<= 72 VGPRs): a wave of it fits next to the two
// 220-VGPR waves the blind rotation keeps on every SIMD, so the keyswitch of batch k+1 can run in the shadow of the
// blind rotation of batch k (Engine::ks_pbs_dev, pipelined mode).
template <int KSD_S_, int MINW>
__global__ void __launch_bounds__(KS_COLS, MINW) keyswitch_dot4_kernel(KeyswitchPackedArgs a) {
    constexpr int KSD_S = KSD_S_;
    extern __shared__ __align__(16) unsigned char ks_smem[];   // [rows/4][KSD_S] u32: 4 biased digits each
    uint32_t* dig = reinterpret_cast<uint32_t*>(ks_smem);
    const uint32_t col = blockIdx.x * KS_COLS + threadIdx.x;
    const uint32_t b0 = blockIdx.y * KSD_S;
    const uint32_t i0 = blockIdx.z * KS_IC;
    const uint32_t L = a.level, bl = a.base_log;
    const uint32_t half = 1u << (bl - 1);
    const uint32_t rows = KS_IC * L;                            // multiple of 4 (KS_IC is)

    for (uint32_t e = threadIdx.x; e < (uint32_t)(KS_IC * KSD_S); e += KS_COLS) {
        const uint32_t il = e / KSD_S, s = e % KSD_S;
        const uint32_t i = i0 + il, b = b0 + s;
        uint64_t x = 0;
        if (i < a.in_dim && b < a.batch) x = a.lwe_in[(size_t)b * (a.in_dim + 1) + i];
        const uint32_t rep = bl * L;
        uint64_t t = x >> (63 - rep);
        uint64_t state = ((t + 1) >> 1) & ((1ull << rep) - 1);
        const uint64_t mask = (1ull << bl) - 1;
        unsigned char* bytes = ks_smem;
        for (uint32_t lv = 0; lv < L; lv++) {
            uint64_t res = state & mask;
            state >>= bl;
            uint64_t carry = ((res - 1ull) | state) & res;
            carry >>= bl - 1;
            state += carry;
            const int32_t digit = (int32_t)(uint32_t)res - (int32_t)((uint32_t)carry << bl);
            const uint32_t r = il * L + lv;
            bytes[((size_t)(r >> 2) * KSD_S + s) * 4 + (r & 3)] = (unsigned char)(digit + (int32_t)half);
        }
    }
    __syncthreads();

    uint32_t acc[KSD_S][8];
#pragma unroll
    for (int s = 0; s < KSD_S; s++)
#pragma unroll
        for (int t = 0; t < 8; t++) acc[s][t] = 0;
    const bool col_ok = col < a.out_size;
    const uint32_t ccol = col_ok ? col : 0;
    const uint32_t* kp = a.packed + ((size_t)(i0 * L) / 4) * 8 * a.out_size + ccol;
    for (uint32_t r4 = 0; r4 < rows / 4; r4++) {
        uint32_t kb[8];
#pragma unroll
        for (int t = 0; t < 8; t++) kb[t] = kp[((size_t)r4 * 8 + t) * a.out_size];
        uint32_t dw[KSD_S];
#pragma unroll
        for (int q = 0; q < KSD_S / 4; q++) {
            const uint4 dq = *reinterpret_cast<const uint4*>(dig + (size_t)r4 * KSD_S + q * 4);
            dw[4 * q + 0] = dq.x; dw[4 * q + 1] = dq.y; dw[4 * q + 2] = dq.z; dw[4 * q + 3] = dq.w;
        }
#pragma unroll
        for (int t = 0; t < 8; t++) {
#pragma unroll
            for (int s = 0; s < KSD_S; s++) acc[s][t] = __builtin_amdgcn_udot4(kb[t], dw[s], acc[s][t], false);
        }
    }
    if (!col_ok) return;
    const uint64_t corr = a.rowsum[(size_t)blockIdx.z * a.out_size + col] * (uint64_t)half;
#pragma unroll
    for (int s = 0; s < KSD_S; s++) {
        const uint32_t b = b0 + s;
        if (b >= a.batch) break;
        uint64_t p = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) p += (uint64_t)acc[s][t] << (8 * t);
        uint64_t v = corr - p;
        if (blockIdx.z == 0 && col == a.out_size - 1)
            v += a.lwe_in[(size_t)b * (a.in_dim + 1) + a.in_dim];
        atomicAdd(reinterpret_cast<unsigned long long*>(a.lwe_out + (size_t)b * a.out_size + col),
                  (unsigned long long)v);
    }
}

// out[j][:] = sum_{t in [off[j], off[j+1])} coeff[t] * pool[src[t]][:]  ;  out[j][body] += cst[j]
struct LincombArgs {
    const uint64_t* pool;      // [*][size]
    const uint32_t* off;       // [jobs+1]
    const uint32_t* src;       // [terms]
    const int32_t* coeff;      // [terms]
    const uint64_t* cst;       // [jobs] added to the body (already scaled by delta)
    uint64_t* out;             // [jobs][size]
    uint32_t size, jobs;
};

__global__ void __launch_bounds__(256) lincomb_kernel(LincombArgs a) {
    const uint32_t j = blockIdx.x;
    const uint32_t t0 = a.off[j], t1 = a.off[j + 1];
    for (uint32_t e = threadIdx.x; e < a.size; e += blockDim.x) {
        uint64_t v = 0;
        for (uint32_t t = t0; t < t1; t++)
            v += a.pool[(size_t)a.src[t] * a.size + e] * (uint64_t)(int64_t)a.coeff[t];
        if (e == a.size - 1) v += a.cst[j];
        a.out[(size_t)j * a.size + e] = v;
    }
}

// The same gather for `instances` independent copies of a plan at once (fhe_plan_run_batch): workgroup (j, i) combines
// the sources of job j in instance i's pool.  Ciphertext (slot s, instance i) sits at (s * src_slot + i * src_inst) rows,
// the result of (job j, instance i) at (j * out_job + i * out_inst) rows -- slot-major pools [slot][instance] make the
// outputs of one level, ordered (job, instance), ONE contiguous keyswitch / blind-rotation batch; the strides also
// describe the caller's instance-major input and output arrays.  lut_out (optional): the job's table id, per output row.
struct LincombBatchArgs {
    LincombArgs base;
    uint32_t instances;
    uint32_t src_slot, src_inst, out_job, out_inst;
    const uint32_t* lut_in;    // [jobs] or nullptr
    uint32_t* lut_out;         // [jobs * instances] in output row order
};

__global__ void __launch_bounds__(256) lincomb_batch_kernel(LincombBatchArgs b) {
    const LincombArgs& a = b.base;
    const uint32_t j = blockIdx.x / b.instances, i = blockIdx.x % b.instances;
    const uint32_t t0 = a.off[j], t1 = a.off[j + 1];
    const size_t row = (size_t)j * b.out_job + (size_t)i * b.out_inst;
    for (uint32_t e = threadIdx.x; e < a.size; e += blockDim.x) {
        uint64_t v = 0;
        for (uint32_t t = t0; t < t1; t++)
            v += a.pool[((size_t)a.src[t] * b.src_slot + (size_t)i * b.src_inst) * a.size + e] * (uint64_t)(int64_t)a.coeff[t];
        if (e == a.size - 1) v += a.cst[j];
        a.out[row * a.size + e] = v;
    }
    if (b.lut_out && threadIdx.x == 0) b.lut_out[row] = b.lut_in[j];
}

// out[(s * out_slot + i * out_inst)] = in[(s * in_slot + i * in_inst)]: instance-major caller arrays <-> slot-major pools
__global__ void __launch_bounds__(256) lwe_restride_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t size,
                                                           uint32_t instances, uint32_t in_slot, uint32_t in_inst, uint32_t out_slot,
                                                           uint32_t out_inst) {
    const uint32_t s = blockIdx.x / instances, i = blockIdx.x % instances;
    const uint64_t* src = in + ((size_t)s * in_slot + (size_t)i * in_inst) * size;
    uint64_t* dst = out + ((size_t)s * out_slot + (size_t)i * out_inst) * size;
    for (uint32_t e = threadIdx.x; e < size; e += blockDim.x) dst[e] = src[e];
}

}  // namespace fhe
