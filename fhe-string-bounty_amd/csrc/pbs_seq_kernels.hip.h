// pbs_seq_kernels.hip.h -- blind rotation for N = 8192 (PARAM_MESSAGE_{1_5,2_4,3_3,4_2,5_1,6_0}_KS_PBS and the
// 3_3 multi-bit sets, shortint/parameters/mod.rs:793-882, multi_bit.rs:134-152,192-209): the polynomial size
// between "everything in registers and LDS" (N <= 4096, pbs_kernels.hip.h) and "four-step FFT through an HBM
// workspace" (N >= 16384, pbs_large_kernels.hip.h).
//
// Same algorithm (fft64/crypto/bootstrap.rs:242-364, ggsw.rs:477-598).  One workgroup of T = N/2/8 = 512 threads
// per LWE.  What fits on the CU and what does not:
//   * ONE polynomial's spectrum (N/2 complex = 64 KB) fits the LDS exchange planes, so all 512 threads run each
//     size-N/2 transform together, in place, radix 8 (FftPlan<12, 3>: four passes, three exchanges) -- and the
//     l (k+1) digit polynomials of a step go through one after the other;
//   * the (k+1) output spectra being accumulated (ggsw.rs:616-697) stay in VGPRs, 8 slots x (k+1) per thread;
//   * the accumulator ((k+1) N u64 = 128 KB) does NOT fit next to that: its first polynomial (64 KB) lives in
//     LDS, the others in a per-LWE workspace in HBM (16 MB for 256 LWEs: L2 / Infinity-Cache resident), each
//     read twice (rotated and in place) and updated once per step with coalesced 8-byte accesses: 0.25 MB per
//     step and LWE where the four-step kernel moves 1.25 MB.  (First version, whole accumulator in the
//     workspace: rocprofv3 showed 146 GB per 256-LWE launch at 3.6 TB/s, VALU 17 % busy -- memory-bound.)
// Twiddles: pass 0 in VGPRs, passes 1-2 from an LDS table (FftHybridConsts).  Fourier key: [i][level][row][col]
// polynomials in this plan's output order (slot rho * T + tau), converted by bsk_convert_seq_kernel.
// EXTPROD: multi-bit two-kernel path, see blind_rotate_kernel.
#pragma once
#include "pbs_large_kernels.hip.h"

namespace fhe {

template <int LOGN, int K1, int L>
struct BrSeqCfg {
    static constexpr int N = 1 << LOGN, P = N / 2;
    using PL = FftPlan<LOGN - 1, 3>;
    using TW = FftHybridConsts<PL>;
    static constexpr int R = PL::R, T = PL::T, THREADS = T;
    static constexpr int PLANE = P + 2;               // see BrCfg: keeps re/im pairs from being fused into st64 accesses
    static constexpr size_t LDS_PLANES = (size_t)2 * PLANE * 8;
    static constexpr size_t LDS_TW = (size_t)TW::ENTRIES * 16;
    static constexpr int LDS_POLYS = 1;                                // accumulator polynomials kept in LDS
    static constexpr size_t LDS_ACC = (size_t)LDS_POLYS * N * 8;
    static constexpr size_t LDS_CONVERT = LDS_PLANES + LDS_TW;
    static constexpr size_t LDS_FIXED = LDS_PLANES + LDS_TW + LDS_ACC; // + 4 n for the modulus-switched mask
    static constexpr size_t WS_BYTES = (size_t)(K1 - LDS_POLYS) * N * 8;   // the other polynomials, natural coefficient order
};

template <int LOGN, int K1, int L>
__global__ void __launch_bounds__((BrSeqCfg<LOGN, K1, L>::THREADS))
bsk_convert_seq_kernel(const uint64_t* __restrict__ bsk_std, double* __restrict__ fbsk, uint32_t n_polys,
                       double2* /* no workspace: same signature as bsk_convert_large_kernel */) {
    using CFG = BrSeqCfg<LOGN, K1, L>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    double* re = reinterpret_cast<double*>(smem);
    double* im = re + CFG::PLANE;
    double2* lds_tw = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES);
    const int tau = threadIdx.x;
    typename CFG::TW fc;
    CFG::TW::fill(lds_tw, tau, CFG::THREADS);
    fc.init(lds_tw, tau);
    __syncthreads();
    for (uint32_t poly = blockIdx.x; poly < n_polys; poly += gridDim.x) {
        cplx x[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int j = PL::point(tau, m);
            // forward_as_torus (fft/mod.rs:197-218) with the inverse transform's 1/(N/2) folded in, as bsk_convert_kernel
            cplx z;
            z.re = i64_to_f64(bsk_std[(size_t)poly * N + j]) * (5.421010862427522e-20 / P);
            z.im = i64_to_f64(bsk_std[(size_t)poly * N + j + P]) * (5.421010862427522e-20 / P);
            double sn, cs;
            sincospi((double)j / (double)N, &sn, &cs);
            cplx w; w.re = cs; w.im = sn;
            x[m] = cmul(z, w);
        }
        fft_forward<PL>(x, fc, re, im, tau);
        double2* out = reinterpret_cast<double2*>(fbsk) + (size_t)poly * P;
#pragma unroll
        for (int rho = 0; rho < R; rho++) out[rho * T + tau] = make_double2(x[rho].re, x[rho].im);
        __syncthreads();      // planes are reused by the next polynomial
    }
}

template <int LOGN, int K1, int L, bool EXTPROD = false>
__global__ void __launch_bounds__((BrSeqCfg<LOGN, K1, L>::THREADS))
blind_rotate_seq_kernel(BlindRotateLargeArgs la) {
    using CFG = BrSeqCfg<LOGN, K1, L>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    const BlindRotateArgs& args = la.base;
    extern __shared__ __align__(16) unsigned char smem[];
    double* re = reinterpret_cast<double*>(smem);
    double* im = re + CFG::PLANE;
    double2* lds_tw = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES);
    uint64_t* lds_acc = reinterpret_cast<uint64_t*>(smem + CFG::LDS_PLANES + CFG::LDS_TW);   // [LDS_POLYS][N]
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(smem + CFG::LDS_FIXED);      // [n]

    const int tau = threadIdx.x;
    const uint32_t sample = blockIdx.x;
    const uint32_t n = args.n;
    const uint32_t steps = EXTPROD ? n / args.grouping : n;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    uint64_t* ws_acc = reinterpret_cast<uint64_t*>(la.workspace + (size_t)sample * CFG::WS_BYTES);   // [K1 - LDS_POLYS][N]
    // Polynomial p of the accumulator (p is a compile-time constant wherever these are called in the step): LDS, or
    // the workspace through a raw buffer resource -- byte offset = one VGPR + a scalar, no 64-bit address pairs
    // (the first version spent ~60 spilled dwords per step on precomputed global addresses).
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const auto ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws_acc, 0, (int)(CFG::WS_BYTES > 0 ? CFG::WS_BYTES : 8), 0x00020000);
    auto acc_load = [&](int p, uint32_t voff, uint32_t soff) -> uint64_t {
        if (p < CFG::LDS_POLYS) return lds_acc[(size_t)p * N + ((voff + soff) >> 3)];
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(ws_rsrc, (int)voff, (int)(soff + (uint32_t)(p - CFG::LDS_POLYS) * N * 8u), 0);
        return ((uint64_t)v.y << 32) | v.x;
    };
    auto acc_store = [&](int p, uint32_t voff, uint32_t soff, uint64_t val) {
        if (p < CFG::LDS_POLYS) { lds_acc[(size_t)p * N + ((voff + soff) >> 3)] = val; return; }
        u32x2_t v;
        v.x = (uint32_t)val; v.y = (uint32_t)(val >> 32);
        __builtin_amdgcn_raw_buffer_store_b64(v, ws_rsrc, (int)voff, (int)(soff + (uint32_t)(p - CFG::LDS_POLYS) * N * 8u), 0);
    };
    auto acc_poly = [&](int p) -> uint64_t* {
        return p < CFG::LDS_POLYS ? lds_acc + (size_t)p * N : ws_acc + (size_t)(p - CFG::LDS_POLYS) * N;
    };
    const uint32_t tau8 = (uint32_t)tau * 8u;
    const uint32_t bL = args.base_log * L;

    for (uint32_t i = threadIdx.x; i < steps; i += CFG::THREADS) {
        const uint64_t a = lwe[i];
        lds_d[i] = EXTPROD ? 0u : (a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN));
    }
    typename CFG::TW fc;
    CFG::TW::fill(lds_tw, tau, CFG::THREADS);
    fc.init(lds_tw, tau);
    // twisty of point tau + T m: e^{i pi tau / N} * e^{i pi m T / N}; the second factor is a compile-time constant, so
    // only the first is kept (2 VGPR pairs instead of 2 R) and the product is formed where it is used -- the
    // transforms need the registers (opaque per step, or the compiler hoists all R products back out of the loop)
    cplx twist0;
    {
        double sn, cs;
        sincospi((double)tau / (double)N, &sn, &cs);
        twist0.re = cs; twist0.im = sn;
    }
    auto twist_of = [&](int m) -> cplx {
        constexpr double PI = 3.14159265358979323846;
        cplx c;
        c.re = __builtin_cos(PI * (double)(m * T) / (double)N);
        c.im = __builtin_sin(PI * (double)(m * T) / (double)N);
        return m == 0 ? twist0 : cmul(twist0, c);
    };

    // acc <- LUT * X^{-ms(body)}   (bootstrap.rs:254-271)
    {
        const uint32_t d = modulus_switch(lwe[n], LOGN);
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
        for (int e = tau; e < K1 * N; e += CFG::THREADS) {
            const uint32_t p = e >> LOGN, j = e & (N - 1);
            const uint32_t src = (j + rem) & (N - 1);
            const bool neg = ((j + rem) >= (uint32_t)N) != odd;
            const uint64_t v = lut[(size_t)p * N + src];
            acc_poly((int)p)[j] = neg ? (0 - v) : v;
        }
    }
    __syncthreads();

    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)L * K1 * K1 * P;
    if constexpr (EXTPROD) fbsk += (size_t)sample * steps * GGSW_ELEMS;
    const auto key_rsrc = key_resource(reinterpret_cast<const double*>(fbsk), (size_t)steps * GGSW_ELEMS * 16);

    uint32_t d_next = lds_d[0];
    for (uint32_t i = 0; i < steps; i++) {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_next);
        d_next = lds_d[i + 1 < steps ? i + 1 : i];
        if (d == 0xFFFFFFFFu) continue;                          // a_i == 0 (bootstrap.rs:281), workgroup-uniform
        asm volatile("" : "+v"(twist0.re), "+v"(twist0.im));
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
        cplx outf[K1][R];
#pragma unroll
        for (int r = 0; r < K1; r++) {
            // ct1 = acc_r * X^d - acc_r (polynomial_algorithms.rs:463-489), decomposition state per coefficient
            using state_t = typename std::conditional<(L >= 3), uint64_t, uint32_t>::type;
            state_t st_lo[R], st_hi[R];
            const uint32_t rem8 = rem * 8u;
#pragma unroll
            for (int m = 0; m < R; m++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t c8 = (uint32_t)(m * T + h * P) * 8u;           // j = tau + T m + h P
                    const uint32_t j = (uint32_t)PL::point(tau, m) + h * P;
                    uint64_t ct1 = acc_load(r, tau8, c8);
                    if constexpr (!EXTPROD) {
                        uint64_t v = acc_load(r, (tau8 + c8 - rem8) & (8u * N - 1u), 0);
                        v = ((j < rem) != odd) ? (0 - v) : v;
                        ct1 = v - ct1;
                    }
                    state_t st;
                    if constexpr (L >= 3) st = decomp_init_state64(ct1, bL);
                    else st = decomp_init_state(ct1, bL);
                    if (h == 0) st_lo[m] = st; else st_hi[m] = st;
                }
            }
            FHE_PIN_ORDER();      // the key requests below stay below: 64 VGPRs that the accumulator loads above need
#pragma unroll
            for (int it = 0; it < L; it++) {
                const int lvl_idx = L - 1 - it;                  // ggsw.rs:524 (levels reversed)
                // this (level, row)'s K1 key polynomials: requested now, used after the transform
                double2 bv[K1][R];
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int rho = 0; rho < R; rho++)
                        bv[col][rho] = key_load(key_rsrc, tau8 * 2u,
                                                (uint32_t)((i * GGSW_ELEMS + (((size_t)lvl_idx * K1 + r) * K1 + col) * P + rho * T) * 16));
                cplx x[R];
#pragma unroll
                for (int m = 0; m < R; m++) {
                    cplx z;
                    if constexpr (L >= 3) {
                        z.re = (double)decomp_next_digit64(st_lo[m], args.base_log);
                        z.im = (double)decomp_next_digit64(st_hi[m], args.base_log);
                    } else {
                        z.re = (double)decomp_next_digit(st_lo[m], args.base_log);
                        z.im = (double)decomp_next_digit(st_hi[m], args.base_log);
                    }
                    x[m] = cmul(z, twist_of(m));                 // fft/mod.rs:220-239
                }
                if (r > 0 || it > 0) __syncthreads();            // the previous transform's last plane reads are done
                fft_forward<PL>(x, fc, re, im, tau);
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const double2 b = bv[col][rho];
                        const cplx f = x[rho];
                        if (r == 0 && it == 0) {
                            outf[col][rho].re = b.x * f.re - b.y * f.im;
                            outf[col][rho].im = b.x * f.im + b.y * f.re;
                        } else {
                            outf[col][rho].re = fma(b.x, f.re, fma(-b.y, f.im, outf[col][rho].re));
                            outf[col][rho].im = fma(b.x, f.im, fma(b.y, f.re, outf[col][rho].im));
                        }
                    }
            }
        }
        // back to the standard domain and accumulate (fft/mod.rs:285-304, 539-557; 1/(N/2) lives in the key)
#pragma unroll
        for (int col = 0; col < K1; col++) {
            uint64_t a_lo[R], a_hi[R];       // requested before the transform
#pragma unroll
            for (int m = 0; m < R; m++) {
                a_lo[m] = EXTPROD ? 0 : acc_load(col, tau8, (uint32_t)(m * T) * 8u);
                a_hi[m] = EXTPROD ? 0 : acc_load(col, tau8, (uint32_t)(m * T + P) * 8u);
            }
            __syncthreads();
            fft_inverse<PL>(outf[col], fc, re, im, tau);
#pragma unroll
            for (int m = 0; m < R; m++) {
                const cplx t = cmul_conj(outf[col][m], twist_of(m));
                acc_store(col, tau8, (uint32_t)(m * T) * 8u, a_lo[m] + from_torus(t.re));
                acc_store(col, tau8, (uint32_t)(m * T + P) * 8u, a_hi[m] + from_torus(t.im));
            }
        }
        __syncthreads();     // the accumulator update is visible to the next step's rotated reads (same CU, same L1)
    }

    // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146)
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
    for (int e = tau; e < K1 * N; e += CFG::THREADS) {
        const uint32_t p = e >> LOGN, j = e & (N - 1);
        const uint64_t v = acc_poly((int)p)[j];
        if (p == (uint32_t)K1 - 1) {
            if (j == 0) out[(size_t)(K1 - 1) * N] = v;           // body = B[0]
        } else {
            if (j == 0) out[(size_t)p * N] = v;
            else out[(size_t)p * N + (N - j)] = 0 - v;           // out[t] = -A[N - t]
        }
    }
}

}  // namespace fhe
