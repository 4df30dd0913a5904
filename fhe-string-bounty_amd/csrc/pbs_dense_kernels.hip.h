// pbs_dense_kernels.hip.h -- blind rotation for N = 1024 with k = 2 (PARAM_MESSAGE_2_CARRY_1_KS_PBS, 1_CARRY_2, 3_CARRY_0:
// the other polynomial size BASELINE.json's north_star names), batches beyond two LWEs per CU.  Round 4.
//
// Same algorithm as blind_rotate_wide_kernel (bootstrap.rs:242-364, ggsw.rs:477-598; one workgroup = one LWE, every thread
// carries its four points of all k + 1 = 3 polynomials), laid out for FOUR workgroups per CU:
//   * that kernel keeps a set of exchange planes per polynomial, every twiddle and the accumulator in registers: 52 KB of
//     LDS and more than 256 VGPRs per 128-thread workgroup -- two workgroups per CU, one wave per SIMD, 104.7 k PBS/s
//     however large the batch;
//   * here: ONE plane set, the polynomials go through their transforms one after the other (digits of polynomial p are
//     gathered and decomposed just before its transform, its GGSW row multiplied right after), the accumulator lives in
//     LDS only (a thread re-reads its own 8 coefficients per polynomial at the gather and at the update: 2 x 24 KB of LDS
//     traffic per step, 48 VGPRs saved), at most 256 VGPRs: 36 KB per workgroup, two waves on every SIMD;
//   * transforms on FftSwap9 (negacyclic_fft.hip.h): two of the four inter-pass exchanges are register/lane swaps and the
//     twiddles sit in registers.  The first dense version ran the generic plan (every exchange and a twiddle table through
//     LDS, 550 KB per workgroup-step): 133-138 k PBS/s; this one 137-146 k, 142-151 k with the inverse twiddles folded,
//     145-162 k with the two sides pipelined by hand (same box: +2 %).  VALU issue is ~55 % of the time: what is left is
//     the dependent chain of one polynomial's transform at a time (the wide kernel interleaves its polynomials stage by
//     stage, which needs a plane set each); ablation builds show that neither the barriers, the gather nor the key loads
//     are it (profiles/r04_n1024.txt).
// The Fourier key is read in FftSwap9's order: a second copy of the key (54.7 MB) made by bsk_convert_dense_kernel.
#pragma once
#include "pbs_kernels.hip.h"

namespace fhe {

template <int LOGN, int K1>
struct BrDenseCfg {
    static_assert(LOGN == 10, "dense layout: N = 1024 (FftSwap9)");
    using PL = FftSwap9;
    static constexpr int N = 1 << LOGN, P = N / 2, R = PL::R, T = PL::T, THREADS = T;
    static constexpr int PLANE = P + 2;
    static constexpr int GROUP_SLOTS = 2 * P + 4;
    static constexpr size_t LDS_FIXED = (size_t)K1 * N * 8 /*acc*/ + (size_t)GROUP_SLOTS * 8 /*planes*/;     // + 4 n (mask)
};

template <int LOGN, int K1>
__global__ void __launch_bounds__((BrDenseCfg<LOGN, K1>::THREADS), 2)
blind_rotate_dense_kernel(BlindRotateArgs args) {
    using CFG = BrDenseCfg<LOGN, K1>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* lds_acc = reinterpret_cast<uint64_t*>(smem);                       // [K1][N], PL::acc_slot order
    double* lds_x = reinterpret_cast<double*>(smem + (size_t)K1 * N * 8);        // one plane set
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(lds_x + CFG::GROUP_SLOTS);     // [n]

    const int tau = threadIdx.x;
    const uint32_t sample = blockIdx.x;
    const uint32_t n = args.n;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    const uint32_t acc_address = lds_address(lds_acc);     // 8N-aligned: Rotation::source_bytes ORs offsets onto it
    if (acc_address & (8u * N - 1u)) __builtin_trap();
    const uint32_t bL = args.base_log;                     // one decomposition level
    const uint32_t dbias = decomp_bias_constant(bL <= 31 ? bL : 31);

    for (uint32_t i = threadIdx.x; i < n; i += CFG::THREADS) {
        const uint64_t a = lwe[i];
        lds_d[i] = a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN);
    }

    FftConsts<PL> fc;
    fft_init_consts<PL>(fc, tau);
    cplx twist[R], twbias[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        double sn, cs;
        sincospi((double)PL::point(tau, m) / (double)N, &sn, &cs);
        twist[m].re = cs; twist[m].im = sn;
        const double cb = -(double)((1u << (args.base_log - 1)) - 1u);
        twbias[m].re = cb * (cs - sn);
        twbias[m].im = cb * (cs + sn);
    }

    // this thread's own coefficients j = PL::point(tau, m) + h P of polynomial p: slot (j & 1) * 512 + (j >> 1) =
    // own_base + 64 m + 256 h (a constant offset on one address)
    uint64_t* own = lds_acc + PL::acc_slot(PL::point(tau, 0));
    auto own_slot = [](int p, int m, int h) { return p * N + PL::acc_slot(PL::point(0, m) + h * P); };
    static_assert(PL::acc_slot(PL::point(5, 2) + P) == PL::acc_slot(PL::point(5, 0)) + PL::acc_slot(PL::point(0, 2) + P), "own slots: base + constant");

    // acc <- LUT * X^{-ms(body)}   (bootstrap.rs:254-271, polynomial_algorithms.rs:331-353)
    {
        const uint32_t d = modulus_switch(lwe[n], LOGN);
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
#pragma unroll
        for (int p = 0; p < K1; p++)
#pragma unroll
            for (int m = 0; m < R; m++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t j = PL::point(tau, m) + h * P;
                    const uint32_t src = (j + rem) & (N - 1);
                    const bool neg = ((j + rem) >= (uint32_t)N) != odd;
                    uint64_t v = lut[(size_t)p * N + src];
                    own[own_slot(p, m, h)] = neg ? (0 - v) : v;
                }
    }
    __syncthreads();

    constexpr size_t GGSW_ELEMS = (size_t)K1 * K1 * P;
    const uint32_t key_off = (uint32_t)tau * 16u;
    const auto key_rsrc = key_resource(args.fbsk, (size_t)n * GGSW_ELEMS * 16);

    uint32_t d_next = lds_d[0];
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_next);   // workgroup-uniform: scalar
        d_next = lds_d[i + 1 < n ? i + 1 : i];
        if (d == 0xFFFFFFFFu) continue;                                             // a_i == 0 (bootstrap.rs:281)
        const Rotation<PL, LOGN> rot(d, tau);

        double2 brow[K1][R];        // one GGSW row: K1 columns x R points of this thread
        auto request_row = [&](int row) {
#pragma unroll
            for (int col = 0; col < K1; col++)
#pragma unroll
                for (int rho = 0; rho < R; rho++)
#ifdef FHESTR_ABLATE_DENSE_KEY
                    brow[col][rho] = make_double2(1.0 + row + col, 0.5 + rho + (double)i);
#else
                    brow[col][rho] = key_load(key_rsrc, key_off, (uint32_t)((i * GGSW_ELEMS + ((size_t)row * K1 + col) * P + rho * T) * 16));
#endif
        };
        cplx outf[K1][R];
        // products of one polynomial's spectrum with its GGSW row (ggsw.rs:560-598)
        auto multiply_row = [&](int row, const cplx* f4) {
#pragma unroll
            for (int col = 0; col < K1; col++)
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 bv = brow[col][rho];
                    const cplx f = f4[rho];
                    if (row == 0) {
                        outf[col][rho].re = bv.x * f.re - bv.y * f.im;
                        outf[col][rho].im = bv.x * f.im + bv.y * f.re;
                    } else {
                        outf[col][rho].re = fma(bv.x, f.re, fma(-bv.y, f.im, outf[col][rho].re));
                        outf[col][rho].im = fma(bv.x, f.im, fma(bv.y, f.re, outf[col][rho].im));
                    }
                }
        };
        double* const xre = lds_x;
        double* const xim = lds_x + CFG::PLANE;

        // Forward side, software-pipelined by hand (each wave runs ONE dependent chain at a time here, so whatever is
        // independent is placed where the chain waits): polynomial `row`'s products are issued behind the LDS stores of
        // polynomial row + 1's first stage; its GGSW row is requested before the barrier of its own transform.
        cplx spec[R];               // the previous polynomial's spectrum
#pragma unroll
        for (int row = 0; row < K1; row++) {
            // ct1 = acc * X^d - acc of polynomial `row` (polynomial_algorithms.rs:463-489), decomposed, twisted
            uint32_t row_base = (acc_address + (uint32_t)row * 8u * N) | rot.rbits8;
            asm volatile("" : "+s"(row_base));
            cplx xr[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                uint32_t st[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t address, sm32;
                    rot.source_bytes(m, h, row_base, address, sm32);
#ifdef FHESTR_ABLATE_DENSE_GATHER
                    const uint64_t gathered = (uint64_t)address * 0x9E3779B97F4A7C15ull;
#else
                    const uint64_t gathered = lds_load_u64(address);
#endif
                    const uint64_t sm = ((uint64_t)sm32 << 32) | sm32;
                    const uint64_t v = (gathered ^ sm) - sm;
                    st[h] = decomp_single_biased(v - own[own_slot(row, m, h)], bL, dbias);
                }
                xr[m] = digit_point(st[0], st[1], twist[m], twbias[m]);      // fft/mod.rs:220-239
            }
            swap10_fwd_stage1(xr, fc, xre, xim, tau);
            if (row > 0) multiply_row(row - 1, spec);
            wave_local_fence();
            swap10_fwd_stage2(xr, fc, xre, xim, tau);
            wave_local_fence();          // the slab stores below reuse the rows the exchange above read
            swap10_fwd_stage3(xr, fc, xre, xim, tau);
            request_row(row);
            FHE_DENSE_SYNC();
            swap9_forward_tail(xr, xre, xim, tau);
            FHE_DENSE_SYNC();            // the next polynomial's first stores vs the other wave's reads of this one's last pass
#pragma unroll
            for (int rho = 0; rho < R; rho++) spec[rho] = xr[rho];
        }
        multiply_row(K1 - 1, spec);

        // every gather of this step is done (the barrier after the last forward transform): the accumulator may change.
        // Inverse side: polynomial p - 1's accumulator update goes behind polynomial p's first stores.
        auto update_acc = [&](int p) {
#pragma unroll
            for (int m = 0; m < R; m++) {
                const cplx t = cmul_conj(outf[p][m], twist[m]);
                own[own_slot(p, m, 0)] += from_torus(t.re);
                own[own_slot(p, m, 1)] += from_torus(t.im);
            }
        };
#pragma unroll
        for (int p = 0; p < K1; p++) {
            swap9_inverse_head(outf[p], xre, xim, tau);
            if (p > 0) update_acc(p - 1);
            FHE_DENSE_SYNC();
            swap9_inverse_tail(outf[p], fc, xre, xim, tau);
            if (p + 1 < K1) FHE_DENSE_SYNC();      // the next inverse's first stores land in the other wave's rows
        }
        update_acc(K1 - 1);
        FHE_DENSE_SYNC();
    }

    // sample extraction (glwe_sample_extraction.rs:91-147)
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
    for (int p = 0; p < K1; p++)
#pragma unroll
        for (int m = 0; m < R; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t j = PL::point(tau, m) + h * P;
                const uint64_t v = own[own_slot(p, m, h)];
                if (p == K1 - 1) {
                    if (j == 0) out[(size_t)(K1 - 1) * N] = v;
                } else {
                    if (j == 0) out[(size_t)p * N] = v;
                    else out[(size_t)p * N + (N - j)] = 0 - v;
                }
            }
}

// Standard-domain polynomials -> the dense kernel's Fourier layout (FftSwap9's order).  One polynomial per workgroup; the
// arithmetic of bsk_convert_kernel (forward_as_torus, fft/mod.rs:197-218; the inverse's 1/(N/2) folded in).
template <int LOGN, int K1>
__global__ void __launch_bounds__((BrDenseCfg<LOGN, K1>::THREADS))
bsk_convert_dense_kernel(const uint64_t* __restrict__ bsk_std, double* __restrict__ fbsk, uint32_t n_polys) {
    using CFG = BrDenseCfg<LOGN, K1>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    __shared__ __align__(16) double planes[CFG::GROUP_SLOTS];
    const int tau = threadIdx.x;
    const uint32_t poly = blockIdx.x;
    if (poly >= n_polys) return;             // whole workgroup
    FftConsts<PL> fc;
    fft_init_consts<PL>(fc, tau);
    cplx x[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int j = PL::point(tau, m);
        const uint64_t a = bsk_std[(size_t)poly * N + j];
        const uint64_t b = bsk_std[(size_t)poly * N + j + P];
        cplx z;
        z.re = i64_to_f64(a) * (5.421010862427522e-20 / P);
        z.im = i64_to_f64(b) * (5.421010862427522e-20 / P);
        double sn, cs;
        sincospi((double)j / (double)N, &sn, &cs);
        cplx w; w.re = cs; w.im = sn;
        x[m] = cmul(z, w);
    }
    fft_forward<PL>(x, fc, planes, planes + CFG::PLANE, tau);
    double2* out = reinterpret_cast<double2*>(fbsk) + (size_t)poly * P;
#pragma unroll
    for (int rho = 0; rho < R; rho++) out[rho * T + tau] = make_double2(x[rho].re, x[rho].im);
}

}  // namespace fhe
