// pbs_multibit_kernels.hip.h -- multi-bit programmable bootstrap (grouping factor G) for gfx950.
//
// Replaces multi_bit_blind_rotate_assign / multi_bit_programmable_bootstrap_lwe_ciphertext
//   tfhe/src/core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs:18-83 (prepare_multi_bit_ggsw),
//   :295-546 (blind rotation: one external product per group of G mask elements), :1035-1127 (PBS)
// for the parameter shape of PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS (N = 2048, k = 1, one
// level; shortint/parameters/multi_bit.rs:115-135,173-190), grouping factors 2 and 3.
//
// The reference pipelines "build the group's GGSW" (CPU threads) against "external product"
// (one thread).  Here one workgroup per LWE does both inside the step, thread-locally:
//   * the Fourier key holds 2^G GGSWs per group; GGSW `sel` is multiplied by the transform of the
//     monomial X^{ms(sum of the selected mask elements)}.  In this engine's spectrum order slot
//     (rho, tau) carries the value at zeta = w^(1 - 4 f), w = e^{i pi / N}, f = f_tau + 256 rho, so the
//     monomial's transform at that slot is w^{d (1 - 4 f)} = w^{d (1 - 4 f_tau)} * (-i)^{d rho}: one
//     root-table lookup per selector per thread, the four register slots follow by quarter turns;
//   * the step is a plain external product acc <- GGSW (x) acc (no rotate-and-subtract), so the
//     accumulator is decomposed straight out of its registers: no LDS copy of the accumulator, no
//     gather, and only the two barriers of the forward / inverse transforms remain per step --
//     n/G steps instead of n.
// Thread layout, FFT plan and Fourier-key slot order are those of blind_rotate_kernel with the
// FftSwap10 plan (pbs_kernels.hip.h); the key is converted by the same bsk_convert_kernel.
//
// Small batches (far fewer LWEs than CUs) split the step the way the reference does, over CUs instead
// of CPU threads: multibit_combine_kernel builds every (LWE, group) GGSW on the whole GPU first (one
// workgroup per group and slot half keeps the group's 2^G key GGSWs in registers and walks the batch),
// then blind_rotate_multibit_kernel<..., PRE = true> runs n/G plain external products against its own
// LWE's combined GGSWs (64 KB per step, prefetched like the classic kernel's key).  Same operations in the
// same order as the fused step, so both paths give bit-identical ciphertexts.
#pragma once
#include "pbs_kernels.hip.h"

namespace fhe {

template <int LOGN, int LOGR, int K1, int G>
struct BrMultiBitCfg {
    using Base = BrCfg<LOGN, LOGR, K1, 1>;
    using PL = typename Base::PL;
    static constexpr int N = Base::N, P = Base::P, R = Base::R, T = Base::T;
    static constexpr int THREADS = Base::THREADS;
    static constexpr int SEL = (1 << G) - 1;                   // monomial-carrying GGSWs per group
    // register slots of every GGSW one combine workgroup keeps: 2^G * K1 * SLOTS c64 <= 128 VGPRs
    static constexpr int COMBINE_CHUNK = 8;    // LWEs one combine workgroup walks with the key GGSWs in registers
    static constexpr int COMBINE_SLOTS = (SEL + 1) * K1 * R * 4 <= 128 ? R : ((SEL + 1) * K1 * R * 2 <= 128 ? R / 2 : R / 4);
    static constexpr int ROOT_LO_BITS = (LOGN + 1) / 2, ROOT_HI_BITS = LOGN + 1 - ROOT_LO_BITS;
    // LDS: forward planes + inverse planes (as the classic split kernel) + two-level root table of
    // e^{i pi m / N}, m < 2N; the per-group monomial degrees ((n/G) * SEL u32) follow dynamically
    static constexpr size_t LDS_ROOTS = ((size_t)(1 << ROOT_LO_BITS) + (size_t)(1 << ROOT_HI_BITS)) * 16;
    static constexpr size_t LDS_FIXED = 2 * (size_t)K1 * Base::GROUP_SLOTS * 8 + LDS_ROOTS;
};

// frequency of register slot 0 of thread tau (FftSwap10: regs k4 | wave k3 | lane(5,4) k2 | lane(3,2) k0 |
// lane(1,0) k1, f = k0 + 4 k1 + 16 k2 + 64 k3 + 256 k4), as 1 - 4 f mod 2N: the monomial X^d has the value
// w^{d (1 - 4 f)} at that slot, w = e^{i pi / N}
template <int N>
__device__ __forceinline__ uint32_t multibit_slot_exponent(int tau) {
    const int lane = tau & 63, wv = tau >> 6;
    const uint32_t f_tau = ((lane >> 2) & 3) + 4 * (lane & 3) + 16 * (lane >> 4) + 64 * wv;
    return (1u - 4u * f_tau) & (2u * N - 1u);
}

// e^{i pi m / N} = root_lo[m & mask] * root_hi[m >> ROOT_LO_BITS], m < 2N
template <class CFG>
__device__ __forceinline__ void multibit_fill_roots(double2* root_lo, double2* root_hi) {
    for (int e = threadIdx.x; e < (1 << CFG::ROOT_LO_BITS); e += CFG::THREADS) {
        double sn, cs;
        sincospi((double)e / (double)CFG::N, &sn, &cs);
        root_lo[e] = make_double2(cs, sn);
    }
    for (int e = threadIdx.x; e < (1 << CFG::ROOT_HI_BITS); e += CFG::THREADS) {
        double sn, cs;
        sincospi((double)((size_t)e << CFG::ROOT_LO_BITS) / (double)CFG::N, &sn, &cs);
        root_hi[e] = make_double2(cs, sn);
    }
}

// value of X^d at slot 0 of a thread with slot exponent c_tau, and the quarter turn (-i)^d that leads from
// one register slot to the next
template <class CFG>
__device__ __forceinline__ void multibit_monomial(uint32_t d, uint32_t c_tau, const double2* root_lo,
                                                  const double2* root_hi, cplx& mono, cplx& turn) {
    const uint32_t mi = (d * c_tau) & (2u * CFG::N - 1u);
    const double2 a = root_lo[mi & ((1u << CFG::ROOT_LO_BITS) - 1u)], b = root_hi[mi >> CFG::ROOT_LO_BITS];
    mono.re = a.x * b.x - a.y * b.y;
    mono.im = a.x * b.y + a.y * b.x;
    const int q = d & 3;
    turn.re = q == 0 ? 1.0 : q == 2 ? -1.0 : 0.0;
    turn.im = q == 1 ? -1.0 : q == 3 ? 1.0 : 0.0;
}

struct MultiBitCombineArgs {
    BlindRotateArgs a;
    double2* combined;      // [count][n/G][K1][K1][P] c64, slot order of the Fourier key
};

// prepare_multi_bit_ggsw (:18-83) for a whole batch: grid (n/G, R / SLOTS, ceil(batch / COMBINE_CHUNK)); the
// workgroup keeps the group's 2^G GGSWs (its SLOTS register slots of them) in VGPRs and writes
// G0 + sum_sel G_sel * X^{d_sel} for its chunk of the batch's LWEs.
template <int LOGN, int LOGR, int K1, int G>
__global__ void __launch_bounds__((BrMultiBitCfg<LOGN, LOGR, K1, G>::THREADS))
multibit_combine_kernel(MultiBitCombineArgs ca) {
    using CFG = BrMultiBitCfg<LOGN, LOGR, K1, G>;
    constexpr int N = CFG::N, P = CFG::P, T = CFG::T, SEL = CFG::SEL, SLOTS = CFG::COMBINE_SLOTS;
    extern __shared__ __align__(16) unsigned char smem[];
    double2* root_lo = reinterpret_cast<double2*>(smem);
    double2* root_hi = root_lo + (1 << CFG::ROOT_LO_BITS);
    const BlindRotateArgs& args = ca.a;
    const int g = threadIdx.x / T, tau = threadIdx.x % T;
    const uint32_t grp = blockIdx.x, rho0 = blockIdx.y * SLOTS;
    const uint32_t n = args.n, groups = n / G;
    constexpr size_t GGSW_ELEMS = (size_t)K1 * K1 * P;
    multibit_fill_roots<CFG>(root_lo, root_hi);
    const uint32_t c_tau = multibit_slot_exponent<N>(tau);

    const double2* gk = reinterpret_cast<const double2*>(args.fbsk) + (size_t)grp * (SEL + 1) * GGSW_ELEMS;
    double2 gv[SEL + 1][K1][SLOTS];
#pragma unroll
    for (int s = 0; s <= SEL; s++)
#pragma unroll
        for (int r = 0; r < K1; r++)
#pragma unroll
            for (int h = 0; h < SLOTS; h++)
                gv[s][r][h] = gk[(size_t)s * GGSW_ELEMS + ((size_t)((g + r) % K1) * K1 + g) * P + (rho0 + h) * T + tau];
    __syncthreads();

    const uint32_t first = blockIdx.z * CFG::COMBINE_CHUNK;
    const uint32_t last = min(args.batch, first + CFG::COMBINE_CHUNK);
    for (uint32_t sample = first; sample < last; sample++) {
        const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1) + (size_t)grp * G;
        cplx comb[K1][SLOTS];
#pragma unroll
        for (int r = 0; r < K1; r++)
#pragma unroll
            for (int h = 0; h < SLOTS; h++) { comb[r][h].re = gv[0][r][h].x; comb[r][h].im = gv[0][r][h].y; }
#pragma unroll
        for (int s = 1; s <= SEL; s++) {
            uint64_t sum = 0;
#pragma unroll
            for (int b = 0; b < G; b++)
                if ((s >> (G - 1 - b)) & 1) sum += lwe[b];
            const uint32_t d = modulus_switch(sum, LOGN);
            cplx mono, turn;
            multibit_monomial<CFG>(d, c_tau, root_lo, root_hi, mono, turn);
            for (uint32_t q = 0; q < rho0; q++) mono = cmul(mono, turn);   // exact: entries of turn are 0 / +-1
#pragma unroll
            for (int h = 0; h < SLOTS; h++) {
#pragma unroll
                for (int r = 0; r < K1; r++) {
                    const double2 v = gv[s][r][h];
                    comb[r][h].re = fma(v.x, mono.re, fma(-v.y, mono.im, comb[r][h].re));
                    comb[r][h].im = fma(v.x, mono.im, fma(v.y, mono.re, comb[r][h].im));
                }
                mono = cmul(mono, turn);
            }
        }
        double2* out = ca.combined + ((size_t)sample * groups + grp) * GGSW_ELEMS;
#pragma unroll
        for (int r = 0; r < K1; r++)
#pragma unroll
            for (int h = 0; h < SLOTS; h++)
                out[((size_t)((g + r) % K1) * K1 + g) * P + (rho0 + h) * T + tau] = make_double2(comb[r][h].re, comb[r][h].im);
    }
}

// ---- any other shape: two-kernel path only ------------------------------------------------------
// The Fourier key of every blind-rotation variant is "the forward transform's output order", whatever that
// is for its FFT plan.  Instead of deriving slot -> frequency per plan, the engine transforms the monomial X
// once with the variant's own conversion kernel and reads the exponents off the result: slot s holds
// zeta_s = w^{e_s} (w = e^{i pi / N}), so X^d has the value w^{d e_s mod 2N} there.  multibit_combine_generic_kernel
// then prepares the GGSWs elementwise for any (N, k, levels), and the classic kernels' EXTPROD mode
// (pbs_kernels.hip.h, pbs_large_kernels.hip.h) multiplies them in.
struct MultiBitCombineGenericArgs {
    const uint64_t* lwe_small;     // [count][n+1] (already offset to the sub-batch)
    const double2* fbsk;           // [n/G][2^G][ggsw_elems]
    const uint32_t* slot_exp;      // [P]
    double2* combined;             // [count][n/G][ggsw_elems]
    uint32_t n, logN, P, ggsw_elems, count;
};

template <int G>
__global__ void __launch_bounds__(256)
multibit_combine_generic_kernel(MultiBitCombineGenericArgs a) {
    constexpr int SEL = (1 << G) - 1, EPT = 2, CHUNK = 8;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lo_bits = (a.logN + 1) / 2, hi_bits = a.logN + 1 - lo_bits, N = 1u << a.logN;
    double2* root_lo = reinterpret_cast<double2*>(smem);
    double2* root_hi = root_lo + (1u << lo_bits);
    for (uint32_t e = threadIdx.x; e < (1u << lo_bits); e += 256) {
        double sn, cs;
        sincospi((double)e / (double)N, &sn, &cs);
        root_lo[e] = make_double2(cs, sn);
    }
    for (uint32_t e = threadIdx.x; e < (1u << hi_bits); e += 256) {
        double sn, cs;
        sincospi((double)((size_t)e << lo_bits) / (double)N, &sn, &cs);
        root_hi[e] = make_double2(cs, sn);
    }
    const uint32_t grp = blockIdx.x, groups = a.n / G;
    const double2* gk = a.fbsk + (size_t)grp * (SEL + 1) * a.ggsw_elems;
    uint32_t elem[EPT], expo[EPT];
    double2 gv[SEL + 1][EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        elem[k] = (blockIdx.y * EPT + k) * 256 + threadIdx.x;
        const bool live = elem[k] < a.ggsw_elems;
        expo[k] = live ? a.slot_exp[elem[k] % a.P] : 0;
#pragma unroll
        for (int s = 0; s <= SEL; s++) gv[s][k] = live ? gk[(size_t)s * a.ggsw_elems + elem[k]] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const uint32_t first = blockIdx.z * CHUNK, last = min(a.count, first + CHUNK);
    for (uint32_t sample = first; sample < last; sample++) {
        const uint64_t* lwe = a.lwe_small + (size_t)sample * (a.n + 1) + (size_t)grp * G;
        cplx comb[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) { comb[k].re = gv[0][k].x; comb[k].im = gv[0][k].y; }
#pragma unroll
        for (int s = 1; s <= SEL; s++) {
            uint64_t sum = 0;
#pragma unroll
            for (int b = 0; b < G; b++)
                if ((s >> (G - 1 - b)) & 1) sum += lwe[b];
            const uint32_t d = modulus_switch(sum, (int)a.logN);
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const uint32_t mi = (d * expo[k]) & (2u * N - 1u);
                const double2 x = root_lo[mi & ((1u << lo_bits) - 1u)], y = root_hi[mi >> lo_bits];
                const double mre = x.x * y.x - x.y * y.y, mim = x.x * y.y + x.y * y.x;
                const double2 v = gv[s][k];
                comb[k].re = fma(v.x, mre, fma(-v.y, mim, comb[k].re));
                comb[k].im = fma(v.x, mim, fma(v.y, mre, comb[k].im));
            }
        }
        double2* out = a.combined + ((size_t)sample * groups + grp) * a.ggsw_elems;
#pragma unroll
        for (int k = 0; k < EPT; k++)
            if (elem[k] < a.ggsw_elems) out[elem[k]] = make_double2(comb[k].re, comb[k].im);
    }
}

// PRE: args.fbsk points at multibit_combine_kernel's output for this batch instead of the Fourier key
template <int LOGN, int LOGR, int K1, int G, bool PRE = false>
__global__ void __launch_bounds__((BrMultiBitCfg<LOGN, LOGR, K1, G>::THREADS))
blind_rotate_multibit_kernel(BlindRotateArgs args) {
    using CFG = BrMultiBitCfg<LOGN, LOGR, K1, G>;
    using BASE = typename CFG::Base;
    using PL = typename CFG::PL;
    static_assert(PL::SWAP, "multi-bit kernel is written for the FftSwap10 plan");
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T, SEL = CFG::SEL;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds_x = reinterpret_cast<double*>(smem);                              // [K1][GROUP_SLOTS]
    double* lds_f = lds_x + (size_t)K1 * BASE::GROUP_SLOTS;                       // [K1][GROUP_SLOTS]
    double2* root_lo = reinterpret_cast<double2*>(lds_f + (size_t)K1 * BASE::GROUP_SLOTS);
    double2* root_hi = root_lo + (1 << CFG::ROOT_LO_BITS);
    uint32_t* lds_deg = reinterpret_cast<uint32_t*>(root_hi + (1 << CFG::ROOT_HI_BITS));   // [n/G][SEL]

    const int g = threadIdx.x / T, tau = threadIdx.x % T;
    const uint32_t sample = blockIdx.x;
    const uint32_t n = args.n, groups = n / G;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    double* xre = lds_x + (size_t)g * BASE::GROUP_SLOTS;
    double* xim = xre + BASE::PLANE;
    const uint32_t bL = args.base_log;

    if constexpr (!PRE) {
        // monomial degrees of every group (:57-70: wrapping sum of the selected mask elements, then
        // modulus switch), selector bit G-1-b <-> mask element b
        for (uint32_t e = threadIdx.x; e < groups * SEL; e += CFG::THREADS) {
            const uint32_t grp = e / SEL, sel = e % SEL + 1;
            uint64_t sum = 0;
#pragma unroll
            for (int b = 0; b < G; b++)
                if ((sel >> (G - 1 - b)) & 1) sum += lwe[(size_t)grp * G + b];
            lds_deg[e] = modulus_switch(sum, LOGN);
        }
        multibit_fill_roots<CFG>(root_lo, root_hi);
    }

    FftConsts<PL> fc;
    fft_init_consts<PL>(fc, tau);
    cplx twist[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        double sn, cs;
        sincospi((double)PL::point(tau, m) / (double)N, &sn, &cs);
        twist[m].re = cs; twist[m].im = sn;
    }
    const uint32_t c_tau = multibit_slot_exponent<N>(tau);

    // acc <- LUT * X^{-ms(body)}
    uint64_t acc_lo[R], acc_hi[R];
    {
        const uint32_t d = modulus_switch(lwe[n], LOGN);
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
#pragma unroll
        for (int m = 0; m < R; m++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t j = PL::point(tau, m) + h * P;
                const uint32_t src = (j + rem) & (N - 1);
                const bool neg = ((j + rem) >= (uint32_t)N) != odd;
                uint64_t v = lut[(size_t)g * N + src];
                v = neg ? (0 - v) : v;
                if (h == 0) acc_lo[m] = v; else acc_hi[m] = v;
            }
        }
    }
    __syncthreads();

    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)K1 * K1 * P;

    double2 ahead[PRE ? K1 : 1][PRE ? R : 1];
    auto request_ahead = [&](uint32_t grp) {
        if constexpr (PRE) {
            const double2* ck = fbsk + ((size_t)sample * groups + grp) * GGSW_ELEMS;
#pragma unroll
            for (int r = 0; r < K1; r++)
#pragma unroll
                for (int rho = 0; rho < R; rho++)
                    ahead[r][rho] = ck[((size_t)((g + r) % K1) * K1 + g) * P + rho * T + tau];
        }
    };
    request_ahead(0);

    for (uint32_t grp = 0; grp < groups; grp++) {
        // ---- request this group's 2^G GGSWs (column g, rows (g + r) % K1) now: they arrive from L2
        //      while the accumulator is decomposed and transformed ----
        //      (grouping factor 2: all four at once; 3: two in flight, the next one requested while the
        //      current one is folded in -- eight would need 256 VGPRs)
        //      PRE: one GGSW, this LWE's combined one
        const double2* gk = PRE ? fbsk + ((size_t)sample * groups + grp) * GGSW_ELEMS
                                : fbsk + (size_t)grp * (SEL + 1) * GGSW_ELEMS;
        constexpr bool PREFETCH_ALL = PRE || (SEL + 1) * K1 * R * 4 <= 96;
        constexpr int NBUF = PRE ? 1 : (PREFETCH_ALL ? SEL + 1 : 2);
        double2 gv[NBUF][K1][R];
        auto request = [&](int s) {
#pragma unroll
            for (int r = 0; r < K1; r++) {
                const int row = (g + r) % K1;
#pragma unroll
                for (int rho = 0; rho < R; rho++)
                    gv[s % NBUF][r][rho] = gk[(size_t)s * GGSW_ELEMS + ((size_t)row * K1 + g) * P + rho * T + tau];
            }
        };
        if constexpr (PRE) {      // requested one step ahead: a whole step to arrive from HBM
#pragma unroll
            for (int r = 0; r < K1; r++)
#pragma unroll
                for (int rho = 0; rho < R; rho++) gv[0][r][rho] = ahead[r][rho];
            if (grp + 1 < groups) request_ahead(grp + 1);
        } else {
#pragma unroll
            for (int s = 0; s < NBUF; s++) request(s);
        }

        // ---- external product acc <- GGSW (x) acc (ggsw.rs:477-598 on a zeroed destination) ----
        cplx x[K1][R];
#pragma unroll
        for (int m = 0; m < R; m++) {     // (the biased two-instruction digit of the classic kernels would cost this
            cplx z;                       //  kernel 16 more VGPRs for its constants: the prefetched GGSWs need them)
            z.re = (double)decomp_single_digit(acc_lo[m], bL);
            z.im = (double)decomp_single_digit(acc_hi[m], bL);
            x[0][m] = cmul(z, twist[m]);
        }
        swap10_forward_head(x[0], fc, xre, xim, tau);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < K1; r++) {
            const int row = (g + r) % K1;
            const double* rre = lds_x + (size_t)row * BASE::GROUP_SLOTS;
            swap10_forward_tail(x[r], rre, rre + BASE::PLANE, tau);
        }

        // ---- the group's GGSW: G0 + sum_sel G_sel * monomial_sel (prepare_multi_bit_ggsw, :18-83) ----
        cplx comb[K1][R];
#pragma unroll
        for (int r = 0; r < K1; r++)
#pragma unroll
            for (int rho = 0; rho < R; rho++) { comb[r][rho].re = gv[0][r][rho].x; comb[r][rho].im = gv[0][r][rho].y; }
        if constexpr (!PRE) {
#pragma unroll
            for (int s = 1; s <= SEL; s++) {
                if (!PREFETCH_ALL && s + 1 <= SEL) request(s + 1);      // into the buffer selector s-1 just left
                cplx mono, turn;
                multibit_monomial<CFG>(lds_deg[grp * SEL + (s - 1)], c_tau, root_lo, root_hi, mono, turn);
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
#pragma unroll
                    for (int r = 0; r < K1; r++) {
                        const double2 v = gv[s % NBUF][r][rho];
                        comb[r][rho].re = fma(v.x, mono.re, fma(-v.y, mono.im, comb[r][rho].re));
                        comb[r][rho].im = fma(v.x, mono.im, fma(v.y, mono.re, comb[r][rho].im));
                    }
                    mono = cmul(mono, turn);
                }
            }
        }

        cplx outf[R];
#pragma unroll
        for (int r = 0; r < K1; r++) {
#pragma unroll
            for (int rho = 0; rho < R; rho++) {
                const cplx bv = comb[r][rho], f = x[r][rho];
                if (r == 0) {
                    outf[rho].re = bv.re * f.re - bv.im * f.im;
                    outf[rho].im = bv.re * f.im + bv.im * f.re;
                } else {
                    outf[rho].re = fma(bv.re, f.re, fma(-bv.im, f.im, outf[rho].re));
                    outf[rho].im = fma(bv.re, f.im, fma(bv.im, f.re, outf[rho].im));
                }
            }
        }
        double* fre = lds_f + (size_t)g * BASE::GROUP_SLOTS;
        double* fim = fre + BASE::PLANE;
        swap10_inverse_head(outf, fre, fim, tau);
        __syncthreads();
        swap10_inverse_tail(outf, fc, fre, fim, tau);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const cplx t = cmul_conj(outf[m], twist[m]);
            acc_lo[m] = from_torus(t.re);                  // the destination was zero: no accumulate
            acc_hi[m] = from_torus(t.im);
        }
    }

    // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146)
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
    for (int m = 0; m < R; m++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t j = PL::point(tau, m) + h * P;
            const uint64_t v = h == 0 ? acc_lo[m] : acc_hi[m];
            if (g == K1 - 1) {
                if (j == 0) out[(size_t)(K1 - 1) * N] = v;
            } else {
                if (j == 0) out[(size_t)g * N] = v;
                else out[(size_t)g * N + (N - j)] = 0 - v;
            }
        }
    }
}

}  // namespace fhe
