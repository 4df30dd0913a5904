// pbs_large_kernels.hip.h -- blind rotation for polynomial sizes whose accumulator and spectra do
// not fit the 160 KB LDS of a CU (N >= 16384: PARAM_MESSAGE_4_CARRY_4 N = 32768, shortint/parameters/mod.rs:1063-1077;
// N = 8192 is instantiable here too but runs on pbs_seq_kernels.hip.h since round 2).
//
// Same algorithm as pbs_kernels.hip.h (fft64/crypto/bootstrap.rs:242-364, ggsw.rs:477-598), other
// data placement: one workgroup per LWE keeps its accumulator (k+1)*N u64 and two spectrum buffers
// in a private HBM/L2 workspace, and the size-P = N/2 complex FFT is done "four-step": P = P1*P2,
//   phase 1  P2 column transforms of size P1 (stride P2) + twiddle w_P^{q1*b}      -> tmp
//   phase 2  P1 row transforms of size P2 (contiguous), Fourier multiply-accumulate against the
//            GGSW rows in registers, inverse row transforms, conjugate twiddle      -> tmp2
//   phase 3  P2 inverse column transforms, untwist, torus rounding, accumulate into acc
// Each size-P1 / size-P2 transform is the in-register radix-8 kernel of negacyclic_fft.hip.h run by
// 16 (or 8) threads with wave-local LDS exchanges; four adjacent columns share a wavefront quarter
// so that every strided access still moves whole 64-byte sectors.  Three workgroup barriers per CMUX
// step order the phases (global memory is coherent inside a workgroup: one CU, one vector L1).
// The spectrum order is whatever the two in-place transforms leave; the Fourier key is converted
// with the same code (bsk_convert_large_kernel), so the pointwise product needs no reordering.
#pragma once
#include <type_traits>

#include "pbs_kernels.hip.h"

namespace fhe {

template <int LOGN, int K1, int L>
struct BrLargeCfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGP = LOGN - 1;
    static constexpr int P = 1 << LOGP;
    static constexpr int LOGP1 = (LOGP + 1) / 2, LOGP2 = LOGP / 2;
    static constexpr int P1 = 1 << LOGP1, P2 = 1 << LOGP2;
    using PA = FftPlan<LOGP1, 3>;      // column transforms (size P1)
    using PB = FftPlan<LOGP2, 3>;      // row transforms (size P2)
    static constexpr int R = 8;
    static constexpr int TA = PA::T, TB = PB::T;
    static constexpr int THREADS = 512;
    static constexpr int SUBS_A = THREADS / TA, SUBS_B = THREADS / TB;   // transforms in flight
    // LDS slots per transform: two planes + padding that puts neighbouring transforms 16 slots apart mod 32
    static constexpr int SLOTS_A = 2 * P1 + 16, SLOTS_B = 2 * P2 + 16;
    static constexpr size_t LDS_PLANES =
        (size_t)(SUBS_A * SLOTS_A > SUBS_B * SLOTS_B ? SUBS_A * SLOTS_A : SUBS_B * SLOTS_B) * 8;
    // unit roots e^{2 pi i e / 2N}, e < 2N, as a product of two table entries (low / high bits of e):
    // replaces sincospi in the twist and inter-step twiddles (it was > half of the kernel's VALU work)
    static constexpr int ROOT_BITS = LOGN + 1, ROOT_LO = ROOT_BITS / 2, ROOT_HI = ROOT_BITS - ROOT_LO;
    static constexpr size_t LDS_ROOTS = ((size_t)(1 << ROOT_LO) + (size_t)(1 << ROOT_HI)) * 16;
    // inter-pass twiddles of the two sub-transforms as LDS tables (FftTwiddleTable) instead of
    // 64 VGPRs each: the registers are needed for loads in flight
    static constexpr size_t LDS_TW_A = (size_t)FftTwiddleTable<PA>::ENTRIES * 16;
    static constexpr size_t LDS_TW_B = (size_t)FftTwiddleTable<PB>::ENTRIES * 16;
    static constexpr size_t LDS_BYTES = LDS_PLANES + LDS_ROOTS + LDS_TW_A + LDS_TW_B;
    // per-LWE workspace in HBM (bytes): acc | tmp[L*K1][P] c64 | tmp2[K1][P] c64
    static constexpr size_t WS_ACC = (size_t)K1 * N * 8;
    static constexpr size_t WS_TMP = (size_t)L * K1 * P * 16;
    static constexpr size_t WS_TMP2 = (size_t)K1 * P * 16;
    static constexpr size_t WS_BYTES = WS_ACC + WS_TMP + WS_TMP2;
    // Workspace arrays are P1 x P2 matrices (row = column-transform index, col = row-transform index)
    // stored in column tiles of width W = the number of columns one workgroup iteration transforms:
    // [col / W][row][col % W].  A column iteration then touches one dense P1*W tile (instead of
    // P1 chunks of W elements a whole matrix row apart) and a row still reads W-element runs, so
    // DRAM pages are opened once per iteration rather than once per P2/W iterations.
    static constexpr int LOGW = ilog2c(SUBS_A < P2 ? SUBS_A : P2);
    __host__ __device__ static constexpr size_t tix(int row, int col) {
        return ((size_t)(col >> LOGW) << (LOGP1 + LOGW)) | ((size_t)row << LOGW) | (size_t)(col & ((1 << LOGW) - 1));
    }
    // coefficient j (< N) of an accumulator polynomial: halves [0,P) and [P,N) tiled separately
    __host__ __device__ static constexpr size_t aix(uint32_t j) {
        return (size_t)(j & ~(uint32_t)(P - 1)) + tix((int)((j & (P - 1)) >> LOGP2), (int)(j & (P2 - 1)));
    }
};

// frequency index of the value an in-place DIF transform leaves at last-pass address A
template <class PL>
__device__ __forceinline__ int freq_of_addr(int A) {
    int q = 0, weight = 1;
#pragma unroll
    for (int s = 0; s < PL::NP; s++) {
        const int lr = PL::log_radix(s);
        const int lS1 = PL::log_S(s) - lr;
        const int digit = (A >> lS1) & ((1 << lr) - 1);
        q += digit * weight;
        weight <<= lr;
    }
    return q;
}

// last-pass address of register slot rho of thread tau
template <class PL>
__device__ __forceinline__ int slot_addr(int tau, int rho) {
    constexpr int s = PL::NP - 1;
    const int rr = 1 << PL::log_radix(s);
    return pass_addr<PL>(s, tau, rho / rr, rho % rr);
}

__device__ __forceinline__ cplx unit_root(double turns) {   // e^{2 pi i turns}
    double sn, cs;
    sincospi(2.0 * turns, &sn, &cs);
    cplx w; w.re = cs; w.im = sn;
    return w;
}

// e^{2 pi i e / 2N} from the two LDS tables (e taken mod 2N)
template <class CFG>
struct RootTable {
    const double2* lo;
    const double2* hi;
    __device__ __forceinline__ void init(double2* base, int tid, int nthreads) {
        double2* l = base;
        double2* h = base + (1 << CFG::ROOT_LO);
        for (int e = tid; e < (1 << CFG::ROOT_LO); e += nthreads) {
            const cplx w = unit_root((double)e / (double)(2 * CFG::N));
            l[e] = make_double2(w.re, w.im);
        }
        for (int e = tid; e < (1 << CFG::ROOT_HI); e += nthreads) {
            const cplx w = unit_root((double)((size_t)e << CFG::ROOT_LO) / (double)(2 * CFG::N));
            h[e] = make_double2(w.re, w.im);
        }
        lo = l; hi = h;
    }
    __device__ __forceinline__ cplx get(uint32_t e) const {
        e &= (2u * CFG::N - 1u);
        const double2 a = lo[e & ((1u << CFG::ROOT_LO) - 1u)], b = hi[e >> CFG::ROOT_LO];
        cplx r;
        r.re = a.x * b.x - a.y * b.y;
        r.im = a.x * b.y + a.y * b.x;
        return r;
    }
};

// ---- phase helpers -------------------------------------------------------------------------------
// column transform forward: x[m] holds point (a = tau + TA*m, column b); result slot rho is written to
// dst[slot_addr * P2 + b] after the inter-step twiddle w_P^{-q1*b} (forward sign convention e^{-2 pi i})
template <class CFG>
__device__ __forceinline__ void column_forward_store(cplx* x, const FftTwiddleTable<typename CFG::PA>& fc, double* re,
                                                     double* im, int tau, int b, double2* dst, const RootTable<CFG>& roots) {
    using PA = typename CFG::PA;
    fft_forward<PA>(x, fc, re, im, tau);
#pragma unroll
    for (int rho = 0; rho < CFG::R; rho++) {
        const int A = slot_addr<PA>(tau, rho);
        const int q1 = freq_of_addr<PA>(A);
        const cplx w = roots.get(0u - 4u * (uint32_t)(q1 * b));     // e^{-2 pi i q1 b / P}, 1/P = 4/(2N)
        const cplx v = cmul(x[rho], w);
        dst[CFG::tix(A, b)] = make_double2(v.re, v.im);
    }
}

// ------------------------------------------------------------------------------------------------
template <int LOGN, int K1, int L>
__global__ void __launch_bounds__((BrLargeCfg<LOGN, K1, L>::THREADS))
bsk_convert_large_kernel(const uint64_t* __restrict__ bsk_std, double* __restrict__ fbsk, uint32_t n_polys,
                         double2* __restrict__ workspace /* gridDim.x * P c64 */) {
    using CFG = BrLargeCfg<LOGN, K1, L>;
    using PA = typename CFG::PA;
    using PB = typename CFG::PB;
    constexpr int N = CFG::N, P = CFG::P, P1 = CFG::P1, P2 = CFG::P2, R = CFG::R;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x;
    const int subA = tid / CFG::TA, tauA = tid % CFG::TA;
    const int subB = tid / CFG::TB, tauB = tid % CFG::TB;
    double2* twa = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES + CFG::LDS_ROOTS);
    double2* twb = twa + FftTwiddleTable<PA>::ENTRIES;
    FftTwiddleTable<PA>::fill(twa, threadIdx.x, CFG::THREADS);
    FftTwiddleTable<PB>::fill(twb, threadIdx.x, CFG::THREADS);
    const FftTwiddleTable<PA> fca{twa, tauA};
    const FftTwiddleTable<PB> fcb{twb, tauB};
    double2* tmp = workspace + (size_t)blockIdx.x * P;
    RootTable<CFG> roots;
    roots.init(reinterpret_cast<double2*>(smem + CFG::LDS_PLANES), tid, CFG::THREADS);
    __syncthreads();
    for (uint32_t poly = blockIdx.x; poly < n_polys; poly += gridDim.x) {
        const uint64_t* src = bsk_std + (size_t)poly * N;
        for (int b0 = 0; b0 < P2; b0 += CFG::SUBS_A) {
            const int b = b0 + subA;
            cplx x[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int j = (tauA + CFG::TA * m) * P2 + b;
                cplx z;   // forward_as_torus (fft/mod.rs:197-218) with the inverse's 1/P folded in
                z.re = i64_to_f64(src[j]) * (5.421010862427522e-20 / P);
                z.im = i64_to_f64(src[j + P]) * (5.421010862427522e-20 / P);
                x[m] = cmul(z, roots.get((uint32_t)j));   // twisty e^{i pi j / N}
            }
            column_forward_store<CFG>(x, fca, lds + (size_t)subA * CFG::SLOTS_A, lds + (size_t)subA * CFG::SLOTS_A + P1 + 2,
                                      tauA, b, tmp, roots);
        }
        __syncthreads();
        double2* out = reinterpret_cast<double2*>(fbsk) + (size_t)poly * P;
        for (int r0 = 0; r0 < P1; r0 += CFG::SUBS_B) {
            const int r = r0 + subB;
            cplx x[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                const double2 v = tmp[CFG::tix(r, tauB + CFG::TB * m)];
                x[m].re = v.x; x[m].im = v.y;
            }
            fft_forward<PB>(x, fcb, lds + (size_t)subB * CFG::SLOTS_B, lds + (size_t)subB * CFG::SLOTS_B + P2 + 2, tauB);
#pragma unroll
            for (int rho = 0; rho < R; rho++) out[(size_t)r * P2 + rho * CFG::TB + tauB] = make_double2(x[rho].re, x[rho].im);
        }
        __syncthreads();
    }
}

struct BlindRotateLargeArgs {
    BlindRotateArgs base;
    unsigned char* workspace;    // batch * WS_BYTES
};

// ------------------------------------------------------------------------------------------------
// EXTPROD: see blind_rotate_kernel (multi-bit PBS through the two-kernel path)
template <int LOGN, int K1, int L, bool EXTPROD = false>
__global__ void __launch_bounds__((BrLargeCfg<LOGN, K1, L>::THREADS))
blind_rotate_large_kernel(BlindRotateLargeArgs la) {
    using CFG = BrLargeCfg<LOGN, K1, L>;
    using PA = typename CFG::PA;
    using PB = typename CFG::PB;
    constexpr int N = CFG::N, P = CFG::P, P1 = CFG::P1, P2 = CFG::P2, R = CFG::R, NT = CFG::THREADS;
    const BlindRotateArgs& args = la.base;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    __shared__ uint32_t s_d;   // modulus-switched mask element of the current step

    const int tid = threadIdx.x;
    const int subA = tid / CFG::TA, tauA = tid % CFG::TA;
    const int subB = tid / CFG::TB, tauB = tid % CFG::TB;
    double* areA = lds + (size_t)subA * CFG::SLOTS_A;
    double* aimA = areA + P1 + 2;
    double* breB = lds + (size_t)subB * CFG::SLOTS_B;
    double* bimB = breB + P2 + 2;
    const uint32_t sample = blockIdx.x;
    const uint32_t n = args.n;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    unsigned char* ws = la.workspace + (size_t)sample * CFG::WS_BYTES;
    uint64_t* acc = reinterpret_cast<uint64_t*>(ws);                                   // [K1][N]
    double2* tmp = reinterpret_cast<double2*>(ws + CFG::WS_ACC);                       // [L*K1][P]
    double2* tmp2 = reinterpret_cast<double2*>(ws + CFG::WS_ACC + CFG::WS_TMP);        // [K1][P]
    const uint32_t bL = args.base_log * L;

    double2* twa = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES + CFG::LDS_ROOTS);
    double2* twb = twa + FftTwiddleTable<PA>::ENTRIES;
    FftTwiddleTable<PA>::fill(twa, threadIdx.x, CFG::THREADS);
    FftTwiddleTable<PB>::fill(twb, threadIdx.x, CFG::THREADS);
    const FftTwiddleTable<PA> fca{twa, tauA};
    const FftTwiddleTable<PB> fcb{twb, tauB};
    RootTable<CFG> roots;
    roots.init(reinterpret_cast<double2*>(smem + CFG::LDS_PLANES), tid, NT);

    // acc <- LUT * X^{-ms(body)}
    {
        const uint32_t d = modulus_switch(lwe[n], LOGN);
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
        for (int e = tid; e < K1 * N; e += NT) {
            const uint32_t p = e >> LOGN, j = e & (N - 1);
            const uint32_t src = (j + rem) & (N - 1);
            const bool neg = ((j + rem) >= (uint32_t)N) != odd;
            const uint64_t v = lut[(size_t)p * N + src];
            acc[(size_t)p * N + CFG::aix(j)] = neg ? (0 - v) : v;
        }
    }
    __syncthreads();

    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)L * K1 * K1 * P;

    const uint32_t steps = EXTPROD ? n / args.grouping : n;
    if constexpr (EXTPROD) fbsk += (size_t)sample * steps * GGSW_ELEMS;
    FHE_STAMP_DECL;
    FHE_STAMP(-1);
    for (uint32_t i = 0; i < steps; i++) {
        if (tid == 0) {
            const uint64_t a = lwe[i];
            s_d = EXTPROD ? 0u : (a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN));
        }
        __syncthreads();
        FHE_STAMP(0);    // step head (mask element broadcast)
        const uint32_t d = s_d;
        if (d == 0xFFFFFFFFu) { __syncthreads(); continue; }
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;

        // ---- phase 1: decompose (acc*X^d - acc), twist, column transforms, twiddle -> tmp ----
        // The accumulator words of iteration t+1 are requested before iteration t is transformed
        // (the exchanges' wavefront fences would otherwise pin the loads behind the FFT).
        {
            constexpr int COLS = P2 / CFG::SUBS_A, ITERS = K1 * COLS;
            using state_t = typename std::conditional<(L >= 3), uint64_t, uint32_t>::type;
            uint64_t rot[2 * R], own[2 * R];
            auto issue = [&](int t) {
                const int b = (t % COLS) * CFG::SUBS_A + subA;
                const uint64_t* ap = acc + (size_t)(t / COLS) * N;
#pragma unroll
                for (int m = 0; m < R; m++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t j = (tauA + CFG::TA * m) * P2 + b + h * P;
                        if constexpr (!EXTPROD) rot[2 * m + h] = ap[CFG::aix((j - rem) & (N - 1))];
                        own[2 * m + h] = ap[CFG::aix(j)];
                    }
                }
            };
            issue(0);
            for (int t = 0; t < ITERS; t++) {
                const int p = t / COLS, b = (t % COLS) * CFG::SUBS_A + subA;
                // decomposition state: 32 bits suffice while base_log * level <= 31 (L <= 2 here)
                state_t st_lo[R], st_hi[R];
#pragma unroll
                for (int m = 0; m < R; m++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t j = (tauA + CFG::TA * m) * P2 + b + h * P;
                        const bool neg = (j < rem) != odd;
                        uint64_t v = EXTPROD ? 0 : rot[2 * m + h];
                        v = neg ? (0 - v) : v;
                        const uint64_t ct1 = EXTPROD ? own[2 * m + h] : v - own[2 * m + h];
                        state_t st;
                        if constexpr (L >= 3) st = decomp_init_state64(ct1, bL);
                        else st = decomp_init_state(ct1, bL);
                        if (h == 0) st_lo[m] = st; else st_hi[m] = st;
                    }
                }
                if (t + 1 < ITERS) issue(t + 1);
#pragma unroll
                for (int it = 0; it < L; it++) {
                    cplx x[R];
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const int j = (tauA + CFG::TA * m) * P2 + b;
                        cplx z;
                        if constexpr (L >= 3) {
                            z.re = (double)decomp_next_digit64(st_lo[m], args.base_log);
                            z.im = (double)decomp_next_digit64(st_hi[m], args.base_log);
                        } else {
                            z.re = (double)decomp_next_digit(st_lo[m], args.base_log);
                            z.im = (double)decomp_next_digit(st_hi[m], args.base_log);
                        }
                        x[m] = cmul(z, roots.get((uint32_t)j));
                    }
                    column_forward_store<CFG>(x, fca, areA, aimA, tauA, b, tmp + (size_t)(it * K1 + p) * P, roots);
                }
            }
        }
        FHE_STAMP(1);    // phase 1 (this wave's share)
        __syncthreads();
        FHE_STAMP(2);    // wait for the slowest wave of phase 1

        // ---- phase 2: row transforms, multiply-accumulate with the GGSW, inverse row transforms -> tmp2 ----
        const double2* bk0 = fbsk + (size_t)i * GGSW_ELEMS;
        for (int r0 = 0; r0 < P1; r0 += CFG::SUBS_B) {
            const int r = r0 + subB;
            cplx outf[K1][R];
            // (it, row) pairs in the order the reference accumulates them (ggsw.rs:524); the spectrum
            // row of pair u+1 and the GGSW rows of pair u are in flight while pair u is transformed
            constexpr int PAIRS = L * K1;
            double2 xin[R];
            auto issue_row = [&](int u) {
                const double2* spoly = tmp + (size_t)u * P;          // u = it * K1 + row
#pragma unroll
                for (int m = 0; m < R; m++) xin[m] = spoly[CFG::tix(r, tauB + CFG::TB * m)];
            };
            issue_row(0);
#pragma unroll
            for (int u = 0; u < PAIRS; u++) {
                const int it = u / K1, row = u % K1;
                const int lvl_idx = L - 1 - it;                      // ggsw.rs:524
                cplx x[R];
#pragma unroll
                for (int m = 0; m < R; m++) { x[m].re = xin[m].x; x[m].im = xin[m].y; }
                double2 bv[K1][R];
#pragma unroll
                for (int col = 0; col < K1; col++) {
                    const double2* bk = bk0 + (((size_t)lvl_idx * K1 + row) * K1 + col) * P + (size_t)r * P2;
#pragma unroll
                    for (int rho = 0; rho < R; rho++) bv[col][rho] = bk[rho * CFG::TB + tauB];
                }
                if (u + 1 < PAIRS) issue_row(u + 1);
                fft_forward<PB>(x, fcb, breB, bimB, tauB);
#pragma unroll
                for (int col = 0; col < K1; col++) {
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const double2 b2 = bv[col][rho];
                        if (u == 0) {
                            outf[col][rho].re = b2.x * x[rho].re - b2.y * x[rho].im;
                            outf[col][rho].im = b2.x * x[rho].im + b2.y * x[rho].re;
                        } else {
                            outf[col][rho].re = fma(b2.x, x[rho].re, fma(-b2.y, x[rho].im, outf[col][rho].re));
                            outf[col][rho].im = fma(b2.x, x[rho].im, fma(b2.y, x[rho].re, outf[col][rho].im));
                        }
                    }
                }
            }
            const int q1 = freq_of_addr<PA>(r);
#pragma unroll
            for (int col = 0; col < K1; col++) {
                fft_inverse<PB>(outf[col], fcb, breB, bimB, tauB);
                double2* dpoly = tmp2 + (size_t)col * P;
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const int b = tauB + CFG::TB * m;
                    const cplx w = roots.get(4u * (uint32_t)(q1 * b));   // conj of the forward twiddle
                    const cplx v = cmul(outf[col][m], w);
                    dpoly[CFG::tix(r, b)] = make_double2(v.re, v.im);
                }
            }
        }
        FHE_STAMP(3);    // phase 2
        __syncthreads();
        FHE_STAMP(4);    // barrier after phase 2

        // ---- phase 3: inverse column transforms, untwist, torus rounding, accumulate ----
        for (int p = 0; p < K1; p++) {
            uint64_t* ap = acc + (size_t)p * N;
            const double2* sp = tmp2 + (size_t)p * P;
            for (int b0 = 0; b0 < P2; b0 += CFG::SUBS_A) {
                const int b = b0 + subA;
                cplx x[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 v = sp[CFG::tix(slot_addr<PA>(tauA, rho), b)];
                    x[rho].re = v.x; x[rho].im = v.y;
                }
                uint64_t a_lo[R], a_hi[R];      // requested now, needed after the transform
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const size_t ja = CFG::aix((uint32_t)((tauA + CFG::TA * m) * P2 + b));
                    a_lo[m] = EXTPROD ? 0 : ap[ja];     // EXTPROD: zeroed destination
                    a_hi[m] = EXTPROD ? 0 : ap[ja + P];
                }
                fft_inverse<PA>(x, fca, areA, aimA, tauA);
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const int j = (tauA + CFG::TA * m) * P2 + b;
                    const cplx t = cmul_conj(x[m], roots.get((uint32_t)j));
                    const size_t ja = CFG::aix((uint32_t)j);
                    ap[ja] = a_lo[m] + from_torus(t.re);
                    ap[ja + P] = a_hi[m] + from_torus(t.im);
                }
            }
        }
        FHE_STAMP(5);    // phase 3
        __syncthreads();
        FHE_STAMP(6);    // barrier after phase 3
    }
#ifdef FHESTR_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)
        for (int sg = 0; sg < STAMP_SEGS; sg++)
            g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * STAMP_SEGS + sg] = stamp_acc[sg];
#endif

    // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146)
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
    for (int e = tid; e < K1 * N; e += NT) {
        const uint32_t p = e >> LOGN, j = e & (N - 1);
        const uint64_t v = acc[(size_t)p * N + CFG::aix(j)];
        if (p == K1 - 1) {
            if (j == 0) out[(size_t)(K1 - 1) * N] = v;
        } else {
            if (j == 0) out[(size_t)p * N] = v;
            else out[(size_t)p * N + (N - j)] = 0 - v;
        }
    }
}

}  // namespace fhe
