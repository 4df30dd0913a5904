// negacyclic_fft.hip.h -- in-register / LDS-exchanged complex FFT of size P = N/2 for gfx950.
//
// Replaces the reference's f64 negacyclic transform
//   tfhe/src/core_crypto/fft_impl/fft64/math/fft/mod.rs:378-405,487-557 (forward_as_integer /
//   forward_as_torus / add_backward_in_place_as_torus), conversions :197-304, twisties :58-69,
// and the un-vendored concrete-fft 0.3.0 Plan::fwd / Plan::inv it calls.
//
// Design (MI355X-first, not a translation):
//  * one polynomial is transformed by a "group" of T = P/R threads, each holding R complex points
//    in VGPRs (f64 VALU is the binding unit on CDNA4: 16 FMA lanes/clk/SIMD);
//  * decimation-in-frequency, in place: pass s does radix-r_s butterflies fully in registers, then
//    the points are exchanged through LDS (split re/im planes, 8-byte accesses).  Output stays in
//    the scrambled "last-pass" order -- the Fourier-domain key is stored in exactly that order
//    (see bsk_convert_kernel), so no reordering pass ever runs; the inverse walks the passes back;
//  * all twiddles are per-thread constants computed once per kernel (sincospi) and kept in VGPRs
//    for the whole 742-step blind rotation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Device arithmetic may fuse multiply-add (v_fma_f64); host code of the library stays unfused so that
// client-side key generation is bit-reproducible against the oracle.
#pragma clang fp contract(fast)

namespace fhe {

struct cplx {
    double re, im;
};

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    cplx r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
__device__ __forceinline__ cplx cmul_conj(cplx a, cplx b) {  // a * conj(b)
    cplx r;
    r.re = a.re * b.re + a.im * b.im;
    r.im = a.im * b.re - a.re * b.im;
    return r;
}

// exp(-2*pi*i * j / 16), j = 0..15 (forward sign)
__device__ constexpr double kCos16[16] = {
    1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
    0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848,
    -1.0, -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508977173,
    0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848};
__device__ constexpr double kSin16[16] = {
    0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
    1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
    0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848,
    -1.0, -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508977173};

// multiply by exp(-+ 2*pi*i * j / len) with len in {2,4,8,16}; j, len are compile-time after
// unrolling, so the branches fold away.
template <bool INV>
__device__ __forceinline__ cplx mul_unit_root(cplx d, int j, int len) {
    const int e = (j * 16) / len;  // exponent in 16ths of a turn
    cplx r;
    if (e == 0) return d;
    if (e == 4) {  // forward: * (-i) ; inverse: * (+i)
        if (!INV) { r.re = d.im; r.im = -d.re; } else { r.re = -d.im; r.im = d.re; }
        return r;
    }
    if (e == 8) { r.re = -d.re; r.im = -d.im; return r; }
    if (e == 12) {
        if (!INV) { r.re = -d.im; r.im = d.re; } else { r.re = d.im; r.im = -d.re; }
        return r;
    }
    const double c = kCos16[e];
    const double s = INV ? kSin16[e] : -kSin16[e];
    if (e == 2 || e == 6 || e == 10 || e == 14) {  // |c| == |s| == sqrt(1/2): 2 mul + 2 add
        const double h = 0.70710678118654752440;
        const double sc = c > 0 ? 1.0 : -1.0, ss = s > 0 ? 1.0 : -1.0;
        // (re + i im) * h * (sc + i ss)
        r.re = h * (sc * d.re - ss * d.im);
        r.im = h * (ss * d.re + sc * d.im);
        return r;
    }
    r.re = d.re * c - d.im * s;
    r.im = d.re * s + d.im * c;
    return r;
}

__host__ __device__ constexpr int bitrev(int x, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++)
        if (x & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}
__host__ __device__ constexpr int ilog2c(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}

// Size-RR DFT on registers x[0..RR), natural order in and out.  y[q] = sum_m x[m] e^{-+2 pi i m q / RR}.
template <int RR, bool INV>
__device__ __forceinline__ void small_dft(cplx* x) {
    if (RR == 1) return;
#pragma unroll
    for (int len = RR; len >= 2; len >>= 1) {
        const int half = len / 2;
#pragma unroll
        for (int s = 0; s < RR; s += len) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                cplx a = x[s + j], b = x[s + j + half];
                cplx u, d;
                u.re = a.re + b.re; u.im = a.im + b.im;
                d.re = a.re - b.re; d.im = a.im - b.im;
                x[s + j] = u;
                x[s + j + half] = mul_unit_root<INV>(d, j, len);
            }
        }
    }
    // bit-reversal permutation (pure register renaming after unrolling)
    constexpr int LB = ilog2c(RR);
    cplx y[RR];
#pragma unroll
    for (int q = 0; q < RR; q++) y[q] = x[bitrev(q, LB)];
#pragma unroll
    for (int q = 0; q < RR; q++) x[q] = y[q];
}

// Compile-time description of the pass structure for P points with R per thread.
template <int LP, int LR>
struct FftPlan {
    static constexpr bool SWAP = false;
    static constexpr int LOGP = LP;
    static constexpr int LOGR = LR;
    static constexpr int P = 1 << LOGP;
    static constexpr int R = 1 << LOGR;
    static constexpr int T = P / R;                         // threads per polynomial
    static constexpr int FULL = LOGP / LOGR;                // passes of radix R
    static constexpr int LOGLAST = LOGP - FULL * LOGR;      // log2 of the trailing radix (0 = none)
    static constexpr int NP = FULL + (LOGLAST ? 1 : 0);     // number of passes
    static constexpr int NTW = NP - 1;                      // passes that carry inter-pass twiddles
    __host__ __device__ static constexpr int log_radix(int s) { return s < FULL ? LOGR : LOGLAST; }
    // complex point carried in register m of thread tau, and where coefficient j (< N = 2P) of a
    // polynomial sits in the accumulator's LDS copy
    __host__ __device__ static constexpr int point(int tau, int m) { return tau + T * m; }
    __host__ __device__ static constexpr int acc_slot(int j) { return j; }   // j < N = 2P
    // log2 of sub-transform size at the start of pass s
    __host__ __device__ static constexpr int log_S(int s) { return LOGP - (s < FULL ? s : FULL) * LOGR; }
};

// P = 1024 points, 4 per thread, 256 threads (4 waves): the plan PARAM_MESSAGE_2_CARRY_2 (N = 2048)
// runs on.  Same radix-4 decimation-in-frequency passes as FftPlan<10, 2>, but two of the four
// inter-pass exchanges never touch LDS: a 4x4 transpose between the thread's 4 registers and lane
// bits (5,4) is two rounds of v_permlane32_swap / v_permlane16_swap (gfx950), ~8 cycles per dword on
// the VALU against ~25 (ds_write_b64) + ~9 (ds_read_b64) per 8-byte exchange through LDS, whose
// write port (~94 B/clk/CU) is what the CMUX loop saturates first.  The one exchange that crosses
// wavefronts comes LAST in the forward direction (first in the inverse), so in the split kernel it
// doubles as the spectrum hand-over between the polynomial groups: every group reads the other
// groups' points out of that exchange and runs the final butterfly on them itself.
//
// Index bits of point j = (b9..b0), frequency f = k0 + 4 k1 + 16 k2 + 64 k3 + 256 k4:
//   start        regs b9b8 | lane(5,4) b7b6 | lane(3..0) b5..b2 | wave b1b0   (j = 256 m + 4 lane + wave)
//   pass 1  -> k0, register/lane(5,4) swap
//   pass 2  -> k1, wave-local LDS exchange        (regs b5b4, lane(5,4) b3b2, lane(3..0) k0k1)
//   pass 3  -> k2, register/lane(5,4) swap
//   pass 4  -> k3, LDS transpose regs <-> wave    (workgroup barrier; rows of 64 slots)
//   pass 5  -> k4                                 (regs k4 | wave k3 | lane(5,4) k2 | lane(3..0) k0k1)
// The inverse walks the same steps backwards.  Plane rows (64 slots each, 16 per plane): wave w owns
// the "slab" rows 4w..4w+3 (wave-local exchange, then its side of the transpose); the other side of
// the transpose is the "comb" {w, w+4, w+8, w+12}.  A forward and an inverse transform may share
// planes when every wave reads only its own comb (wide kernel); the split kernel, whose groups read
// each other's combs, gives the inverse its own planes.
struct FftSwap10 {
    static constexpr bool SWAP = true;
    static constexpr int LOGW = 2;          // log2(waves per polynomial): the index bits that live in the wave number
    static constexpr int LOGP = 10;
    static constexpr int LOGR = 2;
    static constexpr int P = 1024;
    static constexpr int R = 4;
    static constexpr int T = 256;
    static constexpr int FULL = 5;
    static constexpr int LOGLAST = 0;
    static constexpr int NP = 5;
    static constexpr int NTW = 4;
    __host__ __device__ static constexpr int log_radix(int) { return 2; }
    __host__ __device__ static constexpr int log_S(int s) { return 10 - 2 * s; }
    __host__ __device__ static constexpr int point(int tau, int m) { return 256 * m + 4 * (tau & 63) + (tau >> 6); }
    // accumulator copy (N = 2048 coefficients) stored "transposed", slot = (j mod 4) * 512 + j / 4, so
    // that the rotation gather of a wave (coefficients 4 apart) reads consecutive slots
    __host__ __device__ static constexpr int acc_slot(int j) { return ((j & 3) << 9) | (j >> 2); }
};

// P = 512 points, 4 per thread, 128 threads (2 waves): the same scheme for N = 1024 (round 4: the dense kernel of the
// k = 2 parameter sets, whose LDS pipe -- every inter-pass exchange and the twiddle table of the generic plan -- was
// what bound it).  j = 128 m + 2 lane + wave; four radix-4 passes on (b8b7), (b6b5), (b4b3), (b2b1) with the two
// register/lane swaps and the wave-local exchange of FftSwap10, then ONE radix-2 pass on b0 across the two waves:
//   pass 4 -> k3, all four registers to the wave's slab rows (workgroup barrier)
//   pass 5 -> k4: the thread of wave w takes rows r = 2w, 2w+1 of BOTH waves -- two butterflies, four outputs
//             (regs (r & 1, k4) | wave r >> 1 | lane(5,4) k2 | lane(3..0) k0 k1)
// Plane rows: 8 of 64 slots (wave w owns rows 4w..4w+3).  Used through fft_forward / fft_inverse only (one polynomial
// at a time); the split kernel's spectrum hand-over is FftSwap10's.
struct FftSwap9 {
    static constexpr bool SWAP = true;
    static constexpr int LOGW = 1;
    static constexpr int LOGP = 9;
    static constexpr int LOGR = 2;
    static constexpr int P = 512;
    static constexpr int R = 4;
    static constexpr int T = 128;
    static constexpr int FULL = 4;
    static constexpr int LOGLAST = 1;
    static constexpr int NP = 5;
    static constexpr int NTW = 4;
    __host__ __device__ static constexpr int log_radix(int s) { return s < 4 ? 2 : 1; }
    __host__ __device__ static constexpr int log_S(int s) { return 9 - 2 * s; }
    __host__ __device__ static constexpr int point(int tau, int m) { return 128 * m + 2 * (tau & 63) + (tau >> 6); }
    // accumulator copy (N = 1024 coefficients): slot = (j mod 2) * 512 + j / 2 (a wave's gather reads consecutive slots)
    __host__ __device__ static constexpr int acc_slot(int j) { return ((j & 1) << 9) | (j >> 1); }
};

// P = 2048 points, 4 per thread, 512 threads (8 waves): N = 4096 (round 4).  The generic plan moves every point through LDS
// five times per transform -- 655 KB of LDS stores per CMUX step against ~85 B/clk/CU for ds_write_b64: the LDS pipe was
// busy half of the time, VALU a third.  Here j = 512 m + 8 lane + wave; radix-4 passes on (b10b9)(b8b7)(b6b5)(b4b3) with
// FftSwap10's two register/lane swaps and wave-local exchange, then ONE radix-8 pass on (b2b1b0) across the eight waves:
//   all four registers to the wave's slab rows 4w + r (32 rows of 64 slots; workgroup barrier), then the thread of wave w
//   takes r = w >> 1 and the output parity h = w & 1: it reads the 8 inputs of (r, lane), forms u[n'] = x[n'] + x[n'+4]
//   (h = 0) or (x[n'] - x[n'+4]) w8^n' (h = 1) and runs one radix-4 on them: outputs X[2q + h] in register q.
// The inverse mirrors it: radix-4 on the thread's four X[2q + h], times conj(w8^n') for h = 1, into rows 16 h + 4 n' + r;
// after the barrier wave n adds (b2 = 0) or subtracts (b2 = 1) the even and odd halves of its own four points while loading.
// Two waves share every row they read across the barrier, so -- unlike FftSwap10's slab/comb -- a barrier separates the
// forward transform's last loads from the inverse's first stores, and the inverse's first loads from its next stores.
struct FftSwap11 {
    static constexpr bool SWAP = true;
    static constexpr int LOGW = 3;
    static constexpr int LOGP = 11;
    static constexpr int LOGR = 2;
    static constexpr int P = 2048;
    static constexpr int R = 4;
    static constexpr int T = 512;
    static constexpr int FULL = 4;
    static constexpr int LOGLAST = 3;
    static constexpr int NP = 5;
    static constexpr int NTW = 4;
    __host__ __device__ static constexpr int log_radix(int s) { return s < 4 ? 2 : 3; }
    __host__ __device__ static constexpr int log_S(int s) { return 11 - 2 * s; }
    __host__ __device__ static constexpr int point(int tau, int m) { return 512 * m + 8 * (tau & 63) + (tau >> 6); }
    // accumulator copy (N = 4096 coefficients): slot = (j mod 8) * 512 + j / 8
    __host__ __device__ static constexpr int acc_slot(int j) { return ((j & 7) << 9) | (j >> 3); }
};

// Plan used by the blind-rotation kernels for (log2 P, log2 R).
template <int LP, int LR>
struct PlanFor { using type = FftPlan<LP, LR>; };
#ifndef FHESTR_NO_SWAP_FFT
template <>
struct PlanFor<10, 2> { using type = FftSwap10; };
#endif

// ---- LDS slot swizzle (8-byte slots) ----------------------------------------------------------
// ds_read_b64 serves a wave as two 32-lane groups (32 distinct slots mod 32 = conflict free),
// ds_write_b64 as four 16-lane groups (16 distinct slots mod 16).  The in-place exchange touches
// each layout with both, and later passes stride by powers of two, so the plain index conflicts
// 2-8 ways.  Fix: a GF(2)-linear map of the low five slot bits, slot' = (a & ~31) | XOR_j a_j*col[j];
// the column constants come from scripts/find_swizzle.py (exhaustive check of every pass layout).
// Linear => slot'(base ^ c) = slot'(base) ^ slot'(c): the per-register part folds to a constant.
template <int LP, int LR>
struct SwzCols {
    static constexpr bool identity = true;
    static constexpr int col[16] = {1, 2, 4, 8, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
#define FHE_SWZ(LP, LR, ...)                                     \
    template <>                                                  \
    struct SwzCols<LP, LR> {                                     \
        static constexpr bool identity = false;                  \
        static constexpr int col[16] = {__VA_ARGS__};            \
    };
FHE_SWZ(10, 2, 5, 26, 3, 8, 9, 23, 16, 8, 10, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(10, 3, 9, 4, 17, 2, 3, 24, 5, 16, 3, 12, 0, 0, 0, 0, 0, 0)
FHE_SWZ(10, 4, 6, 9, 24, 3, 2, 1, 14, 4, 21, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(9, 3, 5, 2, 4, 26, 8, 4, 29, 16, 1, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(9, 2, 5, 24, 2, 6, 17, 11, 16, 2, 2, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(8, 2, 10, 1, 25, 4, 12, 3, 16, 8, 0, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(7, 2, 2, 4, 10, 25, 1, 5, 16, 0, 0, 0, 0, 0, 0, 0, 0, 0)
FHE_SWZ(11, 2, 18, 28, 9, 10, 1, 4, 16, 4, 8, 2, 9, 0, 0, 0, 0, 0)
// (12, 3) -- N = 8192, pbs_seq_kernels.hip.h -- has a conflict-free map too (11, 6, 18, 10, 5, 2, 12, 16, 16, 4, 13, 20), but with it that
// kernel is slower (22.4 vs 18.6 ms: 36 % of its LDS cycles are conflicts, the address arithmetic and registers cost more)
// 128 points, 8 per thread, 16 threads per transform, transforms 16 slots (mod 32) apart -- the
// large-N kernels' sub-transforms.  Derived by hand: the three pass layouts leave address bits
// {0,1,2,3}, {0,4,5,6} and {3,4,5,6} free across the 16 lanes of a transform; the low four slot bits
// (a0^a6, a1^a4, a2^a5, a3^a6) are a bijection of each set, bit 4 is kept, so the two transforms of a
// 32-lane group (offset 16) never meet.
FHE_SWZ(7, 3, 1, 2, 4, 8, 18, 4, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0)
#undef FHE_SWZ

template <class PL>
__device__ __forceinline__ int lds_slot(int a) {
    using SW = SwzCols<PL::LOGP, PL::LOGR>;
    if (SW::identity) return a;
    int bank = 0;
#pragma unroll
    for (int j = 0; j < PL::LOGP; j++) bank ^= (-((a >> j) & 1)) & SW::col[j];
    return (a & ~31) | bank;
}

// Element address (in points) handled by thread tau, group gi, butterfly input m in pass s.
// Full-radix passes: tau = Q*S_{s+1} + t'.  Trailing (grouped) pass: pair index = tau*groups + gi,
// so that the exchange feeding it stays inside S_{s+1}-thread neighbourhoods (wave-local).
template <class PL>
__device__ __forceinline__ int pass_addr(int s, int tau, int gi, int m) {
    const int lr = PL::log_radix(s);
    const int lS = PL::log_S(s);
    const int lS1 = lS - lr;                 // log2 S_{s+1}
    const int lg = PL::LOGR - lr;            // log2 groups
    const int pi = (tau << lg) + gi;         // (Q, t') pair index
    const int Q = pi >> lS1;
    const int tp = pi & ((1 << lS1) - 1);
    return (Q << lS) + (m << lS1) + tp;
}

// Synchronise the threads that exchange data between pass s and pass s+1.  Those are aligned
// neighbourhoods of S_{s+1} threads; when that fits one wavefront, in-order LDS execution of a
// single wave makes a workgroup barrier unnecessary (only the compiler must not reorder).
template <class PL>
__device__ __forceinline__ void exchange_sync(int s_next) {
    if ((1 << PL::log_S(s_next)) <= 64 && PL::T >= 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Per-thread constants of one polynomial group.
template <class PL>
struct FftConsts {
    cplx tw[PL::NTW > 0 ? PL::NTW : 1][PL::R];  // tw[s][q] = exp(-2 pi i q t' / S_s); tw[s][0] = 1 unused
    __device__ __forceinline__ cplx get(int s, int q) const { return tw[s][q]; }
};

// The same constants kept in an LDS table shared by the workgroup instead of 4*NTW*R VGPRs per
// thread (the large-N kernels need the registers for loads in flight).  Entry (s, q, t') sits at
// off(s) + q * n_tp(s) + t': for a fixed q the threads of a transform read consecutive 16-byte
// entries (conflict free), threads of different transforms broadcast.
template <class PL>
struct FftTwiddleTable {
    __host__ __device__ static constexpr int n_tp(int s) { return 1 << (PL::log_S(s) - PL::LOGR); }
    __host__ __device__ static constexpr int off(int s) {
        int o = 0;
        for (int i = 0; i < s; i++) o += PL::R * n_tp(i);
        return o;
    }
    static constexpr int ENTRIES = off(PL::NTW);
    const double2* base;
    int tau;
    __device__ __forceinline__ static void fill(double2* b, int tid, int nthreads) {
#pragma unroll
        for (int s = 0; s < PL::NTW; s++) {
            const int ntp = n_tp(s);
            for (int e = tid; e < PL::R * ntp; e += nthreads) {
                const int q = e / ntp, tp = e % ntp;
                double sn, cs;
                sincospi(-2.0 * (double)(q * tp) / (double)(1 << PL::log_S(s)), &sn, &cs);
                b[off(s) + e] = make_double2(cs, sn);
            }
        }
    }
    __device__ __forceinline__ cplx get(int s, int q) const {
        const double2 v = base[off(s) + q * n_tp(s) + (tau & (n_tp(s) - 1))];
        cplx r; r.re = v.x; r.im = v.y;
        return r;
    }
};

// Pass 0's twiddles in VGPRs, the later passes' (few distinct values: n_tp shrinks by the radix every pass) in an
// LDS table -- for plans with many passes, where NTW * R complex constants per thread push the kernel into
// spilling (N = 4096: five radix-4 passes, 80 VGPRs).  Table layout as FftTwiddleTable, without pass 0.
template <class PL>
struct FftHybridConsts {
    using TBL = FftTwiddleTable<PL>;
    static constexpr int ENTRIES = TBL::ENTRIES - TBL::off(1);
    cplx tw0[PL::R];
    const double2* base;     // entries of passes 1 .. NTW-1
    int tau;
    __device__ __forceinline__ static void fill(double2* b, int tid, int nthreads) {
#pragma unroll
        for (int s = 1; s < PL::NTW; s++) {
            const int ntp = TBL::n_tp(s);
            for (int e = tid; e < PL::R * ntp; e += nthreads) {
                const int q = e / ntp, tp = e % ntp;
                double sn, cs;
                sincospi(-2.0 * (double)(q * tp) / (double)(1 << PL::log_S(s)), &sn, &cs);
                b[TBL::off(s) - TBL::off(1) + e] = make_double2(cs, sn);
            }
        }
    }
    __device__ __forceinline__ void init(const double2* table, int tau_) {
        base = table;
        tau = tau_;
        const int tp = tau_ & (TBL::n_tp(0) - 1);
#pragma unroll
        for (int q = 0; q < PL::R; q++) {
            double sn, cs;
            sincospi(-2.0 * (double)(q * tp) / (double)(1 << PL::log_S(0)), &sn, &cs);
            tw0[q].re = cs; tw0[q].im = sn;
        }
    }
    __device__ __forceinline__ cplx get(int s, int q) const {
        if (s == 0) return tw0[q];
        const double2 v = base[TBL::off(s) - TBL::off(1) + q * TBL::n_tp(s) + (tau & (TBL::n_tp(s) - 1))];
        cplx r; r.re = v.x; r.im = v.y;
        return r;
    }
};

template <class PL>
__device__ __forceinline__ void fft_init_consts(FftConsts<PL>& c, int tau) {
#pragma unroll
    for (int s = 0; s < PL::NTW; s++) {
        const int lS = PL::log_S(s);
        const int lS1 = lS - PL::LOGR;
        int tp = tau & ((1 << lS1) - 1);
        if (PL::SWAP) {   // low index bits still to be transformed after pass s (see FftSwap10)
            const int lane = tau & 63;
            int w = tau >> 6;
            // FftSwap9, last twiddled pass: the angle depends on the wave number alone -- scalar registers (the dense
            // kernel has none of the vector kind to spare)
            if constexpr (PL::LOGP == 9 || PL::LOGP == 11) { if (s == 3) w = __builtin_amdgcn_readfirstlane(w); }
            constexpr int LW = PL::SWAP ? (PL::LOGP - 8) : 0;       // FftSwap10: 2 wave bits, FftSwap9: 1
            tp = s == 0 ? ((lane << LW) | w) : s == 1 ? (((lane & 15) << LW) | w) : s == 2 ? (((lane >> 4) << LW) | w) : w;
        }
#pragma unroll
        for (int q = 0; q < PL::R; q++) {
            double sn, cs;
            // angle = -2*pi*q*tp / S_s  ->  sincospi(-2*q*tp/S_s)
            sincospi(-2.0 * (double)(q * tp) / (double)(1 << lS), &sn, &cs);
            c.tw[s][q].re = cs;
            c.tw[s][q].im = sn;
        }
    }
}

#ifdef FHESTR_NO_PIN
#define FHE_PIN_ORDER() do {} while (0)
#else
#define FHE_PIN_ORDER() __builtin_amdgcn_sched_barrier(0)
#endif
// ---- FftSwap10 building blocks --------------------------------------------------------------
__device__ __forceinline__ void wave_local_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool ROW16>
__device__ __forceinline__ void swap_halves(double& a, double& b) {
    // ROW16: a[lanes with bit 4 set] <-> b[lanes with bit 4 clear]; else the same on lane bit 5
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a);
    const unsigned blo = __double2loint(b), bhi = __double2hiint(b);
    if (ROW16) {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = __hiloint2double(h[0], l[0]);
        b = __hiloint2double(h[1], l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = __hiloint2double(h[0], l[0]);
        b = __hiloint2double(h[1], l[1]);
    }
}
// 4x4 transpose between register index (2 bits) and lane bits (5,4); its own inverse.
__device__ __forceinline__ void swap_regs_lanes(cplx* x) {
    swap_halves<false>(x[0].re, x[2].re); swap_halves<false>(x[0].im, x[2].im);
    swap_halves<false>(x[1].re, x[3].re); swap_halves<false>(x[1].im, x[3].im);
    swap_halves<true>(x[0].re, x[1].re);  swap_halves<true>(x[0].im, x[1].im);
    swap_halves<true>(x[2].re, x[3].re);  swap_halves<true>(x[2].im, x[3].im);
}
// slab <-> comb transpose (regs <-> wave).  Slot = row * 64 + lane: conflict free as is.
__device__ __forceinline__ int swap10_slab(int tau, int r) { return (((tau >> 6) * 4 + r) << 6) | (tau & 63); }
__device__ __forceinline__ int swap10_comb(int tau, int r) { return ((r * 4 + (tau >> 6)) << 6) | (tau & 63); }
// Wave-local exchange inside the wave's slab rows.  Side A: regs k1 | lane(5,4) k0 | lane(3..0) b5..b2;
// side B: regs b5b4 | lane(5,4) b3b2 | lane(3..0) k0hi k0lo k1hi k1lo.  With (u3..u0) = b5..b2, slot
// within the 4 slab rows:
//   c0 = u0^k0lo  c1 = u1^k0hi  c2 = u2^k1hi  c3 = u3^k1lo  c4 = k0lo  c5 = k0hi  row = k1
// -- every 16-lane group of either side covers 16 distinct slots mod 16 and every 32-lane group 32
// distinct slots mod 32 (the ds_write_b64 / ds_read_b64 conflict-free conditions).
__device__ __forceinline__ int swap10_side_a(int tau, int r) {
    const int lane = tau & 63, w = tau >> 6;
    const int c = lane ^ ((lane >> 4) & 1) ^ (((lane >> 5) & 1) << 1) ^ ((r >> 1) << 2) ^ ((r & 1) << 3);
    return ((4 * w + r) << 6) | c;
}
__device__ __forceinline__ int swap10_side_b(int tau, int r) {
    const int lane = tau & 63, w = tau >> 6;
    const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1;
    const int l4 = (lane >> 4) & 1, l5 = (lane >> 5) & 1;
    const int c = (l4 ^ l2) | ((l5 ^ l3) << 1) | (((r & 1) ^ l1) << 2) | (((r >> 1) ^ l0) << 3) | (l2 << 4) | (l3 << 5);
    return ((4 * w + (lane & 3)) << 6) | c;
}
// Size-4 DFT (natural order) that hands every output to emit(q, y) the moment it exists, in the
// order 0, 2, 1, 3, so stores can start while the remaining outputs are still being added up.
template <bool INV, class Emit>
__device__ __forceinline__ void dft4_emit(const cplx* x, Emit emit) {
    cplx u0, u1, d0, d1, y;
    u0.re = x[0].re + x[2].re; u0.im = x[0].im + x[2].im;
    u1.re = x[1].re + x[3].re; u1.im = x[1].im + x[3].im;
    d0.re = x[0].re - x[2].re; d0.im = x[0].im - x[2].im;
    d1.re = x[1].re - x[3].re; d1.im = x[1].im - x[3].im;
    y.re = u0.re + u1.re; y.im = u0.im + u1.im; emit(0, y);
    y.re = u0.re - u1.re; y.im = u0.im - u1.im; emit(2, y);
    // forward: d1 * (-i) = (d1.im, -d1.re); inverse: d1 * (+i) = (-d1.im, d1.re)
    if (!INV) { y.re = d0.re + d1.im; y.im = d0.im - d1.re; } else { y.re = d0.re - d1.im; y.im = d0.im + d1.re; }
    emit(1, y);
    if (!INV) { y.re = d0.re - d1.im; y.im = d0.im + d1.re; } else { y.re = d0.re + d1.im; y.im = d0.im - d1.re; }
    emit(3, y);
}
// Inverse pass with its twiddles folded in: y = IDFT4(x0, x1 conj(w1), x2 conj(w2), x3 conj(w3)) in natural order.  The
// difference of a butterfly whose second operand is a product, a - w b, is taken as 2 a - (a + w b): the sum costs the four
// multiply-adds the product alone would, the difference two -- 24 f64 instructions instead of 12 (three products) + 16
// (butterflies); 32 of the ~1,290 instructions of a P22 CMUX step.  emit(q, y) as in dft4_emit (order 0, 2, 1, 3).
template <class Emit>
__device__ __forceinline__ void idft4_twiddled_emit(const cplx* x, const cplx* tw, Emit emit) {
    cplx t1, u0, u1, d0, d1, y;
    u0.re = fma(x[2].re, tw[2].re, fma(x[2].im, tw[2].im, x[0].re));
    u0.im = fma(x[2].im, tw[2].re, fma(-x[2].re, tw[2].im, x[0].im));
    d0.re = fma(2.0, x[0].re, -u0.re);
    d0.im = fma(2.0, x[0].im, -u0.im);
    t1 = cmul_conj(x[1], tw[1]);
    u1.re = fma(x[3].re, tw[3].re, fma(x[3].im, tw[3].im, t1.re));
    u1.im = fma(x[3].im, tw[3].re, fma(-x[3].re, tw[3].im, t1.im));
    d1.re = fma(2.0, t1.re, -u1.re);
    d1.im = fma(2.0, t1.im, -u1.im);
    y.re = u0.re + u1.re; y.im = u0.im + u1.im; emit(0, y);
    y.re = u0.re - u1.re; y.im = u0.im - u1.im; emit(2, y);
    y.re = d0.re - d1.im; y.im = d0.im + d1.re; emit(1, y);      // d1 * (+i)
    y.re = d0.re + d1.im; y.im = d0.im - d1.re; emit(3, y);
}
__device__ __forceinline__ void idft4_twiddled(cplx* x, const cplx* tw) {
    cplx y[4];
    idft4_twiddled_emit(x, tw, [&](int q, cplx v) { y[q] = v; });
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = y[q];
}
#ifndef FHESTR_FUSED_IDFT
#define FHESTR_FUSED_IDFT 1
#endif

template <bool INV>
__device__ __forceinline__ void swap10_twiddle(cplx* x, const cplx* tw) {
#pragma unroll
    for (int q = 1; q < 4; q++) x[q] = INV ? cmul_conj(x[q], tw[q]) : cmul(x[q], tw[q]);
}

// Forward transform in two halves around its one workgroup barrier.
//   head: passes 1-4 of this thread's polynomial, left in the slab side of the (re, im) planes
//   tail: after the barrier, the comb side of ANY polynomial's planes -> last pass -> x[rho]
// The head in three stages separated by the wave-local fences, so that callers carrying several
// polynomials per thread can run stage by stage over all of them (one polynomial's LDS round trip
// then hides behind the other's butterflies).
template <class C>
__device__ __forceinline__ void swap10_fwd_stage1(cplx* x, const C& c, double* re, double* im, int tau) {
    small_dft<4, false>(x);
    swap10_twiddle<false>(x, c.tw[0]);
    swap_regs_lanes(x);
    small_dft<4, false>(x);
    // twiddle and store point by point: the next point's multiplies issue while the LDS write port
    // drains the previous one
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r) x[r] = cmul(x[r], c.tw[1][r]);
        const int a = swap10_side_a(tau, r); re[a] = x[r].re; im[a] = x[r].im;
        FHE_PIN_ORDER();
    }
}
template <class C>
__device__ __forceinline__ void swap10_fwd_stage2(cplx* x, const C& c, const double* re, const double* im, int tau) {
#pragma unroll
    for (int r = 0; r < 4; r++) { const int a = swap10_side_b(tau, r); x[r].re = re[a]; x[r].im = im[a]; }
    small_dft<4, false>(x);
    swap10_twiddle<false>(x, c.tw[2]);
    swap_regs_lanes(x);
    small_dft<4, false>(x);
}
template <class C>
__device__ __forceinline__ void swap10_fwd_stage3(cplx* x, const C& c, double* re, double* im, int tau) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r) x[r] = cmul(x[r], c.tw[3][r]);
        const int a = swap10_slab(tau, r); re[a] = x[r].re; im[a] = x[r].im;
        FHE_PIN_ORDER();
    }
}
template <class C>
__device__ __forceinline__ void swap10_forward_head(cplx* x, const C& c, double* re, double* im, int tau) {
    swap10_fwd_stage1(x, c, re, im, tau);
    wave_local_fence();
    swap10_fwd_stage2(x, c, re, im, tau);
    wave_local_fence();          // the slab stores below reuse the rows the exchange above read
    swap10_fwd_stage3(x, c, re, im, tau);
}
__device__ __forceinline__ void swap10_forward_tail(cplx* x, const double* re, const double* im, int tau) {
#pragma unroll
    for (int r = 0; r < 4; r++) { const int a = swap10_comb(tau, r); x[r].re = re[a]; x[r].im = im[a]; }
    small_dft<4, false>(x);
}
// Inverse: head = first pass + comb stores; barrier; tail = slab loads + the remaining passes.
__device__ __forceinline__ void swap10_inverse_head(const cplx* x, double* re, double* im, int tau) {
    dft4_emit<true>(x, [&](int r, cplx y) {
        const int a = swap10_comb(tau, r); re[a] = y.re; im[a] = y.im;
        FHE_PIN_ORDER();
    });
}
template <class C>
__device__ __forceinline__ void swap10_inv_stage1_compute(cplx* x, const C& c);
template <class C>
__device__ __forceinline__ void swap10_inv_stage1(cplx* x, const C& c, const double* re, const double* im, int tau) {
#pragma unroll
    for (int r = 0; r < 4; r++) { const int a = swap10_slab(tau, r); x[r].re = re[a]; x[r].im = im[a]; }
    swap10_inv_stage1_compute(x, c);
}
template <class C>
__device__ __forceinline__ void swap10_inv_stage1_compute(cplx* x, const C& c) {
    if (FHESTR_FUSED_IDFT) {
        idft4_twiddled(x, c.tw[3]);
        swap_regs_lanes(x);           // the twiddles of the next pass (c.tw[2]) are folded into stage 2's butterfly
        return;
    }
    swap10_twiddle<true>(x, c.tw[3]);
    small_dft<4, true>(x);
    swap_regs_lanes(x);
    swap10_twiddle<true>(x, c.tw[2]);
}
template <class C>
__device__ __forceinline__ void swap10_inv_stage2(const cplx* x, const C& c, double* re, double* im, int tau) {
    auto store = [&](int r, cplx y) {
        const int a = swap10_side_b(tau, r); re[a] = y.re; im[a] = y.im;
        FHE_PIN_ORDER();
    };
    if (FHESTR_FUSED_IDFT) idft4_twiddled_emit(x, c.tw[2], store);
    else dft4_emit<true>(x, store);
}
template <class C>
__device__ __forceinline__ void swap10_inv_stage3(cplx* x, const C& c, const double* re, const double* im, int tau) {
#pragma unroll
    for (int r = 0; r < 4; r++) { const int a = swap10_side_a(tau, r); x[r].re = re[a]; x[r].im = im[a]; }
    if (FHESTR_FUSED_IDFT) {
        idft4_twiddled(x, c.tw[1]);
        swap_regs_lanes(x);
        idft4_twiddled(x, c.tw[0]);
        return;
    }
    swap10_twiddle<true>(x, c.tw[1]);
    small_dft<4, true>(x);
    swap_regs_lanes(x);
    swap10_twiddle<true>(x, c.tw[0]);
    small_dft<4, true>(x);
}
template <class C>
__device__ __forceinline__ void swap10_inverse_tail(cplx* x, const C& c, double* re, double* im, int tau) {
    swap10_inv_stage1(x, c, re, im, tau);
    wave_local_fence();          // the exchange below reuses the slab rows just read
    swap10_inv_stage2(x, c, re, im, tau);
    wave_local_fence();
    swap10_inv_stage3(x, c, re, im, tau);
}

// NPOLY polynomials carried by the same threads (planes of polynomial p at re0 + p*poly_stride,
// imaginary plane im_off slots further): one workgroup barrier serves all of them.
template <int NPOLY>
__device__ __forceinline__ void swap10_forward(cplx (*x)[4], const FftConsts<FftSwap10>& c, double* re0,
                                               int poly_stride, int im_off, int tau) {
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage1(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_forward_tail(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
}
template <int NPOLY>
__device__ __forceinline__ void swap10_inverse(cplx (*x)[4], const FftConsts<FftSwap10>& c, double* re0,
                                               int poly_stride, int im_off, int tau) {
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inverse_head(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage1(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
}

#ifdef FHESTR_ABLATE_DENSE_BARRIERS      // timing experiment only (wrong results): where does the dense kernel's time go?
#define FHE_DENSE_SYNC() do {} while (0)
#else
#define FHE_DENSE_SYNC() __syncthreads()
#endif
// FftSwap9: passes 1-4 and both in-wave exchanges are FftSwap10's stages (slab rows 4w + r, two waves); the pass across the
// waves is a radix-2.
// The pieces, for callers that put independent work of their own between them (the dense kernel: the previous polynomial's
// products behind stage 1's LDS stores, its accumulator update behind the inverse's first stores).
__device__ __forceinline__ void swap9_forward_tail(cplx* x, const double* re, const double* im, int tau) {
    const int lane = tau & 63, w = tau >> 6;
#pragma unroll
    for (int rl = 0; rl < 2; rl++) {
        const int a0 = ((2 * w + rl) << 6) | lane, a1 = ((4 + 2 * w + rl) << 6) | lane;     // b0 = 0 / b0 = 1
        const double ar = re[a0], ai = im[a0], br = re[a1], bi = im[a1];
        x[2 * rl].re = ar + br; x[2 * rl].im = ai + bi;
        x[2 * rl + 1].re = ar - br; x[2 * rl + 1].im = ai - bi;
    }
}
__device__ __forceinline__ void swap9_inverse_head(const cplx* x, double* re, double* im, int tau) {
    const int lane = tau & 63, w = tau >> 6;
#pragma unroll
    for (int rl = 0; rl < 2; rl++) {
        const int a0 = ((2 * w + rl) << 6) | lane, a1 = ((4 + 2 * w + rl) << 6) | lane;
        re[a0] = x[2 * rl].re + x[2 * rl + 1].re; im[a0] = x[2 * rl].im + x[2 * rl + 1].im;
        FHE_PIN_ORDER();
        re[a1] = x[2 * rl].re - x[2 * rl + 1].re; im[a1] = x[2 * rl].im - x[2 * rl + 1].im;
        FHE_PIN_ORDER();
    }
}
template <class C>
__device__ __forceinline__ void swap9_inverse_tail(cplx* x, const C& c, double* re, double* im, int tau) {
    swap10_inv_stage1(x, c, re, im, tau);
    wave_local_fence();          // the exchange below reuses the slab rows just read
    swap10_inv_stage2(x, c, re, im, tau);
    wave_local_fence();
    swap10_inv_stage3(x, c, re, im, tau);
}
// NPOLY polynomials carried by the same threads, stage by stage (one polynomial's LDS round trip behind the others' butterflies)
template <int NPOLY>
__device__ __forceinline__ void swap9_forward_multi(cplx (*x)[4], const FftConsts<FftSwap9>& c, double* re0, int poly_stride, int im_off, int tau) {
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage1(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap9_forward_tail(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();             // the other wave reads this wave's rows in its tail: nobody stores into them before both have read
}
template <int NPOLY>
__device__ __forceinline__ void swap9_inverse_multi(cplx (*x)[4], const FftConsts<FftSwap9>& c, double* re0, int poly_stride, int im_off, int tau) {
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap9_inverse_head(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage1(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
}
__device__ __forceinline__ void swap9_forward(cplx* x, const FftConsts<FftSwap9>& c, double* re, double* im, int tau) {
    swap10_fwd_stage1(x, c, re, im, tau);
    wave_local_fence();
    swap10_fwd_stage2(x, c, re, im, tau);
    wave_local_fence();          // the slab stores below reuse the rows the exchange above read
    swap10_fwd_stage3(x, c, re, im, tau);
    FHE_DENSE_SYNC();
    swap9_forward_tail(x, re, im, tau);
}
__device__ __forceinline__ void swap9_inverse(cplx* x, const FftConsts<FftSwap9>& c, double* re, double* im, int tau) {
    swap9_inverse_head(x, re, im, tau);
    FHE_DENSE_SYNC();
    swap9_inverse_tail(x, c, re, im, tau);
}

// Swap-plan constants with the twiddles of passes 1 and 2 in an LDS table (few distinct values: tp < S/4) and fetched into
// registers just before the stage that multiplies by them; pass 0's stay in VGPRs, pass 3's are wave-uniform (scalar).  For the
// kernels whose register file is full without them (N = 4096: 36 VGPRs of loop-invariant twiddles were what spilled).
// Table: [s = 1][q - 1][tp < S1 / 4], then [s = 2][q - 1][tp < S2 / 4].
template <class PL>
struct FftSwapLdsConsts {
    static constexpr int N1 = 1 << (PL::log_S(1) - 2), N2 = 1 << (PL::log_S(2) - 2);
    static constexpr int ENTRIES = 3 * N1 + 3 * N2;
    mutable cplx tw[4][4];
    const double2* table;
    int tp1, tp2;
    __device__ __forceinline__ static void fill(double2* b, int tid, int nthreads) {
        for (int e = tid; e < ENTRIES; e += nthreads) {
            const bool second = e >= 3 * N1;
            const int i = second ? e - 3 * N1 : e, n = second ? N2 : N1, lS = second ? PL::log_S(2) : PL::log_S(1);
            const int q = i / n + 1, tp = i % n;
            double sn, cs;
            sincospi(-2.0 * (double)(q * tp) / (double)(1 << lS), &sn, &cs);
            b[e] = make_double2(cs, sn);
        }
    }
    __device__ __forceinline__ void init(const double2* t, int tau) {
        table = t;
        const int lane = tau & 63, w = tau >> 6;
        tp1 = ((lane & 15) << PL::LOGW) | w;
        tp2 = ((lane >> 4) << PL::LOGW) | w;
        const int tp0 = (lane << PL::LOGW) | w, ws = __builtin_amdgcn_readfirstlane(w);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double sn, cs;
            sincospi(-2.0 * (double)(q * tp0) / (double)(1 << PL::log_S(0)), &sn, &cs);
            tw[0][q].re = cs; tw[0][q].im = sn;
            sincospi(-2.0 * (double)(q * ws) / (double)(1 << PL::log_S(3)), &sn, &cs);
            tw[3][q].re = cs; tw[3][q].im = sn;
        }
    }
    __device__ __forceinline__ cplx get(int s, int q) const { return tw[s][q]; }      // (generic code paths never taken by swap plans)
    __device__ __forceinline__ void load(int s) const {          // s = 1 or 2
#pragma unroll
        for (int q = 1; q < 4; q++) {
            const double2 v = s == 1 ? table[(q - 1) * N1 + tp1] : table[3 * N1 + (q - 1) * N2 + tp2];
            tw[s][q].re = v.x; tw[s][q].im = v.y;
        }
    }
};
template <class PL> __device__ __forceinline__ void swap_load_tw(const FftConsts<PL>&, int) {}
template <class PL> __device__ __forceinline__ void swap_load_tw(const FftSwapLdsConsts<PL>& c, int s) { c.load(s); }

// ---- FftSwap11 (2048 points, eight waves) ------------------------------------------------------------------------
__device__ __forceinline__ void swap11_forward_tail(cplx* x, const double* re, const double* im, int tau) {
    const int lane = tau & 63, w = __builtin_amdgcn_readfirstlane(tau >> 6), r = w >> 1;
    cplx in[8];
#pragma unroll
    for (int n = 0; n < 8; n++) { const int a = ((4 * n + r) << 6) | lane; in[n].re = re[a]; in[n].im = im[a]; }
    constexpr double C8 = 0.70710678118654752440;
    if ((w & 1) == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) { x[q].re = in[q].re + in[q + 4].re; x[q].im = in[q].im + in[q + 4].im; }
    } else {                                       // (x[n'] - x[n'+4]) * exp(-2 pi i n' / 8)
        cplx d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { d[q].re = in[q].re - in[q + 4].re; d[q].im = in[q].im - in[q + 4].im; }
        x[0] = d[0];
        x[1].re = (d[1].re + d[1].im) * C8; x[1].im = (d[1].im - d[1].re) * C8;
        x[2].re = d[2].im;                  x[2].im = -d[2].re;
        x[3].re = (d[3].im - d[3].re) * C8; x[3].im = -(d[3].re + d[3].im) * C8;
    }
    small_dft<4, false>(x);
}
__device__ __forceinline__ void swap11_inverse_head(const cplx* x, double* re, double* im, int tau) {
    const int lane = tau & 63, w = __builtin_amdgcn_readfirstlane(tau >> 6), r = w >> 1, h = w & 1;
    cplx v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = x[q];
    small_dft<4, true>(v);
    constexpr double C8 = 0.70710678118654752440;
    if (h) {                                       // * exp(+2 pi i n' / 8)
        cplx t;
        t.re = (v[1].re - v[1].im) * C8; t.im = (v[1].re + v[1].im) * C8; v[1] = t;
        t.re = -v[2].im;                 t.im = v[2].re;                  v[2] = t;
        t.re = -(v[3].re + v[3].im) * C8; t.im = (v[3].re - v[3].im) * C8; v[3] = t;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int a = ((16 * h + 4 * q + r) << 6) | lane;
        re[a] = v[q].re; im[a] = v[q].im;
        FHE_PIN_ORDER();
    }
}
// wave n = 4 b2 + n': its points r = 0..3 are E[n'][r] +- O[n'][r]
__device__ __forceinline__ void swap11_inverse_gather(cplx* x, const double* re, const double* im, int tau) {
    const int lane = tau & 63, w = __builtin_amdgcn_readfirstlane(tau >> 6), np = w & 3;
    const double sg = (w >> 2) ? -1.0 : 1.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = ((4 * np + r) << 6) | lane, b = ((16 + 4 * np + r) << 6) | lane;
        x[r].re = fma(sg, re[b], re[a]);
        x[r].im = fma(sg, im[b], im[a]);
    }
}
template <int NPOLY, class C>
__device__ __forceinline__ void swap11_forward(cplx (*x)[4], const C& c, double* re0, int poly_stride, int im_off, int tau) {
    swap_load_tw(c, 1);
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage1(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
    swap_load_tw(c, 2);
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_fwd_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) {
        swap11_forward_tail(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
        FHE_PIN_ORDER();         // one polynomial's eight inputs in registers at a time
    }
    __syncthreads();             // rows are shared by pairs of waves: nobody stores into them before everybody has read
}
template <int NPOLY, class C>
__device__ __forceinline__ void swap11_inverse(cplx (*x)[4], const C& c, double* re0, int poly_stride, int im_off, int tau) {
#pragma unroll
    for (int p = 0; p < NPOLY; p++) {
        swap11_inverse_head(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
        FHE_PIN_ORDER();
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap11_inverse_gather(x[p], re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    __syncthreads();             // the wave-local exchange below stores into rows other waves have just read
    swap_load_tw(c, 2);
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage1_compute(x[p], c);
    wave_local_fence();
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage2(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
    wave_local_fence();
    swap_load_tw(c, 1);
#pragma unroll
    for (int p = 0; p < NPOLY; p++) swap10_inv_stage3(x[p], c, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
}

// Forward transform.  In: x[m] = point (tau + T*m) of the (already twisted) input.
// Out: x[rho] in last-pass layout.  `re`/`im` are this group's LDS planes (P doubles each).
template <class PL, class C>
__device__ __forceinline__ void fft_forward(cplx* x, const C& c, double* re, double* im,
                                            int tau) {
    if constexpr (PL::SWAP && PL::LOGP == 9) {
        swap9_forward(x, c, re, im, tau);
        return;
    } else if constexpr (PL::SWAP && PL::LOGP == 11) {
        swap11_forward<1>(reinterpret_cast<cplx(*)[4]>(x), c, re, 0, (int)(im - re), tau);
        return;
    } else if constexpr (PL::SWAP) {
        swap10_forward<1>(reinterpret_cast<cplx(*)[4]>(x), c, re, 0, (int)(im - re), tau);
        return;
    }
    constexpr int R = PL::R;
#pragma unroll
    for (int s = 0; s < PL::NP; s++) {
        const int lr = PL::log_radix(s);
        const int rr = 1 << lr;
        const int groups = R / rr;
        if (s > 0) {
            exchange_sync<PL>(s);
#pragma unroll
            for (int gi = 0; gi < groups; gi++) {
                const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
                for (int m = 0; m < rr; m++) {
                    const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
                    x[gi * rr + m].re = re[a];
                    x[gi * rr + m].im = im[a];
                }
            }
        }
        if (lr == PL::LOGR) {
            small_dft<R, false>(x);
        } else {
#pragma unroll
            for (int gi = 0; gi < groups; gi++) small_dft<(1 << PL::LOGLAST), false>(x + gi * rr);
        }
        if (s < PL::NTW) {
#pragma unroll
            for (int q = 1; q < R; q++) x[q] = cmul(x[q], c.get(s, q));
        }
        if (s + 1 < PL::NP) {
#pragma unroll
            for (int gi = 0; gi < groups; gi++) {
                const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
                for (int m = 0; m < rr; m++) {
                    const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
                    re[a] = x[gi * rr + m].re;
                    im[a] = x[gi * rr + m].im;
                }
            }
        }
    }
}

// Inverse transform (unscaled).  In: x[rho] in last-pass layout.  Out: x[m] = point (tau + T*m).
template <class PL, class C>
__device__ __forceinline__ void fft_inverse(cplx* x, const C& c, double* re, double* im,
                                            int tau) {
    if constexpr (PL::SWAP && PL::LOGP == 9) {
        swap9_inverse(x, c, re, im, tau);
        return;
    } else if constexpr (PL::SWAP && PL::LOGP == 11) {
        swap11_inverse<1>(reinterpret_cast<cplx(*)[4]>(x), c, re, 0, (int)(im - re), tau);
        return;
    } else if constexpr (PL::SWAP) {
        swap10_inverse<1>(reinterpret_cast<cplx(*)[4]>(x), c, re, 0, (int)(im - re), tau);
        return;
    }
    constexpr int R = PL::R;
#pragma unroll
    for (int s = PL::NP - 1; s >= 0; s--) {
        const int lr = PL::log_radix(s);
        const int rr = 1 << lr;
        const int groups = R / rr;
        if (s + 1 < PL::NP) {
            exchange_sync<PL>(s + 1);
#pragma unroll
            for (int gi = 0; gi < groups; gi++) {
                const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
                for (int m = 0; m < rr; m++) {
                    const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
                    x[gi * rr + m].re = re[a];
                    x[gi * rr + m].im = im[a];
                }
            }
        }
        if (FHESTR_FUSED_IDFT && R == 4 && s < PL::NTW && lr == PL::LOGR) {      // twiddles folded into the butterflies
            cplx tw[4];
#pragma unroll
            for (int q = 1; q < 4; q++) tw[q] = c.get(s, q);
            idft4_twiddled(x, tw);
        } else {
        if (s < PL::NTW) {
#pragma unroll
            for (int q = 1; q < R; q++) x[q] = cmul_conj(x[q], c.get(s, q));
        }
        if (lr == PL::LOGR) {
            small_dft<R, true>(x);
        } else {
#pragma unroll
            for (int gi = 0; gi < groups; gi++) small_dft<(1 << PL::LOGLAST), true>(x + gi * rr);
        }
        }
        if (s > 0) {
#pragma unroll
            for (int gi = 0; gi < groups; gi++) {
                const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
                for (int m = 0; m < rr; m++) {
                    const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
                    re[a] = x[gi * rr + m].re;
                    im[a] = x[gi * rr + m].im;
                }
            }
        }
    }
}

// ---- NP polynomials per thread ------------------------------------------------------------------
// Same transforms, but every thread carries the R points of NPOLY independent polynomials (same
// indices): the pass loop is outermost so one synchronisation covers all of them and the compiler
// can overlap one polynomial's LDS round trip with another one's butterflies.  `re0` is the first
// polynomial's real plane; planes of polynomial p start at re0 + p*poly_stride, imaginary plane at
// +im_off.
// helpers: move the R points of one polynomial between registers and its LDS planes in the layout of pass s
template <class PL>
__device__ __forceinline__ void pass_load(cplx* x, int s, const double* re, const double* im, int tau) {
    const int rr = 1 << PL::log_radix(s);
    const int groups = PL::R / rr;
#pragma unroll
    for (int gi = 0; gi < groups; gi++) {
        const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
        for (int m = 0; m < rr; m++) {
            const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
            x[gi * rr + m].re = re[a];
            x[gi * rr + m].im = im[a];
        }
    }
}
template <class PL>
__device__ __forceinline__ void pass_store(const cplx* x, int s, double* re, double* im, int tau) {
    const int rr = 1 << PL::log_radix(s);
    const int groups = PL::R / rr;
#pragma unroll
    for (int gi = 0; gi < groups; gi++) {
        const int base = lds_slot<PL>(pass_addr<PL>(s, tau, gi, 0));
#pragma unroll
        for (int m = 0; m < rr; m++) {
            const int a = base ^ lds_slot<PL>(pass_addr<PL>(s, 0, 0, m));
            re[a] = x[gi * rr + m].re;
            im[a] = x[gi * rr + m].im;
        }
    }
}
template <class PL, bool INV, class C>
__device__ __forceinline__ void pass_compute(cplx* x, int s, const C& c) {
    constexpr int R = PL::R;
    const int lr = PL::log_radix(s);
    const int rr = 1 << lr;
    if (FHESTR_FUSED_IDFT && INV && R == 4 && s < PL::NTW && lr == PL::LOGR) {      // twiddles folded into the butterflies
        cplx tw[4];
#pragma unroll
        for (int q = 1; q < 4; q++) tw[q] = c.get(s, q);
        idft4_twiddled(x, tw);
        return;
    }
    if (INV && s < PL::NTW) {
#pragma unroll
        for (int q = 1; q < R; q++) x[q] = cmul_conj(x[q], c.get(s, q));
    }
    if (lr == PL::LOGR) {
        small_dft<R, INV>(x);
    } else {
#pragma unroll
        for (int gi = 0; gi < R / rr; gi++) small_dft<(1 << PL::LOGLAST), INV>(x + gi * rr);
    }
    if (!INV && s < PL::NTW) {
#pragma unroll
        for (int q = 1; q < R; q++) x[q] = cmul(x[q], c.get(s, q));
    }
}
__device__ __forceinline__ bool pass_sync_is_wave_local(int log_S_next) { return (1 << log_S_next) <= 64; }

// Forward transform of NPOLY polynomials carried by the same threads.  Exchanges that need a
// workgroup barrier are done for all polynomials at once; wave-local exchanges are software
// pipelined: polynomial p's store + the next pass's load are issued before polynomial p+1's
// butterflies, so the LDS round trip of one stream hides behind the VALU work of the other.
template <class PL, int NPOLY, class C>
__device__ __forceinline__ void fft_forward_multi(cplx (*x)[PL::R], const C& c, double* re0,
                                                  int poly_stride, int im_off, int tau) {
    if constexpr (PL::SWAP && PL::LOGP == 11) {
        swap11_forward<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    } else if constexpr (PL::SWAP && PL::LOGP == 9) {
        swap9_forward_multi<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    } else if constexpr (PL::SWAP) {
        swap10_forward<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    }
    bool loaded = true;   // pass 0 operands are already in registers
#pragma unroll
    for (int s = 0; s < PL::NP; s++) {
        const bool last = s + 1 == PL::NP;
        const bool local_next = !last && pass_sync_is_wave_local(PL::log_S(s + 1));
        if (!loaded) {
#pragma unroll
            for (int p = 0; p < NPOLY; p++) pass_load<PL>(x[p], s, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
        }
#pragma unroll
        for (int p = 0; p < NPOLY; p++) {
            pass_compute<PL, false>(x[p], s, c);
            if (!last) {
                pass_store<PL>(x[p], s, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
                if (local_next) {   // in-order LDS execution of this wave: the loads see the stores above
                    exchange_sync<PL>(s + 1);
                    pass_load<PL>(x[p], s + 1, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
                }
            }
        }
        if (!last && !local_next) exchange_sync<PL>(s + 1);
        loaded = local_next;
    }
}

template <class PL, int NPOLY, class C>
__device__ __forceinline__ void fft_inverse_multi(cplx (*x)[PL::R], const C& c, double* re0,
                                                  int poly_stride, int im_off, int tau) {
    if constexpr (PL::SWAP && PL::LOGP == 11) {
        swap11_inverse<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    } else if constexpr (PL::SWAP && PL::LOGP == 9) {
        swap9_inverse_multi<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    } else if constexpr (PL::SWAP) {
        swap10_inverse<NPOLY>(x, c, re0, poly_stride, im_off, tau);
        return;
    }
    bool loaded = true;   // last-pass operands are in registers
#pragma unroll
    for (int s = PL::NP - 1; s >= 0; s--) {
        // exchange between pass s and pass s-1 is the one exchange_sync(s) describes
        const bool first = s == 0;
        const bool local_next = !first && pass_sync_is_wave_local(PL::log_S(s));
        if (!loaded) {
#pragma unroll
            for (int p = 0; p < NPOLY; p++) pass_load<PL>(x[p], s, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
        }
#pragma unroll
        for (int p = 0; p < NPOLY; p++) {
            pass_compute<PL, true>(x[p], s, c);
            if (!first) {
                pass_store<PL>(x[p], s, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
                if (local_next) {
                    exchange_sync<PL>(s);
                    pass_load<PL>(x[p], s - 1, re0 + p * poly_stride, re0 + p * poly_stride + im_off, tau);
                }
            }
        }
        if (!first && !local_next) exchange_sync<PL>(s);
        loaded = local_next;
    }
}

// ---- integer <-> f64 conversions -------------------------------------------------------------

// torus f64 -> u64: reference commons/math/torus/mod.rs:72-78 (from_torus) =
// round((x - round(x)) * 2^64) as i64 as u64.
#ifndef FHESTR_FROM_TORUS_ROUND64
// Here: the fractional part as fixed point, straight out of the mantissa.  v_fract_f64 is exact
// (x - floor(x) in [0, 1)); adding 1.0 aligns it to 2^-52, so the 52 mantissa bits of the sum ARE
// round_to_nearest_even(frac * 2^52) and the torus word is that mantissa << 12 (a sum that rounds
// up to 2.0 has mantissa 0 = the torus' wrap-around).  [0, 1) instead of the reference's
// [-1/2, 1/2) is the same element mod 2^64.  Against the reference's rounding on the 2^-64 grid
// this adds an unbiased error of at most 2^-53 of the torus, 2^-27 of the f64 FFT's own error
// (the reference tolerates 2^14 ulp, fft/tests.rs:40-46, and measures ~2^38); 2 f64 + 2 integer
// instructions instead of 9 f64.
__device__ __forceinline__ uint64_t from_torus(double x) {
    const double u = __builtin_amdgcn_fract(x) + 1.0;
    const uint32_t lo = (uint32_t)__double2loint(u), hi = (uint32_t)__double2hiint(u);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 20) << 32) | (uint64_t)(lo << 12);
}
#else
// The reference's formula literally; rint() (ties-to-even) stands in for Rust's round() (ties away):
// they differ only for exact .5 inputs, by one ulp of the 2^-64 grid.
__device__ __forceinline__ uint64_t from_torus(double x) {
    double fr = x - rint(x);
    double y = rint(fr * 18446744073709551616.0);
    double h = floor(y * 2.3283064365386963e-10);         // y / 2^32
    double l = fma(h, -4294967296.0, y);                  // exact, in [0, 2^32)
    uint32_t hi = (uint32_t)(int32_t)h;                   // v_cvt_i32_f64 (saturating)
    uint32_t lo = (uint32_t)l;                            // v_cvt_u32_f64
    return ((uint64_t)hi << 32) | lo;
}
#endif

// signed i64 -> f64 (exact for |v| < 2^53, otherwise correctly rounded via two-part sum)
__device__ __forceinline__ double i64_to_f64(uint64_t v) {
    int32_t hi = (int32_t)(v >> 32);
    uint32_t lo = (uint32_t)v;
    return fma((double)hi, 4294967296.0, (double)lo);
}

}  // namespace fhe
