// pbs_kernels.hip.h -- hand-written gfx950 kernels for the programmable bootstrap.
//
//   bsk_convert_kernel   standard-domain bootstrapping key -> Fourier domain, in the FFT's own
//                        scrambled order.  Replaces par_convert_standard_lwe_bootstrap_key_to_fourier
//                        (core_crypto/algorithms/lwe_bootstrap_key_conversion.rs:99-152,
//                         fft64/math/fft/mod.rs:719-764).
//   blind_rotate_kernel  modulus switch + LUT rotation + n CMUX steps (rotate/subtract, signed
//                        decomposition, forward FFTs, Fourier-domain multiply-accumulate against
//                        the GGSW, inverse FFTs, torus rounding) + sample extraction, one LWE per
//                        workgroup.  Replaces FourierLweBootstrapKeyView::{blind_rotate_assign,
//                        bootstrap} (fft64/crypto/bootstrap.rs:242-364), add_external_product_assign
//                        (fft64/crypto/ggsw.rs:477-598), fast_pbs_modulus_switch
//                        (fft_impl/common.rs:26-43), the monomial rotations
//                        (algorithms/polynomial_algorithms.rs:315-354,425-490) and
//                        extract_lwe_sample_from_glwe_ciphertext (glwe_sample_extraction.rs:91-147).
//
// Data layout in HBM:
//   small LWE batch  [B][n+1] u64 (mask then body)            entities/lwe_ciphertext.rs:598-625
//   LUTs             [n_luts][(k+1)][N] u64                   entities/glwe_ciphertext.rs:210-222
//   Fourier BSK      [n][level idx][row][col][rho][tau] c64   (rho = register slot, tau = thread:
//                    one 16-byte element per lane per load -> fully coalesced dwordx4 streams)
//   big LWE batch    [B][kN+1] u64
#pragma once
#include <type_traits>
#include "negacyclic_fft.hip.h"

namespace fhe {

// Diagnostic build only (-DFHESTR_STAMPS, scripts/stamp_profile.py): s_memtime stamps around the
// segments of one CMUX step of blind_rotate_kernel, summed per wave in scalar registers and stored once
// after the loop into a buffer nothing else reads (cdna_hip_programming.md section 7, "In-kernel stamps").
// The fences forbid overlaps the real kernel has: read the SHARES, never this build's run time.
#ifdef FHESTR_WALL
// diagnostic build (-DFHESTR_WALL, scripts/wall_spread.py): wall time of every workgroup on the constant 100 MHz clock
// and the XCD it ran on -- do all workgroups of a launch take the same TIME, not just the same cycles?
__device__ unsigned long long g_wall[4096 * 2];
#endif
#ifdef FHESTR_STAMPS
constexpr int STAMP_SEGS = 10;
__device__ unsigned long long g_stamps[4096 * 8 * STAMP_SEGS];
#define FHE_STAMP_DECL unsigned long long stamp_acc[STAMP_SEGS] = {}; unsigned long long stamp_prev = 0
#define FHE_STAMP(idx)                                                                                   \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        unsigned long long _t;                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((idx) >= 0) stamp_acc[(idx) < 0 ? 0 : (idx)] += _t - stamp_prev;                             \
        stamp_prev = _t;                                                                                 \
    } while (0)
#else
#define FHE_STAMP_DECL do {} while (0)
#define FHE_STAMP(idx) do {} while (0)
#endif

// Two workgroups of the two-LWEs-per-CU kernel share every SIMD, and the hardware arbitrates vector issue by age: the
// workgroup that arrived first runs at almost its solo speed (3.2 ms), the other one gets the leftovers and then finishes
// alone (5.0 ms; profiles/r03_wide_wall.txt: a 512-LWE launch takes as long as its slowest workgroup).  With
// BlindRotateArgs::fair_shift = n > 0 the two take turns at s_setprio 1 in time slices of 2^n ticks of the 100 MHz clock,
// complementary by the parity of a ticket per hardware CU (monotonic across launches), so both progress at the same rate
// and finish together: 512 LWEs 4.64 -> 4.20 ms, 1024 LWEs 8.79 -> 8.24 ms.  The host turns it on when every CU gets an
// even number of workgroups; with an odd number the age order is the better pipeline (768 LWEs: 6.9 ms unfair, 7.3 fair).
__device__ uint32_t g_cu_tickets[2048];

struct BlindRotateArgs {
    const uint64_t* lwe_small;   // [B][n+1]
    const uint32_t* lut_idx;     // [B] or nullptr
    const uint64_t* luts;        // [n_luts][(k+1)N]
    const double* fbsk;          // Fourier BSK (see layout above)
    uint64_t* lwe_out;           // [B][kN+1]
    uint32_t n;                  // small LWE dimension
    uint32_t base_log;           // PBS decomposition base log
    uint32_t batch;
    uint32_t grouping;           // EXTPROD kernels only: mask elements per step (multi-bit grouping factor)
    uint32_t fair_shift;         // two-LWEs-per-CU kernel only: log2 ticks of the priority time slice (0 = hardware age order)
};

// ((x >> (63 - bL)) + 1) >> 1 masked to bL bits == closest_representable(x) >> (64 - bL)
// (commons/math/decomposition/decomposer.rs:98-118 + fft64/math/decomposition.rs:33-35); bL <= 31.
__device__ __forceinline__ uint32_t decomp_init_state(uint64_t x, uint32_t bL) {
    uint32_t t = (uint32_t)(x >> (63 - bL));
    return ((t + 1u) >> 1) & ((1u << bL) - 1u);
}
// commons/math/decomposition/iter.rs:120-127 on a 32-bit state; returns the signed digit.
__device__ __forceinline__ int32_t decomp_next_digit(uint32_t& state, uint32_t base_log) {
    const uint32_t mask = (1u << base_log) - 1u;
    uint32_t res = state & mask;
    state >>= base_log;
    uint32_t carry = ((res - 1u) | state) & res;
    carry >>= base_log - 1;
    state += carry;
    return (int32_t)(res - (carry << base_log));
}
// 64-bit state variants for base_log * level > 31 (e.g. base 2^11 x 3 levels)
__device__ __forceinline__ uint64_t decomp_init_state64(uint64_t x, uint32_t bL) {
    const uint64_t t = x >> (63 - bL);
    return ((t + 1ull) >> 1) & ((1ull << bL) - 1ull);
}
__device__ __forceinline__ int32_t decomp_next_digit64(uint64_t& state, uint32_t base_log) {
    const uint64_t mask = (1ull << base_log) - 1ull;
    uint64_t res = state & mask;
    state >>= base_log;
    uint64_t carry = ((res - 1ull) | state) & res;
    carry >>= base_log - 1;
    state += carry;
    return (int32_t)(uint32_t)(res - (carry << base_log));
}
// Single-level fast path (L == 1): the state IS the only digit's residue, so
// digit = res - (res > B/2 ? B : 0)  -- the same value iter.rs:120-127 yields when state >> b == 0
// (carry = bit b-1 of ((res-1) & res) = [res > B/2]).
__device__ __forceinline__ int32_t decomp_single_digit(uint64_t x, uint32_t b) {
    const uint32_t t = (uint32_t)(x >> 32) >> (31 - b);          // b+1 top bits (b <= 31)
    const uint32_t res = __builtin_amdgcn_ubfe(t + 1u, 1u, b);     // ((t + 1) >> 1) mod B
    const uint32_t half = 1u << (b - 1);
    return (int32_t)(res - ((res + (half - 1u)) & (half << 1)));   // minus B iff res > B/2
}
// The same digit, biased: returns digit + (B/2 - 1) in [0, B) in two instructions.  With res = ((t+1)>>1) mod B
// as above, digit + B/2 - 1 = (res + B/2 - 1) mod B (no wrap for res <= B/2, one wrap = the "minus B" for
// res > B/2), and both the +1 of the rounding and the + B/2 - 1 are additions below the extracted field:
// bits [32-b, 32) of hi + 2^(31-b) + (B/2 - 1) 2^(32-b) = hi + 2^31 - 2^(31-b)  (mod 2^32 = mod B up there).
// The constant bias is taken out again for free: the caller converts the unsigned value to f64 and folds
// -(B/2 - 1) (1 + i) * twist into the addend of the twist multiplication (digit_point).
__device__ __forceinline__ uint32_t decomp_bias_constant(uint32_t b) {
    uint32_t c = 0x80000000u - (1u << (31 - b));
    asm volatile("" : "+s"(c));      // opaque: keeps hipcc from splitting the add into xor 2^31 + sub
    return c;
}
__device__ __forceinline__ uint32_t decomp_single_biased(uint64_t x, uint32_t b, uint32_t bias_constant) {
    const uint32_t hi = (uint32_t)(x >> 32);
    return __builtin_amdgcn_ubfe(hi + bias_constant, 32u - b, b);
}
// (lo + i hi - c (1 + i)) * twist with the constant part kb = -c (1 + i) twist precomputed: 2 cvt + 4 fma
__device__ __forceinline__ cplx digit_point(uint32_t lo_biased, uint32_t hi_biased, cplx tw, cplx kb) {
    const double a = (double)lo_biased, b = (double)hi_biased;
    cplx r;
    r.re = fma(a, tw.re, fma(-b, tw.im, kb.re));
    r.im = fma(a, tw.im, fma(b, tw.re, kb.im));
    return r;
}
// fft_impl/common.rs:26-43 (offset 0, lut_count_log 0): result in [0, 2N]
__device__ __forceinline__ uint32_t modulus_switch(uint64_t x, int logN) {
    uint64_t o = x >> (64 - logN - 2);
    return (uint32_t)((o + 1) >> 1);
}

// The Fourier key as a buffer resource (stride 0, raw): buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen
// takes its row offset from a scalar register and bounds-checks against the key's size.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ auto key_resource(const double* fbsk, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(fbsk), 0, bytes < 0x7FFFFFF0u ? (int)bytes : 0x7FFFFFF0, 0x00020000);
}
template <class RSRC>
__device__ __forceinline__ double2 key_load(RSRC rsrc, uint32_t voff, uint32_t soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}

// LDS slot of coefficient j (< N) inside one polynomial's accumulator copy
template <class PL>
__device__ __forceinline__ uint32_t acc_slot_of(uint32_t j) { return (uint32_t)PL::acc_slot((int)j); }

// Monomial rotation bookkeeping of one CMUX step: for the coefficient j = PL::point(tau, m) + h*P this
// thread owns, where (acc * X^d)[j] = +-acc[(j - rem) mod N] sits in the LDS copy and whether it is
// negated (polynomial_algorithms.rs:463-489; d = odd * N + rem).  (Written for FftSwap10's four waves; FftSwap9's two
// waves are the same with 2 in place of 4 -- swap_wave_bits.)  The swap plan's thread owns
// j = 4 q' + w (w = wave index) and stores coefficient 4 q + r at slot r * N/4 + q, so j - rem =
// 4 (q' - rem/4 - [w < rem%4]) + ((w - rem) mod 4): everything but one add and one and-or per
// coefficient is per-step or compile-time.
template <class PL>
__host__ __device__ constexpr int swap_wave_bits() {
    if constexpr (PL::SWAP) return PL::LOGW; else return 0;
}
template <class PL, int LOGN>
struct Rotation {
    int32_t oddmask;
    uint32_t rem;        // generic plans
    int32_t qb8;         // swap plan: 8 * (lane - rem/4 - borrow), plus 2^31 when the rotation count is odd
    uint32_t rbits8;     // swap plan: byte offset of row ((w - rem) mod 4) of the transposed accumulator copy
    __device__ __forceinline__ Rotation(uint32_t d, int tau) {
        rem = d & ((1u << LOGN) - 1);
        oddmask = -(int32_t)((d >> LOGN) & 1);
        if constexpr (PL::SWAP) {
            constexpr int LW = swap_wave_bits<PL>();      // FftSwap10: j = 4 q + w; FftSwap9: j = 2 q + w
            const int w = __builtin_amdgcn_readfirstlane(tau >> 6), lane = tau & 63;   // wave index: scalar
            const int rr = (int)(rem & ((1u << LW) - 1));
            // the sign of q = lane - rem/4 - borrow + const says "wrapped around"; adding 2^31 flips it when
            // the whole polynomial is negated as well, so one arithmetic shift yields the negation mask
            qb8 = (int32_t)((uint32_t)((lane - (int)(rem >> LW) - (w < rr ? 1 : 0)) << 3) + ((uint32_t)oddmask & 0x80000000u));
            rbits8 = (uint32_t)((w - rr) & ((1 << LW) - 1)) << (LOGN - LW + 3);
        } else {
            qb8 = 0; rbits8 = 0;
        }
    }
    // slot of the source coefficient and its all-ones / all-zero negation mask
    __device__ __forceinline__ void source(int tau, int m, int h, uint32_t& slot, uint32_t& neg) const {
        static_assert(!PL::SWAP, "swap plan: use source_bytes");
        const uint32_t j = (uint32_t)PL::point(tau, m) + (uint32_t)h * PL::P;
        neg = (uint32_t)(((int32_t)(j - rem) >> 31) ^ oddmask);   // j, rem < 2^31
        slot = acc_slot_of<PL>((j - rem) & ((1u << LOGN) - 1));
    }
    // swap plan: byte offset inside one polynomial's accumulator copy (to be OR-ed onto its 8N-aligned base:
    // three instructions per coefficient -- add, and-or, arithmetic shift)
    // `row_base` = the copy's base address | rbits8 (scalar, formed once per step by the caller)
    __device__ __forceinline__ void source_bytes(int m, int h, uint32_t row_base, uint32_t& address, uint32_t& neg) const {
        constexpr int LW = swap_wave_bits<PL>();
        const int32_t q8 = qb8 + ((PL::point(0, m) + h * PL::P) >> LW) * 8;
        neg = (uint32_t)(q8 >> 31);
        address = ((uint32_t)q8 & (((1u << (LOGN - LW)) - 1) << 3)) | row_base;       // v_and_or_b32
    }
};

// LDS access by byte address (32-bit, address space 3)
typedef __attribute__((address_space(3))) const uint64_t lds_cu64_t;
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint64_t lds_load_u64(uint32_t byte_address) { return *(lds_cu64_t*)(uintptr_t)byte_address; }

template <int LOGN, int LOGR, int K1, int L>
struct BrCfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int P = N / 2;
    using PL = typename PlanFor<LOGN - 1, LOGR>::type;
    static constexpr int R = PL::R;
    static constexpr int T = PL::T;
    static constexpr int THREADS = K1 * T;
    // LDS: accumulator copy (rotation gather source) + FFT exchange planes + spectrum broadcast
    // the imaginary plane sits PLANE = P + 2 slots after the real one: an offset that is neither
    // <= 255 slots nor a multiple of 64 slots, so hipcc cannot fuse a re/im pair into one
    // ds_read2st64_b64 / ds_write2st64_b64 (half the LDS bandwidth of two plain 8-byte accesses)
    static constexpr int PLANE = P + 2;
    static constexpr int GROUP_SLOTS = 2 * P + 4;
    static constexpr size_t LDS_FIXED = (size_t)K1 * N * 8 /*acc*/ + (size_t)K1 * GROUP_SLOTS * 8 /*x*/ +
                                        (size_t)K1 * GROUP_SLOTS * 8 /*F*/;
    static size_t lds_bytes(uint32_t n) { return LDS_FIXED + (size_t)n * 4; }   // + modswitched mask
};

// ------------------------------------------------------------------------------------------------
template <int LOGN, int LOGR, int K1, int L>
__global__ void __launch_bounds__((BrCfg<LOGN, LOGR, K1, L>::THREADS))
bsk_convert_kernel(const uint64_t* __restrict__ bsk_std, double* __restrict__ fbsk, uint32_t n_polys) {
    using CFG = BrCfg<LOGN, LOGR, K1, L>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    const int g = threadIdx.x / T, tau = threadIdx.x % T;
    double* xre = reinterpret_cast<double*>(smem) + (size_t)g * CFG::GROUP_SLOTS;
    double* xim = xre + CFG::PLANE;
    const uint32_t poly = blockIdx.x * K1 + g;   // K1 polynomials per workgroup
    const bool active = poly < n_polys;
    FftConsts<PL> fc;
    fft_init_consts<PL>(fc, tau);
    cplx x[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int j = PL::point(tau, m);
        uint64_t a = active ? bsk_std[(size_t)poly * N + j] : 0;
        uint64_t b = active ? bsk_std[(size_t)poly * N + j + P] : 0;
        // forward_as_torus: signed value * 2^-64 (fft/mod.rs:197-218)
        cplx z;
        // the inverse transform's 1/(N/2) (fft/mod.rs:285-304) is folded in here: linear, exact (power of two)
        z.re = i64_to_f64(a) * (5.421010862427522e-20 / P);
        z.im = i64_to_f64(b) * (5.421010862427522e-20 / P);
        double sn, cs;
        sincospi((double)j / (double)N, &sn, &cs);  // twisty e^{i pi j / N}
        cplx w; w.re = cs; w.im = sn;
        x[m] = cmul(z, w);
    }
    fft_forward<PL>(x, fc, xre, xim, tau);
    if (active) {
        double2* out = reinterpret_cast<double2*>(fbsk) + (size_t)poly * P;
#pragma unroll
        for (int rho = 0; rho < R; rho++) out[rho * T + tau] = make_double2(x[rho].re, x[rho].im);
    }
}

// ------------------------------------------------------------------------------------------------
// One workgroup = one LWE sample; K1 groups of T threads, group g owns GLWE polynomial g of the
// accumulator (2R coefficients per thread, in VGPRs for the whole kernel).
//
// LDS regions (disjoint, so that no barrier is needed just to recycle memory):
//   lds_acc [K1][N] u64      copy of the accumulator, gather source of the monomial rotation
//   lds_x   [K1][2][P] f64   FFT exchange planes (swizzled slots)
//   lds_f   [K1][2][P] f64   spectrum broadcast between the polynomial groups
//   lds_d   [n] u32          modulus-switched mask elements (0xFFFFFFFF marks a_i == 0)
// Barriers per CMUX step: forward exchange 0->1, spectrum publish, inverse exchange 1->0, accumulator
// publish; every other exchange is wave-local.
// EXTPROD (multi-bit PBS through the two-kernel path, pbs_multibit_kernels.hip.h): n / grouping steps, each
// a plain external product acc <- GGSW (x) acc on a zeroed destination (no rotate-and-subtract, no
// accumulate) against this LWE's own prepared GGSWs, args.fbsk = [batch][n / grouping] GGSWs in key layout.
template <int LOGN, int LOGR, int K1, int L, bool EXTPROD = false>
__global__ void __launch_bounds__((BrCfg<LOGN, LOGR, K1, L>::THREADS))
blind_rotate_kernel(BlindRotateArgs args) {
    using CFG = BrCfg<LOGN, LOGR, K1, L>;
    static_assert(!EXTPROD || !CFG::PL::SWAP, "N = 2048 has its own multi-bit kernels");
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* lds_acc = reinterpret_cast<uint64_t*>(smem);                       // [K1][N]
    double* lds_x = reinterpret_cast<double*>(smem + (size_t)K1 * N * 8);        // [K1][GROUP_SLOTS]
    double* lds_f = lds_x + (size_t)K1 * CFG::GROUP_SLOTS;                       // [K1][GROUP_SLOTS]
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(lds_f + (size_t)K1 * CFG::GROUP_SLOTS);   // [n]

    int g = threadIdx.x / T;
    const int tau = threadIdx.x % T;
    // a polynomial group is a whole number of wavefronts here: tell the compiler that g is wave-uniform,
    // so the group-dependent address arithmetic (key rows, LDS planes) runs on the scalar unit
    if constexpr (T % 64 == 0) g = __builtin_amdgcn_readfirstlane(g);
#ifdef FHESTR_WALL
    const unsigned long long wall_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // keep-busy launches (Engine::keep_busy) carry replicas: workgroups beyond the batch recompute one of its LWEs and
    // store nothing -- they only keep the idle CUs drawing power, so that the clock has not ramped down when the next
    // large launch arrives (profiles/r03_after_idle.txt)
    const uint32_t sample = blockIdx.x < args.batch ? blockIdx.x : blockIdx.x % args.batch;
    const uint32_t n = args.n;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    uint64_t* my_acc = lds_acc + (size_t)g * N;
    // byte address of this group's accumulator copy: 8N-aligned (the dynamic LDS segment starts at 0 and the
    // copies come first), so the gather can OR its offsets onto it
    const uint32_t my_acc_address = lds_address(my_acc);
    if (PL::SWAP && (my_acc_address & (8u * N - 1u))) __builtin_trap();
    double* xre = lds_x + (size_t)g * CFG::GROUP_SLOTS;
    double* xim = xre + CFG::PLANE;
    const uint32_t bL = args.base_log * L;
    const uint32_t dbias = decomp_bias_constant(bL <= 31 ? bL : 31);

    // modulus switch of the whole mask once (fft_impl/common.rs:26-43); a_i == 0 is skipped (:281)
    const uint32_t steps = EXTPROD ? n / args.grouping : n;
    for (uint32_t i = threadIdx.x; i < steps; i += CFG::THREADS) {
        const uint64_t a = lwe[i];
        lds_d[i] = EXTPROD ? 0u : (a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN));
    }

    // per-thread constants: inter-pass twiddles and twisties (1/P is folded into the Fourier key)
    FftConsts<PL> fc;
    fft_init_consts<PL>(fc, tau);
    cplx twist[R], twbias[R];        // twbias: -(B/2 - 1)(1 + i) * twist, the single-level digit's bias (decomp_single_biased)
#pragma unroll
    for (int m = 0; m < R; m++) {
        double sn, cs;
        sincospi((double)PL::point(tau, m) / (double)N, &sn, &cs);
        twist[m].re = cs; twist[m].im = sn;
        const double cb = -(double)((1u << (args.base_log * L - 1)) - 1u);
        twbias[m].re = cb * (cs - sn);
        twbias[m].im = cb * (cs + sn);
    }

    // acc <- LUT * X^{-ms(body)}   (bootstrap.rs:254-271, polynomial_algorithms.rs:331-353)
    uint64_t acc_lo[R], acc_hi[R];   // coefficients j = PL::point(tau, m) and j + P
    {
        const uint32_t d = modulus_switch(lwe[n], LOGN);
        const uint32_t rem = d & (N - 1);
        const bool odd = (d >> LOGN) & 1;
#pragma unroll
        for (int m = 0; m < R; m++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t j = PL::point(tau, m) + h * P;
                const uint32_t src = (j + rem) & (N - 1);       // out[j] = +-in[j + rem]
                const bool neg = ((j + rem) >= (uint32_t)N) != odd;
                uint64_t v = lut[(size_t)g * N + src];
                v = neg ? (0 - v) : v;
                if (h == 0) acc_lo[m] = v; else acc_hi[m] = v;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        my_acc[acc_slot_of<PL>(PL::point(tau, m))] = acc_lo[m];
        my_acc[acc_slot_of<PL>(PL::point(tau, m) + P)] = acc_hi[m];
    }
    __syncthreads();

    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)L * K1 * K1 * P;   // complex elements per GGSW

    uint32_t d_next = lds_d[0];
    const uint32_t key_off = (uint32_t)tau * 16u;     // this thread's byte offset inside a Fourier polynomial row
    // (EXTPROD: the resource covers this LWE's GGSWs only -- offsets stay below 2^32)
    if constexpr (EXTPROD) fbsk += (size_t)sample * steps * GGSW_ELEMS;
    const auto key_rsrc = key_resource(reinterpret_cast<const double*>(fbsk), (size_t)steps * GGSW_ELEMS * 16);
    FHE_STAMP_DECL;
    FHE_STAMP(-1);
    for (uint32_t i = 0; i < steps; i++) {
        // the modulus-switched mask element is the same for the whole workgroup: as a scalar, the rotation's
        // uniform parts (quotient, remainder, sign) cost no vector instructions
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_next);
        d_next = lds_d[i + 1 < steps ? i + 1 : i];              // prefetch (LDS broadcast read)
        if (d == 0xFFFFFFFFu) continue;                          // block-uniform
        const Rotation<PL, LOGN> rot(d, tau);

        // Fourier GGSW rows of the last decomposition level handled first (ggsw.rs:524): issue the
        // loads now, they land while the forward FFT runs (L2 / Infinity-Cache resident key).  Buffer
        // loads: the row's byte offset is a scalar (soffset), the thread's offset one constant VGPR -- no
        // vector instruction is spent on addresses.
        const double2* bk0 = fbsk + (size_t)i * GGSW_ELEMS;
        double2 bpre[K1][R];
        {
#pragma unroll
            for (int r = 0; r < K1; r++) {
                const int row = (g + r) % K1;       // r = 0 is this group's own row: branch-free below
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const uint32_t soff = (uint32_t)((i * GGSW_ELEMS + (size_t)(L - 1) * K1 * K1 * P +
                                                      ((size_t)row * K1 + g) * P + rho * T) * 16);
                    bpre[r][rho] = key_load(key_rsrc, key_off, soff);
                }
            }
        }

        // ct1 = acc * X^d - acc  (polynomial_algorithms.rs:463-489), then decomposition state
        uint32_t st_lo[R], st_hi[R];
        uint32_t row_base = my_acc_address | rot.rbits8;         // scalar; opaque so that it stays ONE operand
        if constexpr (PL::SWAP) asm volatile("" : "+s"(row_base));   // of the per-coefficient v_and_or_b32
#pragma unroll
        for (int m = 0; m < R; m++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if constexpr (EXTPROD) {                         // the accumulator itself is decomposed
                    const uint64_t own = h == 0 ? acc_lo[m] : acc_hi[m];
                    const uint32_t st = L == 1 ? decomp_single_biased(own, bL, dbias) : decomp_init_state(own, bL);
                    if (h == 0) st_lo[m] = st; else st_hi[m] = st;
                    continue;
                }
                uint32_t sm32;                                   // (acc*X^d)[j] = +-acc[j - rem], select-free
                uint64_t gathered;
                if constexpr (PL::SWAP) {
                    uint32_t address;
                    rot.source_bytes(m, h, row_base, address, sm32);
                    gathered = lds_load_u64(address);
                } else {
                    uint32_t slot;
                    rot.source(tau, m, h, slot, sm32);
                    gathered = my_acc[slot];
                }
                const uint64_t sm = ((uint64_t)sm32 << 32) | sm32;
                const uint64_t v = (gathered ^ sm) - sm;
                const uint64_t own = h == 0 ? acc_lo[m] : acc_hi[m];
                const uint32_t st = L == 1 ? decomp_single_biased(v - own, bL, dbias) : decomp_init_state(v - own, bL);
                if (h == 0) st_lo[m] = st; else st_hi[m] = st;
            }
        }

        FHE_STAMP(0);    // key loads issued, rotation gather + decomposition (incl. the gather's LDS latency)
        cplx outf[R];
        if constexpr (PL::SWAP) {
            // Swap plan (single level): the forward transform's only workgroup barrier is also the
            // spectrum hand-over -- after it every group runs the last radix-4 pass on its own AND
            // on the other groups' points, so no spectrum is ever written back to LDS.
            static_assert(!PL::SWAP || L == 1, "swap plan: single decomposition level only");
            cplx x[K1][R];
#pragma unroll
            for (int m = 0; m < R; m++) x[0][m] = digit_point(st_lo[m], st_hi[m], twist[m], twbias[m]);   // fft/mod.rs:220-239
#ifdef FHESTR_WALL
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wall[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime() - wall_t0;
        g_wall[blockIdx.x * 2 + 1] = xcc & 7u;
    }
#endif
#ifdef FHESTR_STAMPS
            swap10_fwd_stage1(x[0], fc, xre, xim, tau);
            FHE_STAMP(1);    // convert + twist + stage 1 (2 passes, 1 swap transpose, 8 LDS writes)
            wave_local_fence();
            swap10_fwd_stage2(x[0], fc, xre, xim, tau);
            FHE_STAMP(2);    // stage 2 (8 LDS reads, 2 passes, 1 swap transpose)
            wave_local_fence();
            swap10_fwd_stage3(x[0], fc, xre, xim, tau);
            FHE_STAMP(3);    // stage 3 (twiddle + 8 LDS writes)
#else
            swap10_forward_head(x[0], fc, xre, xim, tau);
#endif
            __syncthreads();
            FHE_STAMP(4);    // barrier 1 (forward hand-over)
#pragma unroll
            for (int r = 0; r < K1; r++) {
                const int row = (g + r) % K1;                    // r = 0: own polynomial
                const double* rre = lds_x + (size_t)row * CFG::GROUP_SLOTS;
                swap10_forward_tail(x[r], rre, rre + CFG::PLANE, tau);
            }
            // outf[col = g] = sum_row FBSK[i][0][row][g] * F[row]   (ggsw.rs:616-697)
#pragma unroll
            for (int r = 0; r < K1; r++) {
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 bv = bpre[r][rho];
                    const cplx f = x[r][rho];
                    if (r == 0) {
                        outf[rho].re = bv.x * f.re - bv.y * f.im;
                        outf[rho].im = bv.x * f.im + bv.y * f.re;
                    } else {
                        outf[rho].re = fma(bv.x, f.re, fma(-bv.y, f.im, outf[rho].re));
                        outf[rho].im = fma(bv.x, f.im, fma(bv.y, f.re, outf[rho].im));
                    }
                }
            }
            // inverse on its own planes (lds_f): other groups may still be reading this group's
            // forward planes
            double* fre = lds_f + (size_t)g * CFG::GROUP_SLOTS;
            double* fim = fre + CFG::PLANE;
            FHE_STAMP(5);    // 16 LDS reads + last pass on both polynomials + key wait + multiply-accumulate
            swap10_inverse_head(outf, fre, fim, tau);
            FHE_STAMP(6);    // inverse first pass + 8 LDS writes
            __syncthreads();
            FHE_STAMP(7);    // barrier 2 (inverse hand-over)
            swap10_inverse_tail(outf, fc, fre, fim, tau);
            FHE_STAMP(8);    // inverse passes 2-5 (two LDS round trips, two swap transposes)
        } else {
#pragma unroll
        for (int it = 0; it < L; it++) {
            const int lvl_idx = L - 1 - it;                      // ggsw.rs:524 (levels reversed)
            cplx x[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                if (L == 1) { x[m] = digit_point(st_lo[m], st_hi[m], twist[m], twbias[m]); continue; }
                cplx z;
                z.re = (double)decomp_next_digit(st_lo[m], args.base_log);
                z.im = (double)decomp_next_digit(st_hi[m], args.base_log);
                x[m] = cmul(z, twist[m]);                        // fft/mod.rs:220-239
            }
            fft_forward<PL>(x, fc, xre, xim, tau);
            // publish this row's spectrum for the other polynomial groups
            if (it > 0) __syncthreads();                         // previous level's readers done
            {
                double* fre = lds_f + (size_t)g * CFG::GROUP_SLOTS;
                double* fim = fre + CFG::PLANE;
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    fre[rho * T + tau] = x[rho].re;
                    fim[rho * T + tau] = x[rho].im;
                }
            }
            __syncthreads();
            // outf[col = g] (+)= sum_row FBSK[i][lvl][row][g] * F[row]   (ggsw.rs:616-697)
            const double2* bk = bk0 + (size_t)lvl_idx * K1 * K1 * P;
#pragma unroll
            for (int r = 0; r < K1; r++) {
                const int row = (g + r) % K1;                    // own row first (spectrum still in registers)
                const double* fre = lds_f + (size_t)row * CFG::GROUP_SLOTS;
                const double* fim = fre + CFG::PLANE;
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 bv = it == 0 ? bpre[r][rho] : bk[((size_t)row * K1 + g) * P + rho * T + tau];
                    cplx f;
                    if (r == 0) {
                        f = x[rho];
                    } else {
                        f.re = fre[rho * T + tau];
                        f.im = fim[rho * T + tau];
                    }
                    if (it == 0 && r == 0) {
                        outf[rho].re = bv.x * f.re - bv.y * f.im;
                        outf[rho].im = bv.x * f.im + bv.y * f.re;
                    } else {
                        outf[rho].re = fma(bv.x, f.re, fma(-bv.y, f.im, outf[rho].re));
                        outf[rho].im = fma(bv.x, f.im, fma(bv.y, f.re, outf[rho].im));
                    }
                }
            }
        }

        // back to the standard domain and accumulate (fft/mod.rs:285-304, 539-557); the 1/(N/2)
        // normalisation lives in the Fourier key (bsk_convert_kernel)
        fft_inverse<PL>(outf, fc, xre, xim, tau);
        }
        // every gather of this step's my_acc happened several barriers ago: safe to overwrite
#pragma unroll
        for (int m = 0; m < R; m++) {
            cplx t = cmul_conj(outf[m], twist[m]);
            if constexpr (EXTPROD) {
                acc_lo[m] = from_torus(t.re);
                acc_hi[m] = from_torus(t.im);
                continue;
            }
            acc_lo[m] += from_torus(t.re);
            acc_hi[m] += from_torus(t.im);
            my_acc[acc_slot_of<PL>(PL::point(tau, m))] = acc_lo[m];
            my_acc[acc_slot_of<PL>(PL::point(tau, m) + P)] = acc_hi[m];
            FHE_PIN_ORDER();      // next point's conversions overlap this point's LDS writes
        }
        __syncthreads();
        FHE_STAMP(9);    // untwist + torus rounding + accumulate + 8 LDS writes + barrier 3
    }
#ifdef FHESTR_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)
        for (int sg = 0; sg < STAMP_SEGS; sg++)
            g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * STAMP_SEGS + sg] = stamp_acc[sg];
#endif

    if (blockIdx.x >= args.batch) return;          // a replica: nothing to store
    // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146)
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
    for (int m = 0; m < R; m++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t j = PL::point(tau, m) + h * P;
            const uint64_t v = h == 0 ? acc_lo[m] : acc_hi[m];
            if (g == K1 - 1) {
                if (j == 0) out[(size_t)(K1 - 1) * N] = v;         // body = B[0]
            } else {
                if (j == 0) out[(size_t)g * N] = v;
                else out[(size_t)g * N + (N - j)] = 0 - v;         // out[t] = -A[N - t]
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Variant "wide": one workgroup = one LWE, T = N/2/R threads, every thread carries the same R
// spectrum slots of ALL k+1 polynomials.  The Fourier-domain multiply-accumulate against the GGSW
// is then thread-local (no spectrum broadcast through LDS, no barrier for it), each thread has
// k+1 independent FFT streams to overlap LDS round trips with butterflies, and the smaller
// footprint (accumulator copy + exchange planes) lets two LWEs share a CU for batches >= 512.
// Barriers per CMUX step: forward exchange 0->1, inverse exchange 1->0, accumulator publish.
// FftSwap11 for N = 4096 (negacyclic_fft.hip.h; -DFHESTR_SWAP11=0: the generic plan).  60 % fewer LDS stores; its twiddles of
// passes 1-2 come from an LDS table (FftSwapLdsConsts) and the digit bias is subtracted as an integer, because with everything in
// VGPRs next to the prefetched key rows the kernel spilled 51 registers whose reloads queue behind the key loads (11.9 ms per
// 256 LWEs against the generic plan's 7.55; now 6.8: profiles/r04_n4096.txt).
#ifndef FHESTR_SWAP11
#define FHESTR_SWAP11 1
#endif
// FftSwap9 under the N = 1024 two-per-CU kernel as well (three polynomials stage by stage, the dense kernel's key copy): correct
// (tests, 87 k-PBS soak); with its key loads batched per GGSW row (ROW_BATCH: left alone the compiler serialised them, 10.0 ms per
// 512 LWEs) it is level with the generic plan, which then took the whole-key prefetch below and is the faster of the two in
// the overlapped mode; off (profiles/r04_n1024.txt).
#ifndef FHESTR_WIDE_SWAP9
#define FHESTR_WIDE_SWAP9 0
#endif
template <int LOGN, int LOGR, int K1, int L, bool KEYPF = false>
struct BrWideCfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int P = N / 2;
    // N = 4096 with four points per thread runs FftSwap11 (round 4); the wide kernel is the only one for that size, so its
    // Fourier key is simply in that plan's order (bsk_convert_wide_kernel)
    // N = 1024 with four points per thread (k = 2: PARAM_MESSAGE_2_CARRY_1 ...) runs FftSwap9 and reads the key copy the dense
    // kernel reads (same plan, same order) -- the one-per-CU kernel of that size keeps the generic plan and its own copy
    static constexpr bool OWN_PLAN = (FHESTR_SWAP11 && LOGN == 12 && LOGR == 2) || (FHESTR_WIDE_SWAP9 && LOGN == 10 && LOGR == 2);
    static constexpr bool LDS_TWIDDLES = OWN_PLAN && LOGN == 12;        // FftSwapLdsConsts: where the register file is full
    using PL = typename std::conditional<!OWN_PLAN, typename PlanFor<LOGN - 1, LOGR>::type,
                                         typename std::conditional<LOGN == 12, FftSwap11, FftSwap9>::type>::type;
    static constexpr int R = PL::R;
    static constexpr int T = PL::T;
    static constexpr int THREADS = T;
    static constexpr int PLANE = P + 2;
    static constexpr int GROUP_SLOTS = 2 * P + 4;
    // plans with more than four passes keep only pass 0's twiddles in VGPRs (FftHybridConsts)
    static constexpr bool TW_IN_LDS = !PL::SWAP && PL::NTW > 4;
    static constexpr size_t LDS_TW = TW_IN_LDS ? (size_t)FftHybridConsts<PL>::ENTRIES * 16
                                     : LDS_TWIDDLES ? (size_t)FftSwapLdsConsts<PL>::ENTRIES * 16 : 0;
    static constexpr size_t LDS_FIXED = (size_t)K1 * N * 8 /*acc*/ + (size_t)K1 * GROUP_SLOTS * 8 /*x*/ + LDS_TW;
    // N >= 4096: the accumulator lives in its LDS copy only -- a thread re-reads its own 2 K1 R coefficients at the gather and
    // at the update (the dense kernel's arrangement, pbs_dense_kernels.hip.h) instead of holding them in 4 K1 R VGPRs next to
    // 256 registers' worth of transform state (47 / 67 spilled dwords with one / two levels before)
#ifndef FHESTR_WIDE_ACC_LDS_LOGN
#define FHESTR_WIDE_ACC_LDS_LOGN 12
#endif
    static constexpr bool ACC_IN_LDS = LOGN >= FHESTR_WIDE_ACC_LDS_LOGN;
    // keep the whole Fourier GGSW of a step in VGPRs only when it is small
    // (also where the twiddles moved to LDS: without the prefetch N = 4096 has no spills but runs 9.5 instead of 7.8 ms)
    // ... or, KEYPF, where the kernel runs one wave per SIMD anyway (N = 1024 with k = 2: three polynomials per thread, 270-300
    // VGPRs): 144 more registers cost no occupancy there and hide the key's L2 latency behind the forward transforms (512 LWEs
    // 4.95 -> 4.6-4.8 ms).  A second instantiation, for single launches only: two launches of it overlapped on two streams
    // (pipeline mode 2) mostly fail to co-reside (4.15 ms per 256-LWE call in 5 runs of 6, 2.27 in the sixth; the 272-register
    // build: 2.47-2.61 every time) -- profiles/r04_n1024.txt.
    static constexpr bool PREFETCH_ALL = K1 * K1 * R * 4 <= 64 || KEYPF;
    static constexpr bool ROW_BATCH = !PREFETCH_ALL && OWN_PLAN && L == 1;      // see the products in the kernel
};

template <int LOGN, int LOGR, int K1, int L, bool KEYPF = false>
__global__ void __launch_bounds__((BrWideCfg<LOGN, LOGR, K1, L, KEYPF>::THREADS))
blind_rotate_wide_kernel(BlindRotateArgs args) {
    using CFG = BrWideCfg<LOGN, LOGR, K1, L, KEYPF>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* lds_acc = reinterpret_cast<uint64_t*>(smem);                       // [K1][N]
    double* lds_x = reinterpret_cast<double*>(smem + (size_t)K1 * N * 8);        // [K1][GROUP_SLOTS]
    double2* lds_tw = reinterpret_cast<double2*>(lds_x + (size_t)K1 * CFG::GROUP_SLOTS);    // [LDS_TW / 16]
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(lds_tw + CFG::LDS_TW / 16);               // [n]

    const int tau = threadIdx.x;
#ifdef FHESTR_WALL
    const unsigned long long wall_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t sample = blockIdx.x;
    const uint32_t n = args.n;
    const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
    const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
    const uint32_t acc_address = lds_address(lds_acc);     // 8N-aligned, see blind_rotate_kernel
    if (PL::SWAP && (acc_address & (8u * N - 1u))) __builtin_trap();
    const uint32_t bL = args.base_log * L;
    const uint32_t dbias = decomp_bias_constant(bL <= 31 ? bL : 31);
    if (args.fair_shift && threadIdx.x == 0) {
        uint32_t fair_parity;
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        fair_parity = atomicAdd(&g_cu_tickets[((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu)], 1u) & 1u;
        reinterpret_cast<uint32_t*>(lds_x)[0] = fair_parity;     // a plane word, read back before the first transform (no static LDS: the
                                                                 // accumulator copy must stay 8N-aligned at the start of the dynamic segment)
    }

    for (uint32_t i = threadIdx.x; i < n; i += CFG::THREADS) {
        const uint64_t a = lwe[i];
        lds_d[i] = a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN);
    }

    typename std::conditional<CFG::LDS_TWIDDLES, FftSwapLdsConsts<PL>,
                              typename std::conditional<CFG::TW_IN_LDS, FftHybridConsts<PL>, FftConsts<PL>>::type>::type fc;
    if constexpr (CFG::LDS_TWIDDLES) {
        FftSwapLdsConsts<PL>::fill(lds_tw, threadIdx.x, CFG::THREADS);     // visible after the barrier below
        fc.init(lds_tw, tau);
    } else if constexpr (CFG::TW_IN_LDS) {
        FftHybridConsts<PL>::fill(lds_tw, threadIdx.x, CFG::THREADS);      // visible after the barrier below
        fc.init(lds_tw, tau);
    } else {
        fft_init_consts<PL>(fc, tau);
    }
    cplx twist[R], twbias[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        double sn, cs;
        sincospi((double)PL::point(tau, m) / (double)N, &sn, &cs);
        twist[m].re = cs; twist[m].im = sn;
        const double cb = -(double)((1u << (args.base_log * L - 1)) - 1u);
        twbias[m].re = cb * (cs - sn);
        twbias[m].im = cb * (cs + sn);
    }

    constexpr bool ACC_LDS = CFG::ACC_IN_LDS;
    uint64_t acc_lo[ACC_LDS ? 1 : K1][R], acc_hi[ACC_LDS ? 1 : K1][R];
    auto own_slot = [&](int p, int m, int h) { return (size_t)p * N + acc_slot_of<PL>(PL::point(tau, m) + h * P); };
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
                    v = neg ? (0 - v) : v;
                    if constexpr (!ACC_LDS) { if (h == 0) acc_lo[p][m] = v; else acc_hi[p][m] = v; }
                    lds_acc[(size_t)p * N + acc_slot_of<PL>(j)] = v;
                }
    }
    __syncthreads();

    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)L * K1 * K1 * P;
    const uint32_t key_off = (uint32_t)tau * 16u;
    const auto key_rsrc = key_resource(args.fbsk, (size_t)n * GGSW_ELEMS * 16);

    const uint32_t fair_shift = args.fair_shift;
    uint32_t fair_turn = 0;
    if (fair_shift) {
        fair_turn = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uint32_t*>(lds_x)[0]);
        __syncthreads();
    }
    uint32_t d_next = lds_d[0];
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_next);   // workgroup-uniform: scalar
        d_next = lds_d[i + 1 < n ? i + 1 : i];
        if (d == 0xFFFFFFFFu) continue;
        if (fair_shift) {
            if ((((uint32_t)(__builtin_amdgcn_s_memrealtime() >> fair_shift)) ^ fair_turn) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        const Rotation<PL, LOGN> rot(d, tau);
        const double2* bk0 = fbsk + (size_t)i * GGSW_ELEMS;

        // key rows by buffer loads: scalar row offset + one per-thread byte offset (see blind_rotate_kernel)
        double2 bpre[CFG::PREFETCH_ALL ? K1 : 1][CFG::PREFETCH_ALL ? K1 : 1][R];
        if (CFG::PREFETCH_ALL) {
#pragma unroll
            for (int row = 0; row < K1; row++)
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int rho = 0; rho < R; rho++)
                        bpre[row][col][rho] = key_load(key_rsrc, key_off,
                            (uint32_t)((i * GGSW_ELEMS + (size_t)(L - 1) * K1 * K1 * P + ((size_t)row * K1 + col) * P + rho * T) * 16));
        }

        uint32_t st_lo[K1][R], st_hi[K1][R];
        uint32_t row_base[K1];
#pragma unroll
        for (int p = 0; p < K1; p++) {
            row_base[p] = (acc_address + (uint32_t)p * 8u * N) | rot.rbits8;
            if constexpr (PL::SWAP) asm volatile("" : "+s"(row_base[p]));
        }
#pragma unroll
        for (int p = 0; p < K1; p++)
#pragma unroll
            for (int m = 0; m < R; m++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t sm32;
                    uint64_t gathered;
                    if constexpr (PL::SWAP) {
                        uint32_t address;
                        rot.source_bytes(m, h, row_base[p], address, sm32);
                        gathered = lds_load_u64(address);
                    } else {
                        uint32_t slot;
                        rot.source(tau, m, h, slot, sm32);
                        gathered = lds_acc[(size_t)p * N + slot];
                    }
                    const uint64_t sm = ((uint64_t)sm32 << 32) | sm32;
                    const uint64_t v = (gathered ^ sm) - sm;
                    uint64_t own;
                    if constexpr (ACC_LDS) own = lds_acc[own_slot(p, m, h)];
                    else own = h == 0 ? acc_lo[p][m] : acc_hi[p][m];
                    const uint32_t st = L == 1 ? decomp_single_biased(v - own, bL, dbias) : decomp_init_state(v - own, bL);
                    if (h == 0) st_lo[p][m] = st; else st_hi[p][m] = st;
                }

        cplx outf[K1][R];
#pragma unroll
        for (int it = 0; it < L; it++) {
            const int lvl_idx = L - 1 - it;
            cplx x[K1][R];
#pragma unroll
            for (int p = 0; p < K1; p++)
#pragma unroll
                for (int m = 0; m < R; m++) {
                    if (L == 1) {
                        if constexpr (CFG::LDS_TWIDDLES) {       // no register for the folded bias: subtract it as an integer first
                            const int32_t cbi = (int32_t)((1u << (args.base_log * L - 1)) - 1u);
                            cplx z;
                            z.re = (double)((int32_t)st_lo[p][m] - cbi);
                            z.im = (double)((int32_t)st_hi[p][m] - cbi);
                            x[p][m] = cmul(z, twist[m]);
                        } else {
                            x[p][m] = digit_point(st_lo[p][m], st_hi[p][m], twist[m], twbias[m]);
                        }
                        continue;
                    }
                    cplx z;
                    z.re = (double)decomp_next_digit(st_lo[p][m], args.base_log);
                    z.im = (double)decomp_next_digit(st_hi[p][m], args.base_log);
                    x[p][m] = cmul(z, twist[m]);
                }
            fft_forward_multi<PL, K1>(x, fc, lds_x, CFG::GROUP_SLOTS, CFG::PLANE, tau);
            const double2* bk = bk0 + (size_t)lvl_idx * K1 * K1 * P;
#pragma unroll
            for (int row = 0; row < K1; row++) {
                // FftSwap9 build of this kernel (three polynomials, no key prefetch): a GGSW row is requested as one batch of
                // K1 R loads and only then multiplied -- left to itself the compiler emitted load, s_waitcnt vmcnt(0), four
                // FMAs, 36 times per step (10.0 instead of 4.95 ms per 512 LWEs, profiles/r04_n1024.txt)
                double2 krow[CFG::ROW_BATCH ? K1 : 1][CFG::ROW_BATCH ? R : 1];
                if constexpr (CFG::ROW_BATCH) {
#pragma unroll
                    for (int col = 0; col < K1; col++)
#pragma unroll
                        for (int rho = 0; rho < R; rho++)
                            krow[col][rho] = key_load(key_rsrc, key_off,
                                (uint32_t)((i * GGSW_ELEMS + (size_t)lvl_idx * K1 * K1 * P + ((size_t)row * K1 + col) * P + rho * T) * 16));
                    FHE_PIN_ORDER();
                }
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const double2 bv = CFG::ROW_BATCH ? krow[CFG::ROW_BATCH ? col : 0][CFG::ROW_BATCH ? rho : 0]
                                           : (CFG::PREFETCH_ALL && it == 0)
                                               ? bpre[CFG::PREFETCH_ALL ? row : 0][CFG::PREFETCH_ALL ? col : 0][rho]
                                               : bk[((size_t)row * K1 + col) * P + rho * T + tau];
                        const cplx f = x[row][rho];
                        if (it == 0 && row == 0) {
                            outf[col][rho].re = bv.x * f.re - bv.y * f.im;
                            outf[col][rho].im = bv.x * f.im + bv.y * f.re;
                        } else {
                            outf[col][rho].re = fma(bv.x, f.re, fma(-bv.y, f.im, outf[col][rho].re));
                            outf[col][rho].im = fma(bv.x, f.im, fma(bv.y, f.re, outf[col][rho].im));
                        }
                    }
            }
            if (it + 1 < L) __syncthreads();   // next level's pass-0 writes vs this level's last reads
        }

        fft_inverse_multi<PL, K1>(outf, fc, lds_x, CFG::GROUP_SLOTS, CFG::PLANE, tau);
#pragma unroll
        for (int p = 0; p < K1; p++)
#pragma unroll
            for (int m = 0; m < R; m++) {
                cplx t = cmul_conj(outf[p][m], twist[m]);
                if constexpr (ACC_LDS) {
                    lds_acc[own_slot(p, m, 0)] += from_torus(t.re);
                    lds_acc[own_slot(p, m, 1)] += from_torus(t.im);
                } else {
                acc_lo[p][m] += from_torus(t.re);
                acc_hi[p][m] += from_torus(t.im);
                lds_acc[(size_t)p * N + acc_slot_of<PL>(PL::point(tau, m))] = acc_lo[p][m];
                lds_acc[(size_t)p * N + acc_slot_of<PL>(PL::point(tau, m) + P)] = acc_hi[p][m];
                }
                FHE_PIN_ORDER();
            }
        __syncthreads();
    }

#ifdef FHESTR_WALL
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_wall[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime() - wall_t0;
        g_wall[blockIdx.x * 2 + 1] = wall_t0;
    }
#endif
    uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
    for (int p = 0; p < K1; p++)
#pragma unroll
        for (int m = 0; m < R; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t j = PL::point(tau, m) + h * P;
                uint64_t v;
                if constexpr (ACC_LDS) v = lds_acc[own_slot(p, m, h)];
                else v = h == 0 ? acc_lo[p][m] : acc_hi[p][m];
                if (p == K1 - 1) {
                    if (j == 0) out[(size_t)(K1 - 1) * N] = v;
                } else {
                    if (j == 0) out[(size_t)p * N] = v;
                    else out[(size_t)p * N + (N - j)] = 0 - v;
                }
            }
}

// Standard-domain polynomials -> the wide kernel's Fourier layout when that kernel has a plan of its own (BrWideCfg::OWN_PLAN).
// One polynomial per workgroup; the arithmetic of bsk_convert_kernel.
template <int LOGN, int LOGR, int K1, int L>
__global__ void __launch_bounds__((BrWideCfg<LOGN, LOGR, K1, L>::THREADS))
bsk_convert_wide_kernel(const uint64_t* __restrict__ bsk_std, double* __restrict__ fbsk, uint32_t n_polys) {
    using CFG = BrWideCfg<LOGN, LOGR, K1, L>;
    using PL = typename CFG::PL;
    constexpr int N = CFG::N, P = CFG::P, R = CFG::R, T = CFG::T;
    extern __shared__ __align__(16) unsigned char smem[];
    double* planes = reinterpret_cast<double*>(smem);
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
