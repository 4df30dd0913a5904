// pbs_cluster_kernels.hip.h -- blind rotation for N >= 16384 with SEVERAL compute units per LWE
// (PARAM_MESSAGE_4_CARRY_4_KS_PBS: N = 32768, shortint/parameters/mod.rs:1063-1077).
//
// Same algorithm as pbs_large_kernels.hip.h (fft64/crypto/bootstrap.rs:242-364, ggsw.rs:477-598) and the
// same Fourier key (bsk_convert_large_kernel), other placement: a CLUSTER of C workgroups that sit on ONE
// XCD owns one LWE.  Workgroup m of the cluster
//   * keeps the accumulator coefficients of 16 columns of the P1 x P2 coefficient matrix in VGPRs for the
//     whole blind rotation (16 u64 per thread: no accumulator re-reads),
//   * phase 1: gathers the rotated accumulator from the cluster's published copy, decomposes, runs its
//     columns' forward transforms, twiddles, stores the column spectra into the cluster's T matrix,
//   * phase 2: loads ITS P1/C rows of T, row transforms, multiply-accumulate against the GGSW rows (two
//     thread groups per row split the L(k+1) digit polynomials and swap one partial sum through LDS),
//     inverse row transforms, stores in place,
//   * phase 3: inverse column transforms of its columns, torus rounding, accumulate, publish.
// Per LWE-step the cluster moves 1.5 MB out and 1.5 MB back (+ the 2 MB GGSW all clusters share) through
// the XCD's L2 instead of 5 MB through HBM by one CU, and an LWE's step is worked on by C CUs at once.
//
// Hand-over between the workgroups of a cluster: they are on the same XCD BY CONSTRUCTION -- every workgroup
// reads its XCC id (s_getreg_b32 HW_REG_XCC_ID) and takes a ticket from that XCD's counter, clusters are
// formed from consecutive tickets of one XCD once the whole grid has arrived -- so the XCD's L2 is their
// point of coherence: plain stores (write through the CU's L1 into L2), every storing wave's
// s_waitcnt vmcnt(0), a workgroup barrier, then the workgroup's epoch flag (plain store); consumers poll the
// C flags and read the exchanged bytes with sc1 loads, which bypass the vector L1 and are served by L2
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  Nothing depends on
// WHICH workgroups share an XCD: leftover workgroups exit, LWEs are dealt over the clusters that did form,
// every spin is bounded and raises ctl->error instead of hanging.
#pragma once
#include "pbs_large_kernels.hip.h"

namespace fhe {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int CLUSTER_MAX = 64;          // clusters a launch can form (256 CUs / 4)
constexpr int CLUSTER_MAX_MEMBERS = 16;
constexpr uint32_t CLUSTER_SPIN_LIMIT = 1u << 22;   // polls (~0.5 us each) before a wait gives up

struct ClusterStatus {                   // sticky: the host reads it at its next synchronisation (Engine::cluster_check)
    uint32_t error;                      // != 0: some launch gave up on a hand-over
    uint32_t clusters;                   // clusters the last launch formed
};
struct ClusterCtl {                      // zeroed by the host before every launch
    uint32_t arrived;                    // workgroups that took a ticket
    uint32_t error;                      // != 0: a wait timed out (results are garbage, the launch still ends)
    uint32_t pad[30];
    uint32_t xcd_count[8][32];           // tickets per XCD, one 128-byte line each
    uint32_t flags[CLUSTER_MAX][CLUSTER_MAX_MEMBERS][32];   // epoch flag of every cluster member, a line each
};

template <int LOGN, int K1, int L>
struct BrClusterCfg {
    static_assert(K1 == 2, "cluster kernel: k = 1");
    using LC = BrLargeCfg<LOGN, K1, L>;
    using PA = typename LC::PA;
    using PB = typename LC::PB;
    static constexpr int N = LC::N, P = LC::P, P1 = LC::P1, P2 = LC::P2, LOGP1 = LC::LOGP1, LOGP2 = LC::LOGP2;
    static constexpr int R = 8, TA = LC::TA, TB = LC::TB, THREADS = LC::THREADS;
    static constexpr int GROUPS_A = THREADS / TA;          // column transforms in flight per workgroup
    static constexpr int GROUPS_B = THREADS / TB;          // row transforms in flight per workgroup
    static constexpr int COLS = GROUPS_A / K1;             // columns a workgroup owns
    static constexpr int C = P2 / COLS;                    // workgroups per LWE
    static constexpr int ROWS = P1 / C;                    // rows a workgroup owns
    static_assert(GROUPS_B == 2 * ROWS, "two thread groups per row");
    static_assert(C <= CLUSTER_MAX_MEMBERS, "cluster too wide");
    static constexpr int U = L * K1, UH = U / 2;           // digit polynomials; per half of a row's group pair
    static constexpr int PITCH = P2 + 8;                   // c64 per T row: + 128 bytes so rows start on different L2 channels
    static constexpr size_t WS_T = (size_t)U * P1 * PITCH * 16;     // T[u][row'][col] c64
    static constexpr size_t WS_ACC = (size_t)K1 * N * 8;            // published accumulator [p][col][2*P1] u64
    static constexpr size_t WS_BYTES = WS_T + WS_ACC;               // per cluster
    static constexpr size_t LDS_BYTES = LC::LDS_BYTES;              // + 4 n: modulus-switched mask
    // T rows are stored permuted: the 16 rows a thread group writes with one store instruction (fixed register
    // slot rho, A = slot_addr(tau, rho) = 8 tau + rho) are neighbours, and a workgroup's ROWS rows are one block
    __host__ __device__ static constexpr int row_perm(int A) { return ((A & 7) << (LOGP1 - 3)) | (A >> 3); }
    __host__ __device__ static constexpr int row_unperm(int Ap) { return ((Ap & ((P1 >> 3) - 1)) << 3) | (Ap >> (LOGP1 - 3)); }
};

struct BlindRotateClusterArgs {
    BlindRotateArgs base;
    unsigned char* workspace;    // clusters * WS_BYTES
    ClusterCtl* ctl;
    ClusterStatus* status;
};

template <class RSRC>
__device__ __forceinline__ double2 load_sc1_b128(RSRC rsrc, uint32_t voff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 16 /* sc1: past the L1, served by L2 */);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
template <class RSRC>
__device__ __forceinline__ uint64_t load_sc1_b64(RSRC rsrc, uint32_t voff) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, 0, 16);
    return ((uint64_t)v.y << 32) | v.x;
}

// All C workgroups of the cluster have stored what the next phase reads.  A wait that does not end within
// CLUSTER_SPIN_LIMIT polls marks the launch dead (LDS word + ctl->error + the sticky status): every later wait of
// every cluster returns at once, the kernel drains with garbage and the host reports it (Engine::cluster_check).
template <int C>
__device__ __forceinline__ void cluster_sync(uint32_t* flags, uint32_t member, uint32_t& epoch, volatile uint32_t* s_dead,
                                             ClusterCtl* ctl, ClusterStatus* status) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached L2
    __syncthreads();
    ++epoch;
    if (threadIdx.x < 64 && !*s_dead) {
        const uint32_t lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(flags + member * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t spins = 0;
        for (;;) {
            const uint32_t v = lane < (uint32_t)C ? __hip_atomic_load(flags + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all((int32_t)(v - epoch) >= 0)) break;
            ++spins;
            const bool others_gave_up = (spins & 1023u) == 0 && __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins > CLUSTER_SPIN_LIMIT || others_gave_up) {
                if (lane == 0) {
                    __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&status->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_dead = 1u;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int LOGN, int K1, int L>
__global__ void __launch_bounds__((BrClusterCfg<LOGN, K1, L>::THREADS))
blind_rotate_cluster_kernel(BlindRotateClusterArgs ca) {
    using CFG = BrClusterCfg<LOGN, K1, L>;
    using LC = typename CFG::LC;
    using PA = typename CFG::PA;
    using PB = typename CFG::PB;
    constexpr int N = CFG::N, P = CFG::P, P1 = CFG::P1, P2 = CFG::P2, R = CFG::R, NT = CFG::THREADS;
    constexpr int C = CFG::C, TA = CFG::TA, TB = CFG::TB, LOGP2 = CFG::LOGP2, PITCH = CFG::PITCH, UH = CFG::UH;
    const BlindRotateArgs& args = ca.base;
    ClusterCtl* ctl = ca.ctl;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    __shared__ uint32_t s_form[4];
    __shared__ uint32_t s_dead;

    const int tid = threadIdx.x;

    // ---- cluster formation (agent-scope atomics: valid wherever the workgroups landed) ----
    if (tid == 0) {
        s_dead = 0;
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const uint32_t ticket = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        bool ok = true;
        while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            if (++spins > CLUSTER_SPIN_LIMIT) { ok = false; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        uint32_t base = 0, total = 0, mine = 0;
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t formed = __hip_atomic_load(&ctl->xcd_count[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (uint32_t)C;
            if (x < xcc) base += formed;
            if (x == xcc) mine = formed;
            total += formed;
        }
        if (!ok) {
            __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ca.status->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total = 0;
        }
        const uint32_t local = ticket / (uint32_t)C;
        const bool member_of_one = ok && local < mine && base + local < (uint32_t)CLUSTER_MAX;
        s_form[0] = member_of_one ? base + local : 0xFFFFFFFFu;
        s_form[1] = ticket % (uint32_t)C;
        s_form[2] = total < (uint32_t)CLUSTER_MAX ? total : (uint32_t)CLUSTER_MAX;
        if (blockIdx.x == 0) ca.status->clusters = s_form[2];
    }
    __syncthreads();
    const uint32_t cluster = s_form[0], member = s_form[1], n_clusters = s_form[2];
    if (cluster == 0xFFFFFFFFu) return;          // not part of a complete cluster: the whole workgroup leaves

    // ---- thread roles ----
    // phases 1 and 3: group (p, bl) of TA threads owns column b = member * COLS + bl of polynomial p
    const int gA = tid / TA, tauA = tid % TA;
    const int pA = gA / CFG::COLS, b = (int)member * CFG::COLS + gA % CFG::COLS;
    // phase 2: groups 2 rl + h of TB threads own row A' = member * ROWS + rl; half h takes UH digit polynomials
    const int gB = tid / TB, tauB = tid % TB;
    const int hB = gB & 1, rowp = (int)member * CFG::ROWS + (gB >> 1);
    const int rowA = CFG::row_unperm(rowp);
    double* areA = lds + (size_t)gA * LC::SLOTS_A;
    double* aimA = areA + P1 + 2;
    double* breB = lds + (size_t)gB * LC::SLOTS_B;
    double* bimB = breB + P2 + 2;
    double* preB = lds + (size_t)(gB ^ 1) * LC::SLOTS_B;       // the partner group's planes
    double* pimB = preB + P2 + 2;

    double2* twa = reinterpret_cast<double2*>(smem + LC::LDS_PLANES + LC::LDS_ROOTS);
    double2* twb = twa + FftTwiddleTable<PA>::ENTRIES;
    FftTwiddleTable<PA>::fill(twa, tid, NT);
    FftTwiddleTable<PB>::fill(twb, tid, NT);
    const FftTwiddleTable<PA> fca{twa, tauA};
    const FftTwiddleTable<PB> fcb{twb, tauB};
    RootTable<LC> roots;
    roots.init(reinterpret_cast<double2*>(smem + LC::LDS_PLANES), tid, NT);
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(smem + LC::LDS_BYTES);     // [n] modulus-switched mask

    unsigned char* ws = ca.workspace + (size_t)cluster * CFG::WS_BYTES;
    double2* Tm = reinterpret_cast<double2*>(ws);                              // [U][P1][PITCH]
    uint64_t* accpub = reinterpret_cast<uint64_t*>(ws + CFG::WS_T);           // [K1][P2][2*P1]
    const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)CFG::WS_T, 0x00020000);
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + CFG::WS_T, 0, (int)CFG::WS_ACC, 0x00020000);
    uint32_t* flags = &ctl->flags[cluster][0][0];
    uint32_t epoch = 0;

    const uint32_t n = args.n;
    const uint32_t bL = args.base_log * L;
    const double2* fbsk = reinterpret_cast<const double2*>(args.fbsk);
    constexpr size_t GGSW_ELEMS = (size_t)L * K1 * K1 * P;
    const int q1row = freq_of_addr<PA>(rowA);

    for (uint32_t sample = cluster; sample < args.batch; sample += n_clusters) {
        const uint64_t* lwe = args.lwe_small + (size_t)sample * (n + 1);
        const uint64_t* lut = args.luts + (size_t)(args.lut_idx ? args.lut_idx[sample] : 0) * K1 * N;
        __syncthreads();          // lds_d of the previous sample is no longer read
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t a = lwe[i];
            lds_d[i] = a == 0 ? 0xFFFFFFFFu : modulus_switch(a, LOGN);
        }

        // acc <- LUT * X^{-ms(body)}: this thread's 2R coefficients j = h P + (tau + TA m) P2 + b
        uint64_t own[2 * R];
        {
            const uint32_t d = modulus_switch(lwe[n], LOGN);
            const uint32_t rem = d & (N - 1);
            const bool odd = (d >> LOGN) & 1;
#pragma unroll
            for (int m = 0; m < R; m++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t j = (uint32_t)h * P + (uint32_t)(tauA + TA * m) * P2 + (uint32_t)b;
                    const uint32_t src = (j + rem) & (N - 1);
                    const bool neg = ((j + rem) >= (uint32_t)N) != odd;
                    const uint64_t v = lut[(size_t)pA * N + src];
                    own[2 * m + h] = neg ? (0 - v) : v;
                    accpub[((size_t)pA * P2 + b) * (2 * P1) + h * P1 + tauA + TA * m] = own[2 * m + h];
                }
            }
        }
        cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status);

        FHE_STAMP_DECL;
        FHE_STAMP(-1);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t d = lds_d[i];
            if (d == 0xFFFFFFFFu) continue;       // a_i == 0 (bootstrap.rs:281): the same for the whole cluster
            const uint32_t rem = d & (N - 1);
            const bool odd = (d >> LOGN) & 1;

            // ---- phase 1: rotate, subtract, decompose, twist, column transforms, twiddle -> T ----
            {
                using state_t = typename std::conditional<(L >= 3), uint64_t, uint32_t>::type;
                state_t st_lo[R], st_hi[R];
                const uint32_t bsrc = ((uint32_t)b - rem) & (P2 - 1);
                const uint32_t colbase = ((uint32_t)pA * P2 + bsrc) * (2 * P1);
                uint64_t rot[2 * R];
#pragma unroll
                for (int m = 0; m < R; m++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t j = (uint32_t)h * P + (uint32_t)(tauA + TA * m) * P2 + (uint32_t)b;
                        const uint32_t e = ((j - rem) & (N - 1)) >> LOGP2;
                        rot[2 * m + h] = load_sc1_b64(a_rsrc, (colbase + e) * 8);
                    }
                }
#pragma unroll
                for (int m = 0; m < R; m++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t j = (uint32_t)h * P + (uint32_t)(tauA + TA * m) * P2 + (uint32_t)b;
                        const bool neg = (j < rem) != odd;
                        uint64_t v = rot[2 * m + h];
                        v = neg ? (0 - v) : v;
                        const uint64_t ct1 = v - own[2 * m + h];
                        state_t st;
                        if constexpr (L >= 3) st = decomp_init_state64(ct1, bL);
                        else st = decomp_init_state(ct1, bL);
                        if (h == 0) st_lo[m] = st; else st_hi[m] = st;
                    }
                }
#pragma unroll
                for (int it = 0; it < L; it++) {
                    cplx x[R];
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const int j = (tauA + TA * m) * P2 + b;
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
                    fft_forward<PA>(x, fca, areA, aimA, tauA);
                    double2* dst = Tm + (size_t)(it * K1 + pA) * P1 * PITCH;
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const int A = slot_addr<PA>(tauA, rho);
                        const int q1 = freq_of_addr<PA>(A);
                        const cplx w = roots.get(0u - 4u * (uint32_t)(q1 * b));     // e^{-2 pi i q1 b / P}
                        const cplx v = cmul(x[rho], w);
                        dst[(size_t)CFG::row_perm(A) * PITCH + b] = make_double2(v.re, v.im);
                    }
                }
            }
            FHE_STAMP(1);
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status);
            FHE_STAMP(2);

            // ---- phase 2: row transforms, multiply-accumulate with the GGSW, inverse row transforms, in place ----
            {
                const double2* bk0 = fbsk + (size_t)i * GGSW_ELEMS;
                cplx outf[K1][R];
#pragma unroll
                for (int uu = 0; uu < UH; uu++) {
                    const int u = hB * UH + uu;            // u = it * K1 + row (ggsw.rs:524 order within a half)
                    const int it = u / K1, row = u % K1;
                    const int lvl_idx = L - 1 - it;
                    cplx x[R];
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const double2 v = load_sc1_b128(t_rsrc, (uint32_t)(((u * P1 + rowp) * PITCH + tauB + TB * m) * 16));
                        x[m].re = v.x; x[m].im = v.y;
                    }
                    double2 bv[K1][R];
#pragma unroll
                    for (int col = 0; col < K1; col++) {
                        const double2* bk = bk0 + (((size_t)lvl_idx * K1 + row) * K1 + col) * P + (size_t)rowA * P2;
#pragma unroll
                        for (int rho = 0; rho < R; rho++) bv[col][rho] = bk[rho * TB + tauB];
                    }
                    fft_forward<PB>(x, fcb, breB, bimB, tauB);
#pragma unroll
                    for (int col = 0; col < K1; col++) {
#pragma unroll
                        for (int rho = 0; rho < R; rho++) {
                            const double2 b2 = bv[col][rho];
                            if (uu == 0) {
                                outf[col][rho].re = b2.x * x[rho].re - b2.y * x[rho].im;
                                outf[col][rho].im = b2.x * x[rho].im + b2.y * x[rho].re;
                            } else {
                                outf[col][rho].re = fma(b2.x, x[rho].re, fma(-b2.y, x[rho].im, outf[col][rho].re));
                                outf[col][rho].im = fma(b2.x, x[rho].im, fma(b2.y, x[rho].re, outf[col][rho].im));
                            }
                        }
                    }
                }
                // the halves swap the partial sum of the column the OTHER one finishes (same wavefront: LDS in order)
                wave_local_fence();
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const cplx v = hB ? outf[0][rho] : outf[1][rho];
                    breB[rho * TB + tauB] = v.re;
                    bimB[rho * TB + tauB] = v.im;
                }
                wave_local_fence();
                cplx mine[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const cplx o = hB ? outf[1][rho] : outf[0][rho];
                    mine[rho].re = o.re + preB[rho * TB + tauB];
                    mine[rho].im = o.im + pimB[rho * TB + tauB];
                }
                wave_local_fence();
                fft_inverse<PB>(mine, fcb, breB, bimB, tauB);
                double2* dpoly = Tm + ((size_t)(hB * UH) * P1 + rowp) * PITCH;
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const int bb = tauB + TB * m;
                    const cplx w = roots.get(4u * (uint32_t)(q1row * bb));   // conj of the forward twiddle
                    const cplx v = cmul(mine[m], w);
                    dpoly[bb] = make_double2(v.re, v.im);
                }
            }
            FHE_STAMP(3);
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status);
            FHE_STAMP(4);

            // ---- phase 3: inverse column transforms, untwist, torus rounding, accumulate, publish ----
            {
                cplx x[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const int A = slot_addr<PA>(tauA, rho);
                    const double2 v = load_sc1_b128(t_rsrc, (uint32_t)((((pA * UH) * P1 + CFG::row_perm(A)) * PITCH + b) * 16));
                    x[rho].re = v.x; x[rho].im = v.y;
                }
                fft_inverse<PA>(x, fca, areA, aimA, tauA);
                uint64_t* pub = accpub + ((size_t)pA * P2 + b) * (2 * P1);
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const int j = (tauA + TA * m) * P2 + b;
                    const cplx t = cmul_conj(x[m], roots.get((uint32_t)j));
                    own[2 * m] += from_torus(t.re);
                    own[2 * m + 1] += from_torus(t.im);
                    pub[tauA + TA * m] = own[2 * m];
                    pub[P1 + tauA + TA * m] = own[2 * m + 1];
                }
            }
            FHE_STAMP(5);
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status);
            FHE_STAMP(6);
        }
#ifdef FHESTR_STAMPS
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && sample == cluster)
            for (int sg = 0; sg < STAMP_SEGS; sg++)
                g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * STAMP_SEGS + sg] = stamp_acc[sg];
#endif

        // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146), straight from the registers
        uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
        for (int m = 0; m < R; m++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t j = (uint32_t)h * P + (uint32_t)(tauA + TA * m) * P2 + (uint32_t)b;
                const uint64_t v = own[2 * m + h];
                if (pA == K1 - 1) {
                    if (j == 0) out[(size_t)(K1 - 1) * N] = v;
                } else {
                    if (j == 0) out[(size_t)pA * N] = v;
                    else out[(size_t)pA * N + (N - j)] = 0 - v;
                }
            }
        }
    }
}

}  // namespace fhe
