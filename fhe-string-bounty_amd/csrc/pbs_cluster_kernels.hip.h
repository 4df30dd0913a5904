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
// Per LWE-step the cluster moves 2 MB out and 2 MB back (T 1 MB, its in-place inverse half 0.5 MB, the published
// accumulator 0.5 MB; + the 2 MB GGSW all clusters of an XCD share) through the XCD's L2 instead of 5 MB through HBM by
// one CU, and an LWE's step is worked on by C CUs at once.  With one or two clusters per XCD the exchange stays in L2;
// at four (6 MB per XCD) it spills to the Infinity Cache and the fabric binds.
//
// Hand-over between the workgroups of a cluster: they are on the same XCD BY CONSTRUCTION -- every workgroup
// reads its XCC id (s_getreg_b32 HW_REG_XCC_ID) and takes a ticket from that XCD's counter, clusters are
// formed from consecutive tickets of one XCD once the whole grid has arrived -- so the XCD's L2 is their
// point of coherence: plain stores (write through the CU's L1 into L2), every storing wave's own
// s_waitcnt vmcnt, an arrival count in LDS whose last wave publishes the workgroup's epoch flag (plain store); every
// wave polls the cluster's flag line and reads the exchanged bytes with sc1 loads, which bypass the vector L1 and are
// served by L2 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): no workgroup
// barrier, no fence and no agent-scope atomic inside the CMUX loop (cluster_sync below).  Nothing depends on
// WHICH workgroups share an XCD: leftover workgroups exit, LWEs are dealt over the clusters that did form,
// every spin is bounded and raises ctl->error instead of hanging.
#pragma once
#include "pbs_large_kernels.hip.h"

#ifndef FHESTR_CL_KEY_AUX
#define FHESTR_CL_KEY_AUX 0       // Fourier-key loads: default policy (shared by the clusters of an XCD through L2)
#endif

// cache policy of the loads of exchanged data: sc1 = past the vector L1, served by the XCD's L2.  (Measured and dropped:
// sc1 | nt on these loads, nt on the key loads, and starting half of an XCD's clusters half a step late -- all within
// -4 % .. 0 % at 32 clusters, same L2 miss counts)
#define FHESTR_CL_XCHG_AUX 16

namespace fhe {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int CLUSTER_MAX = 64;          // clusters a launch can form (256 CUs / 4)
constexpr int CLUSTER_MAX_MEMBERS = 32;  // the epoch flags of a cluster share ONE 128-byte line
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
    uint32_t flags[CLUSTER_MAX][CLUSTER_MAX_MEMBERS];       // epoch flag of every cluster member; one line per cluster
    // blind_rotate_xcd_kernel (two workgroups per CU): workgroups per compute unit and per "slot" (first / second arrival on its CU)
    uint32_t slot_count[8][32];          // per XCD: [0] first arrivals, [1] second arrivals; one line each
    uint32_t cu_count[8][256];           // per XCD and HW_ID{se, sh, cu}
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
    // LDS: exchange planes (all real planes, then all imaginary ones: too far apart for hipcc to fuse a re/im pair
    // into ds_read2_b64, half the LDS rate of two plain reads), per-thread constant tables, FFT twiddles, mask
    static constexpr int SLOTS_A = P1 + 16, SLOTS_B = P2 + 16;     // neighbouring transforms 16 slots apart mod 32
    static constexpr int IM_A = GROUPS_A * SLOTS_A + 2, IM_B = GROUPS_B * SLOTS_B + 2;   // not a multiple of 64 slots (ds_read2st64_b64)
    static constexpr size_t LDS_PLANES = (size_t)8 * (IM_A > IM_B ? 2 * IM_A : 2 * IM_B);
    static constexpr size_t LDS_E1 = (size_t)P1 * 16;               // e^{i pi a P2 / N}: twist, row part
    static constexpr size_t LDS_TW2 = (size_t)COLS * P1 * 16;       // [column][rho][tau]: four-step twiddle * column part of the twist
    static constexpr size_t LDS_TW3 = (size_t)ROWS * P2 * 16;       // [row][m][tau]: conjugate twiddle * conjugate column twist
    static constexpr size_t LDS_TWA = (size_t)FftTwiddleTable<PA>::ENTRIES * 16, LDS_TWB = (size_t)FftTwiddleTable<PB>::ENTRIES * 16;
    static constexpr size_t LDS_BYTES = LDS_PLANES + LDS_E1 + LDS_TW2 + LDS_TW3 + LDS_TWA + LDS_TWB;   // + 4 n: modulus-switched mask
    static_assert(LDS_BYTES + 4 * 1280 <= 160 * 1024, "LDS budget");
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
    uint32_t spin_limit;         // polls before a wait gives up (CLUSTER_SPIN_LIMIT; tests lower it)
    uint32_t test_fault;         // -DFHESTR_TEST_HOOKS builds only: member 1 of cluster 0 never publishes its flag of this epoch (0 = off)
};

template <class RSRC>
__device__ __forceinline__ double2 load_sc1_b128(RSRC rsrc, uint32_t voff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, FHESTR_CL_XCHG_AUX);
    double2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
template <class RSRC>
__device__ __forceinline__ uint64_t load_sc1_b64(RSRC rsrc, uint32_t voff) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, 0, FHESTR_CL_XCHG_AUX);
    return ((uint64_t)v.y << 32) | v.x;
}

// All C workgroups of the cluster have stored what the next phase reads.  A wait that does not end within
// CLUSTER_SPIN_LIMIT polls marks the launch dead (LDS word + ctl->error + the sticky status): every later wait of
// every cluster returns at once, the kernel drains with garbage and the host reports it (Engine::cluster_check).
// Every wave drains its vector-memory queue completely (vmcnt(0)): waiting for "all but the N youngest" operations would
// rest on the compiler placing nothing -- no scratch spill either -- behind the caller's prefetches.
#ifndef FHESTR_CL_SYNC
#define FHESTR_CL_SYNC 1
#endif
#if FHESTR_CL_SYNC == 1
// No workgroup barrier anywhere in the hand-over: every wave drains its own stores, counts itself in on an LDS counter
// (the wave that completes the count publishes the workgroup's epoch flag) and then polls the cluster's flag line
// itself.  A wave that sees all C flags at this epoch knows that every wave of every member -- its own workgroup's
// included -- has finished the phase, so the LDS planes may be reused as well.  The C flags of a cluster share one
// 128-byte line (byte-masked stores into L2; one request per poll).
// `need` (round 4): bit m set = this wave waits for member m.  The hand-over before a rotation gather needs only the (at
// most two) members that own the wave's source columns: a global barrier becomes a local dependency there, and members
// that run ahead absorb the jitter of the others.  All members still count every hand-over (epochs stay in step); a wave
// that waits for fewer members may only touch LDS and global data those members' phase cannot still be using.
template <int C, uint32_t WAVES = 8>
__device__ __forceinline__ void cluster_sync(uint32_t* flags, uint32_t member, uint32_t& epoch, uint32_t* s_dead_generic,
                                             ClusterCtl* ctl, ClusterStatus* status, uint32_t spin_limit, uint32_t mute_epoch,
                                             uint32_t need = 0xFFFFFFFFu) {
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32_t;
    lds_vu32_t* s_dead = (lds_vu32_t*)(uintptr_t)lds_address(s_dead_generic);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's stores have reached L2
    ++epoch;
    if (*s_dead) return;
    // Two arrival counters behind the dead word, used in turn by even and odd epochs and reset by the wave that completes the
    // count: with `need`, a wave may be ONE hand-over ahead of its slowest workgroup-mate (never two: the hand-over after a
    // partial one waits for every member, this workgroup included), so arrivals of two consecutive epochs can interleave and
    // must not share a counter (a single running count published nothing when they did: the flag of epoch e then only appeared
    // with epoch e + 1).  The reset is ordered before the resetting wave's own next arrival, which every arrival at epoch + 2 follows.
    const uint32_t arrive_address = lds_address(s_dead_generic + 1 + (epoch & 1u));
    const uint32_t lane = threadIdx.x & 63;
    uint32_t before = 0;
    if (lane == 0) {
        const uint32_t one = 1;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(before) : "v"(arrive_address), "v"(one) : "memory");
    }
    before = __builtin_amdgcn_readfirstlane(before);
    if (before + 1 == WAVES && lane == 0) {
        const uint32_t zero = 0;
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(arrive_address), "v"(zero) : "memory");
        if (epoch != mute_epoch) __hip_atomic_store(flags + member, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    uint32_t spins = 0;
    const bool polls = lane < (uint32_t)C && ((need >> lane) & 1u);
    if (need == 0u) { asm volatile("" ::: "memory"); return; }
    for (;;) {
        const uint32_t v = polls ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all((int32_t)(v - epoch) >= 0)) break;
        ++spins;
        const bool others_gave_up = (spins & 1023u) == 0 && (*s_dead || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (spins > spin_limit || others_gave_up) {
            if (lane == 0) {
                __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&status->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = 1u;
            }
            break;
        }
    }
    asm volatile("" ::: "memory");
}
#else
template <int C, uint32_t WAVES = 8>
__device__ __forceinline__ void cluster_sync(uint32_t* flags, uint32_t member, uint32_t& epoch, uint32_t* s_dead_generic,
                                             ClusterCtl* ctl, ClusterStatus* status, uint32_t spin_limit, uint32_t mute_epoch,
                                             uint32_t /* need: this variant always waits for everyone */ = 0xFFFFFFFFu) {
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32_t;     // a plain LDS access (a generic pointer would be a
    lds_vu32_t* s_dead = (lds_vu32_t*)(uintptr_t)lds_address(s_dead_generic);   // flat load: it waits for vmcnt(0) as well)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's stores have reached L2
    __syncthreads();
    ++epoch;
    if (threadIdx.x < 64 && !*s_dead) {
        const uint32_t lane = threadIdx.x;
        if (lane == 0 && epoch != mute_epoch) __hip_atomic_store(flags + member, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t spins = 0;
        for (;;) {
            const uint32_t v = lane < (uint32_t)C ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all((int32_t)(v - epoch) >= 0)) break;
            ++spins;
            const bool others_gave_up = (spins & 1023u) == 0 && __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins > spin_limit || others_gave_up) {
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

#endif

// Cluster formation, by ONE thread of every workgroup (agent-scope atomics: valid wherever the workgroups landed): a
// ticket from the counter of the XCD this workgroup runs on, a grid-wide arrival count, then clusters = runs of C
// consecutive tickets of one XCD.  s_form: [0] cluster index (0xFFFFFFFF: not part of a complete cluster), [1] member
// index, [2] clusters the launch formed.  A grid that never becomes resident as a whole raises error 2 and forms nothing.
template <int C>
__device__ __forceinline__ void cluster_join(ClusterCtl* ctl, ClusterStatus* status, uint32_t spin_limit, uint32_t* s_form, uint32_t* s_sync) {
    s_sync[0] = 0;
    s_sync[1] = 0;
    s_sync[2] = 0;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const uint32_t ticket = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    bool ok = true;
    while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > (spin_limit > (1u << 16) ? spin_limit : (1u << 16))) { ok = false; break; }
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
        __hip_atomic_store(&status->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        total = 0;
    }
    const uint32_t local = ticket / (uint32_t)C;
    const bool member_of_one = ok && local < mine && base + local < (uint32_t)CLUSTER_MAX;
    s_form[0] = member_of_one ? base + local : 0xFFFFFFFFu;
    s_form[1] = ticket % (uint32_t)C;
    s_form[2] = total < (uint32_t)CLUSTER_MAX ? total : (uint32_t)CLUSTER_MAX;
    if (blockIdx.x == 0) status->clusters = s_form[2];
}

// Formation for two workgroups per compute unit (blind_rotate_xcd_kernel): the XCD's two clusters should each have ONE
// workgroup on every CU, so that a CU always has the other LWE to work on while one waits for a hand-over.  (Consecutive
// tickets put both workgroups of 74 of 256 CUs into the same cluster: those CUs then do a double share of one cluster's
// phase and every hand-over of that cluster waits for them.)  A workgroup therefore also counts itself in on its CU
// (HW_ID se/sh/cu): first arrival -> the XCD's cluster 0, second -> cluster 1, member = rank among the arrivals of that
// kind.  If an XCD's counts are not exactly (C, C) or (C, 0) -- some CU was busy with another kernel -- that XCD falls
// back to consecutive tickets; every workgroup of the XCD reads the same counts and takes the same decision.
template <int C>
__device__ __forceinline__ void cluster_join_per_cu(ClusterCtl* ctl, ClusterStatus* status, uint32_t spin_limit, uint32_t* s_form, uint32_t* s_sync) {
    s_sync[0] = 0;
    s_sync[1] = 0;
    s_sync[2] = 0;
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc &= 7u;
    const uint32_t cu = (hw >> 8) & 0xFFu;
    const uint32_t ticket = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t slot = __hip_atomic_fetch_add(&ctl->cu_count[xcc][cu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t rank = slot < 2 ? __hip_atomic_fetch_add(&ctl->slot_count[xcc][slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFu;
    __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    bool ok = true;
    while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > (spin_limit > (1u << 16) ? spin_limit : (1u << 16))) { ok = false; break; }
        __builtin_amdgcn_s_sleep(4);
    }
    uint32_t base = 0, total = 0, mine = 0;
    bool mine_by_cu = false;
    for (uint32_t x = 0; x < 8; x++) {
        const uint32_t t = __hip_atomic_load(&ctl->xcd_count[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t s0 = __hip_atomic_load(&ctl->slot_count[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t s1 = __hip_atomic_load(&ctl->slot_count[x][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool by_cu = s0 == (uint32_t)C && (s1 == (uint32_t)C || s1 == 0u) && t == s0 + s1;
        const uint32_t formed = by_cu ? (s1 ? 2u : 1u) : t / (uint32_t)C;
        if (x < xcc) base += formed;
        if (x == xcc) { mine = formed; mine_by_cu = by_cu; }
        total += formed;
    }
    if (!ok) {
        __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&status->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        total = 0;
    }
    const uint32_t local = mine_by_cu ? slot : ticket / (uint32_t)C;
    const bool member_of_one = ok && local < mine && base + local < (uint32_t)CLUSTER_MAX;
    s_form[0] = member_of_one ? base + local : 0xFFFFFFFFu;
    s_form[1] = mine_by_cu ? rank : ticket % (uint32_t)C;
    s_form[2] = total < (uint32_t)CLUSTER_MAX ? total : (uint32_t)CLUSTER_MAX;
    if (blockIdx.x == 0) status->clusters = s_form[2];
}

template <int LOGN, int K1, int L>
__global__ void __launch_bounds__((BrClusterCfg<LOGN, K1, L>::THREADS))
blind_rotate_cluster_kernel(BlindRotateClusterArgs ca) {
    using CFG = BrClusterCfg<LOGN, K1, L>;
    using PA = typename CFG::PA;
    using PB = typename CFG::PB;
    constexpr int N = CFG::N, P = CFG::P, P1 = CFG::P1, P2 = CFG::P2, R = CFG::R, NT = CFG::THREADS;
    constexpr int C = CFG::C, TA = CFG::TA, TB = CFG::TB, LOGP2 = CFG::LOGP2, PITCH = CFG::PITCH, UH = CFG::UH;
    const BlindRotateArgs& args = ca.base;
    ClusterCtl* ctl = ca.ctl;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    __shared__ uint32_t s_form[4];
    __shared__ uint32_t s_sync[4];        // [0] dead flag, [1], [2] arrival counters of the hand-overs (even / odd epochs)
    uint32_t& s_dead = s_sync[0];

    const int tid = threadIdx.x;

    // ---- cluster formation (agent-scope atomics: valid wherever the workgroups landed) ----
    if (tid == 0) cluster_join<C>(ctl, ca.status, ca.spin_limit, s_form, s_sync);
    __syncthreads();
    const uint32_t cluster = s_form[0], member = s_form[1], n_clusters = s_form[2];
    if (cluster == 0xFFFFFFFFu) return;          // not part of a complete cluster: the whole workgroup leaves

    // ---- thread roles ----
    // phases 1 and 3: group (p, bl) of TA threads owns column b = member * COLS + bl of polynomial p
    const int gA = tid / TA, tauA = tid % TA;
    const int pA = gA / CFG::COLS, blA = gA % CFG::COLS, b = (int)member * CFG::COLS + blA;
    // phase 2: groups 2 rl + h of TB threads own row A' = member * ROWS + rl; half h takes UH digit polynomials
    const int gB = tid / TB, tauB = tid % TB;
    const int hB = gB & 1, rlB = gB >> 1, rowp = (int)member * CFG::ROWS + rlB;
    const int rowA = CFG::row_unperm(rowp);
    constexpr int IM_A = CFG::IM_A, IM_B = CFG::IM_B;
    double* areA = lds + (size_t)gA * CFG::SLOTS_A;
    double* aimA = areA + IM_A;
    double* breB = lds + (size_t)gB * CFG::SLOTS_B;
    double* bimB = breB + IM_B;
    double* preB = lds + (size_t)(gB ^ 1) * CFG::SLOTS_B;       // the partner group's planes
    double* pimB = preB + IM_B;

    double2* e1 = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES);
    double2* tw2 = e1 + P1;
    double2* tw3 = tw2 + CFG::COLS * P1;
    double2* twa = tw3 + CFG::ROWS * P2;
    double2* twb = twa + FftTwiddleTable<PA>::ENTRIES;
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(smem + CFG::LDS_BYTES);     // [n] modulus-switched mask
    FftTwiddleTable<PA>::fill(twa, tid, NT);
    FftTwiddleTable<PB>::fill(twb, tid, NT);
    const FftTwiddleTable<PA> fca{twa, tauA};
    const FftTwiddleTable<PB> fcb{twb, tauB};
    // Constant factors of this workgroup's columns and rows, one 16-byte LDS read each in the loop.  With the
    // twist e^{i pi j / N}, j = a P2 + b, split into a row part E1[a] and a column part E2[b] = e^{i pi b / N}, the
    // column part commutes with the column transforms and is folded into the four-step twiddles on either side:
    //   tw2[bl][rho][tau] = e^{-2 pi i q1(8 tau + rho) b / P} * E2[b]          (after the forward column transform)
    //   tw3[rl][m][tau]   = e^{+2 pi i q1(row) bb / P} * conj(E2[bb]), bb = tau + TB m   (after the inverse row transform)
    // Angles as integers mod 2N (units of pi / N), so sincospi sees an exact argument.
    for (int e = tid; e < P1; e += NT) {
        double sn, cs;
        sincospi((double)((uint32_t)e * P2) / (double)N, &sn, &cs);
        e1[e] = make_double2(cs, sn);
    }
    for (int e = tid; e < CFG::COLS * P1; e += NT) {
        const int bl = e / P1, rho = (e / TA) % R, tau = e % TA;
        const uint32_t bb = member * CFG::COLS + bl;
        const uint32_t q1 = (uint32_t)freq_of_addr<PA>(slot_addr<PA>(tau, rho));
        const uint32_t ang = (bb - 4u * q1 * bb) & (2u * N - 1u);
        double sn, cs;
        sincospi((double)ang / (double)N, &sn, &cs);
        tw2[e] = make_double2(cs, sn);
    }
    for (int e = tid; e < CFG::ROWS * P2; e += NT) {
        const int rl = e / P2, bb = (e % P2) / TB * TB + e % TB;      // [rl][m][tau] with bb = tau + TB m: e % P2 = m TB + tau
        const uint32_t q1 = (uint32_t)freq_of_addr<PA>(CFG::row_unperm((int)member * CFG::ROWS + rl));
        const uint32_t ang = (4u * q1 * (uint32_t)bb - (uint32_t)bb) & (2u * N - 1u);
        double sn, cs;
        sincospi((double)ang / (double)N, &sn, &cs);
        tw3[e] = make_double2(cs, sn);
    }
    const double2* my_e1 = e1 + tauA;                               // + TA m
    const double2* my_tw2 = tw2 + (size_t)blA * P1 + tauA;          // + TA rho
    const double2* my_tw3 = tw3 + (size_t)rlB * P2 + tauB;          // + TB m

    unsigned char* ws = ca.workspace + (size_t)cluster * CFG::WS_BYTES;
    const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)CFG::WS_T, 0x00020000);
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + CFG::WS_T, 0, (int)CFG::WS_ACC, 0x00020000);
    // byte offsets of this thread inside the cluster's matrices; the rest of every address is a compile-time constant
    const uint32_t voff_t1 = (uint32_t)(((pA * P1 + tauA) * PITCH + b) * 16);              // phase 1 store: + (it K1 P1 + 16 rho) PITCH
    const uint32_t voff_t2 = (uint32_t)((((hB * UH) * P1 + rowp) * PITCH + tauB) * 16);    // phase 2: + (uu P1 PITCH + TB m)
    const uint32_t voff_t3 = (uint32_t)((((pA * UH) * P1 + tauA) * PITCH + b) * 16);       // phase 3 load: + 16 rho PITCH
    const uint32_t voff_pub = (uint32_t)(((pA * P2 + b) * (2 * P1) + tauA) * 8);           // publish: + (h P1 + TA m)
    uint32_t voff_key[UH];       // GGSW rows of this half's digit polynomials: + (col P + rho TB)
#pragma unroll
    for (int uu = 0; uu < UH; uu++) {
        const int u = hB * UH + uu, it = u / K1, row = u % K1;
        voff_key[uu] = (uint32_t)(((((L - 1 - it) * K1 + row) * K1) * P + rowA * P2 + tauB) * 16);
    }
    uint32_t* flags = &ctl->flags[cluster][0];
    uint32_t epoch = 0;
#ifdef FHESTR_TEST_HOOKS
    const uint32_t mute_epoch = (ca.test_fault && cluster == 0 && member == 1) ? ca.test_fault : 0u;
#else
    const uint32_t mute_epoch = 0u;
#endif

    const uint32_t n = args.n;
    const uint32_t bL = args.base_log * L;
    constexpr size_t GGSW_BYTES = (size_t)L * K1 * K1 * P * 16;

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
                    const uint64_t o = own[2 * m + h];
                    u32x2_t w; w.x = (uint32_t)o; w.y = (uint32_t)(o >> 32);
                    __builtin_amdgcn_raw_buffer_store_b64(w, a_rsrc, (int)voff_pub, (h * P1 + TA * m) * 8, 0);
                }
            }
        }
        cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);

        FHE_STAMP_DECL;
        FHE_STAMP(-1);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t d = lds_d[i];
            if (d == 0xFFFFFFFFu) continue;       // a_i == 0 (bootstrap.rs:281): the same for the whole cluster
#ifdef FHESTR_WALL
            // diagnostic build (scripts/wall_spread_cluster.py): when does each cluster reach steps 0, n/8, 2n/8, ...
            // of its first LWE -- do the clusters of one XCD stay in step (and so share the GGSW rows in L2)?
            if (tid == 0 && member == 0 && sample == cluster && i % (n / 8) == 0 && i / (n / 8) < 8)
                g_wall[cluster * 8 + i / (n / 8)] = __builtin_amdgcn_s_memrealtime();
#endif

            // the GGSW rows of this half's first digit polynomial do not depend on the hand-over: request them now
            const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(args.fbsk)) +
#ifdef FHESTR_CL_KEY0      // diagnostic build: every step reads the first GGSW (wrong results; what does the key stream cost?)
                    (size_t)0 * GGSW_BYTES,
#else
                    (size_t)i * GGSW_BYTES,
#endif
                    0, (int)GGSW_BYTES, 0x00020000);
            double2 bv[K1][R];
            auto issue_key = [&](int uu) {
#pragma unroll
                for (int col = 0; col < K1; col++) {
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (int)voff_key[uu], (col * P + rho * TB) * 16, FHESTR_CL_KEY_AUX);
                        __builtin_memcpy(&bv[col][rho], &raw, 16);
                    }
                }
            };
            // ---- phase 1: rotate, subtract, decompose, twist, column transforms, twiddle -> T ----
            {
                using state_t = typename std::conditional<(L >= 3), uint64_t, uint32_t>::type;
                state_t st_lo[R], st_hi[R];
                // coefficient j = e P2 + b with e = h P1 + a; rem = rq P2 + rb: (j - rem) mod N sits in column
                // (b - rb) mod P2 at e' = (e - rq - [b < rb]) mod 2 P1, negated iff that difference wrapped (xor odd)
                const uint32_t rem = d & (N - 1);
                const int32_t oddmask = -(int32_t)((d >> LOGN) & 1);
                const uint32_t rb = rem & (P2 - 1);
                const int32_t shift = (int32_t)tauA - (int32_t)(rem >> LOGP2) - ((uint32_t)b < rb ? 1 : 0);
                const uint32_t colbase8 = ((uint32_t)pA * P2 + (((uint32_t)b - rb) & (P2 - 1))) * (2 * P1) * 8;
#pragma unroll
                for (int m = 0; m < R; m++) {
                    uint64_t ct[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int32_t e = shift + (h * P1 + TA * m);
                        const uint32_t voff = (((uint32_t)e << 3) & ((2u * P1 - 1u) << 3)) | colbase8;
                        const uint64_t v = load_sc1_b64(a_rsrc, voff);
                        const uint64_t msk = (uint64_t)(int64_t)((e >> 31) ^ oddmask);
                        ct[h] = ((v ^ msk) - msk) - own[2 * m + h];
                    }
                    if constexpr (L >= 3) { st_lo[m] = decomp_init_state64(ct[0], bL); st_hi[m] = decomp_init_state64(ct[1], bL); }
                    else { st_lo[m] = decomp_init_state(ct[0], bL); st_hi[m] = decomp_init_state(ct[1], bL); }
                }
#pragma unroll
                for (int it = 0; it < L; it++) {
                    if (it == L - 1) {
                        // The first key rows of phase 2, requested before the phase's LAST transform round.  Round 3 requested them
                        // just before hand-over 1 and let the hand-over wait with s_waitcnt vmcnt(16) "for everything but the 16
                        // youngest loads": that silently assumes the compiler puts no other vector-memory instruction -- a
                        // register spill to scratch is one -- behind them, and a change of register allocation in round 4 broke
                        // exactly that (wrong ciphertexts, caught by tests/test_gpu_cluster.py).  Every hand-over drains the
                        // wave's queue completely now; by then these loads have had a transform round to arrive (same speed).
                        asm volatile("" ::: "memory");
                        issue_key(0);
                        asm volatile("" ::: "memory");
                    }
                    cplx x[R];
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        double zr, zi;
                        if constexpr (L >= 3) {
                            zr = (double)decomp_next_digit64(st_lo[m], args.base_log);
                            zi = (double)decomp_next_digit64(st_hi[m], args.base_log);
                        } else {
                            zr = (double)decomp_next_digit(st_lo[m], args.base_log);
                            zi = (double)decomp_next_digit(st_hi[m], args.base_log);
                        }
                        const double2 w = my_e1[TA * m];
                        x[m].re = zr * w.x - zi * w.y;
                        x[m].im = zr * w.y + zi * w.x;
                    }
                    fft_forward<PA>(x, fca, areA, aimA, tauA);
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const double2 w = my_tw2[TA * rho];
                        double2 v;
                        v.x = x[rho].re * w.x - x[rho].im * w.y;
                        v.y = x[rho].re * w.y + x[rho].im * w.x;
                        u32x4_t raw;
                        __builtin_memcpy(&raw, &v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, t_rsrc, (int)voff_t1, ((it * K1 * P1 + 16 * rho) * PITCH) * 16, 0);
                    }
                }
            }
            FHE_STAMP(1);
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);
            FHE_STAMP(2);

            // ---- phase 2: row transforms, multiply-accumulate with the GGSW, inverse row transforms, in place ----
            {
                cplx outf[K1][R];
                double2 xin[R];
                auto issue_row = [&](int uu) {
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, (int)voff_t2, (uu * P1 * PITCH + TB * m) * 16, FHESTR_CL_XCHG_AUX);
                        __builtin_memcpy(&xin[m], &raw, 16);
                    }
                };
                issue_row(0);
#pragma unroll
                for (int uu = 0; uu < UH; uu++) {
                    cplx x[R];
#pragma unroll
                    for (int m = 0; m < R; m++) { x[m].re = xin[m].x; x[m].im = xin[m].y; }
                    if (uu + 1 < UH) issue_row(uu + 1);
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
                    if (uu + 1 < UH) issue_key(uu + 1);
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
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double2 w = my_tw3[TB * m];
                    double2 v;
                    v.x = mine[m].re * w.x - mine[m].im * w.y;
                    v.y = mine[m].re * w.y + mine[m].im * w.x;
                    u32x4_t raw;
                    __builtin_memcpy(&raw, &v, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, t_rsrc, (int)voff_t2, (TB * m) * 16, 0);
                }
            }
            FHE_STAMP(3);
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);
            FHE_STAMP(4);

            // ---- phase 3: inverse column transforms, untwist, torus rounding, accumulate, publish ----
            {
                cplx x[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, (int)voff_t3, (16 * rho * PITCH) * 16, FHESTR_CL_XCHG_AUX);
                    double2 v;
                    __builtin_memcpy(&v, &raw, 16);
                    x[rho].re = v.x; x[rho].im = v.y;
                }
                fft_inverse<PA>(x, fca, areA, aimA, tauA);
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double2 w = my_e1[TA * m];
                    const double tre = x[m].re * w.x + x[m].im * w.y;      // * conj(E1[a])
                    const double tim = x[m].im * w.x - x[m].re * w.y;
                    own[2 * m] += from_torus(tre);
                    own[2 * m + 1] += from_torus(tim);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint64_t o = own[2 * m + h];
                        u32x2_t w2; w2.x = (uint32_t)o; w2.y = (uint32_t)(o >> 32);
                        __builtin_amdgcn_raw_buffer_store_b64(w2, a_rsrc, (int)voff_pub, (h * P1 + TA * m) * 8, 0);
                    }
                }
            }
            FHE_STAMP(5);
            // hand-over 3 guards the NEXT step's rotation gather only: this wave's four columns come from at most two members
            // (column b of acc X^d is column b - rb of acc), so it waits for those and no others (cluster_sync, `need`).
            // The T stores of the next phase 1 touch this workgroup's own columns only, which no other member reads in phase 3,
            // and the last step of an LWE needs nobody: the next LWE starts with a full hand-over.
            uint32_t need3 = 0;
            {
                uint32_t j = i + 1;
                while (j < n && lds_d[j] == 0xFFFFFFFFu) j++;
                if (j < n) {
                    const uint32_t rbn = lds_d[j] & (P2 - 1);
                    const uint32_t mine = 1u << ((((uint32_t)b - rbn) & (P2 - 1)) / (uint32_t)CFG::COLS);
                    need3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine) | (uint32_t)__builtin_amdgcn_readlane((int)mine, 63);
                }
            }
            cluster_sync<C>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch, need3);
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
