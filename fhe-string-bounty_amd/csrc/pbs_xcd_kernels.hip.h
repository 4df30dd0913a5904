// pbs_xcd_kernels.hip.h -- blind rotation for N = 32768 (PARAM_MESSAGE_4_CARRY_4_KS_PBS, shortint/parameters/mod.rs:1063-1077)
// with ALL compute units of one XCD per LWE and two LWEs in flight per XCD.
//
// Same algorithm, Fourier key and workspace matrices as pbs_cluster_kernels.hip.h (fft64/crypto/bootstrap.rs:242-364,
// ggsw.rs:477-598); what changes is who does what.  Round 3's cluster kernel puts 8 CUs on an LWE and four LWEs on an
// XCD: their exchange matrices (1.6 MB each) overflow the XCD's 4 MB L2, every exchange read goes to the fabric
// (4.66 MB per LWE-step, 5.4 TB/s) and every hand-over is an exposed memory round trip.  Here
//   * a cluster is 32 workgroups of 256 threads -- every CU of the XCD hosts one workgroup of EACH of the XCD's two
//     clusters (256 registers x 4 waves and < 80 KB of LDS per workgroup: two fit), so while one LWE waits for a
//     hand-over the CU works on the other one, and only 2 x 1.6 MB of exchange matrices are live per XCD;
//   * the 12 size-128 sub-transform rounds a workgroup owes per CMUX step (4 column + 4 row forward, 2 row + 2 column
//     inverse, one round = one wavefront = four transforms of 16 threads) are dealt evenly over its four waves, three
//     each, so the four SIMDs of a CU carry the same load whichever cluster a wave belongs to:
//         wave 2j+1 ("owner" of accumulator polynomial j: its 4 columns x 16 coefficients per thread live in VGPRs)
//             phase 1  gather the rotated accumulator, subtract, decompose BOTH levels; level "first" digits -> own
//                      forward column transforms, the other level's digits -> LDS -> wave 2j
//             phase 2  forward transform of T row 2j+1 (four digit polynomials side by side), multiply by the GGSW
//                      row, reduce over the digit polynomials inside the wavefront, hand the row's two sums to wave 2j
//             phase 3  inverse column transforms, torus rounding, accumulate, publish
//         wave 2j
//             phase 1  forward column transforms of the second level's digits
//             phase 2  forward transform of T row 2j, multiply, reduce; then the inverse row transforms of rows 2j
//                      (own sums) and 2j+1 (from LDS), conjugate twiddle, store in place
//             phase 3  nothing (the partner cluster's waves have the SIMD)
//     The critical path of a step is 4 transform rounds + 3 hand-overs instead of 6 + 3.
// Hand-over, cluster formation, bounded spins and error reporting are pbs_cluster_kernels.hip.h's (cluster_sync with
// four waves, 32 epoch flags in one 128-byte line).  Inside a workgroup the two LDS hand-offs use s_barrier.
#pragma once
#include "pbs_cluster_kernels.hip.h"

// Cache policy of this kernel's Fourier-key loads: nt (2) -- streamed, not retained by the XCD's L2.  The 2 MB GGSW of a step is
// read once per cluster; with the default policy (0) the two clusters' key streams (4 MB per step pair) push the exchange
// matrices (2 x 1.6 MB) out of the 4 MB L2: L2 hit rate 0.64, 5.2 MB of fabric traffic per LWE-step, 18.5 ms per 16 LWEs; with
// nt the exchange reads hit (0.78: the misses left are the key itself), 3.7 MB per LWE-step, 17.4 ms
// (profiles/r04_xcd_history.txt).  The 8-CU cluster kernel keeps the default policy: its four clusters per XCD run in step and
// share the key through L2 (nt there: 1,054 instead of 1,113 PBS/s, profiles/r03_cluster_history.txt).
#ifndef FHESTR_XCD_KEY_AUX
#define FHESTR_XCD_KEY_AUX 2
#endif

namespace fhe {

template <int LOGN, int K1, int L>
struct BrXcdCfg {
    static_assert(K1 == 2 && L == 2, "xcd kernel: k = 1, two decomposition levels");
    using LC = BrLargeCfg<LOGN, K1, L>;
    using PA = typename LC::PA;
    using PB = typename LC::PB;
    static constexpr int N = LC::N, P = LC::P, P1 = LC::P1, P2 = LC::P2, LOGP1 = LC::LOGP1, LOGP2 = LC::LOGP2;
    static constexpr int R = 8, TA = LC::TA, TB = LC::TB;
    static_assert(TA == 16 && TB == 16 && P1 == P2, "xcd kernel: 128 x 128 four-step transform");
    static constexpr int THREADS = 256, WAVES = 4;
    static constexpr int GROUPS = THREADS / TA;            // 16-thread transform groups per workgroup, four per wave
    static constexpr int U = L * K1, UH = U / 2;           // digit polynomials
    static constexpr int COLS = GROUPS / U;                // accumulator columns (per polynomial) a workgroup owns: 4
    static constexpr int C = P2 / COLS;                    // workgroups per LWE: 32 = the CUs of an XCD
    static constexpr int ROWS = P1 / C;                    // T rows a workgroup owns: 4, one per wave
    static_assert(ROWS == WAVES && COLS == 4 && C <= CLUSTER_MAX_MEMBERS, "one row per wave, one 64-byte sector per store row");
    static constexpr int PITCH = P2 + 8;                   // as BrClusterCfg: rows start on different L2 channels
    static constexpr size_t WS_T = (size_t)U * P1 * PITCH * 16;
    static constexpr size_t WS_ACC = (size_t)K1 * N * 8;
    static constexpr size_t WS_BYTES = WS_T + WS_ACC;      // per cluster
    static constexpr int SLOTS = P1 + 16;                  // neighbouring transforms 16 slots apart mod 32
    static constexpr int IM = GROUPS * SLOTS + 2;          // not a multiple of 64 slots (ds_read2st64_b64)
    static constexpr size_t LDS_PLANES = (size_t)8 * 2 * IM;
    static constexpr size_t LDS_E1 = (size_t)P1 * 16;
    static constexpr size_t LDS_TW2 = (size_t)COLS * P1 * 16;
    static constexpr size_t LDS_TW3 = (size_t)ROWS * P2 * 16;
    static constexpr size_t LDS_TWA = (size_t)FftTwiddleTable<PA>::ENTRIES * 16, LDS_TWB = (size_t)FftTwiddleTable<PB>::ENTRIES * 16;
    static constexpr size_t LDS_BYTES = LDS_PLANES + LDS_E1 + LDS_TW2 + LDS_TW3 + LDS_TWA + LDS_TWB;   // + 4 n: modulus-switched mask
    static constexpr size_t LDS_TWO_PER_CU = 80 * 1024;    // half of a CU's LDS
    static_assert(LDS_BYTES + 4 * 1280 <= LDS_TWO_PER_CU, "two workgroups per CU");
    __host__ __device__ static constexpr int row_perm(int A) { return ((A & 7) << (LOGP1 - 3)) | (A >> 3); }
    __host__ __device__ static constexpr int row_unperm(int Ap) { return ((Ap & ((P1 >> 3) - 1)) << 3) | (Ap >> (LOGP1 - 3)); }
};

// LDS hand-off between the waves of a workgroup: this wave's LDS operations are done, then s_barrier.  (__syncthreads()
// would also drain the wave's vector-memory queue, where the key rows of the next phase are deliberately in flight.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int LOGN, int K1, int L>
__global__ void __launch_bounds__(256, 2)
blind_rotate_xcd_kernel(BlindRotateClusterArgs ca) {
    using CFG = BrXcdCfg<LOGN, K1, L>;
    using PA = typename CFG::PA;
    using PB = typename CFG::PB;
    constexpr int N = CFG::N, P = CFG::P, P1 = CFG::P1, P2 = CFG::P2, R = CFG::R, NT = CFG::THREADS;
    constexpr int C = CFG::C, TA = CFG::TA, TB = CFG::TB, LOGP2 = CFG::LOGP2, PITCH = CFG::PITCH, UH = CFG::UH;
    constexpr int SLOTS = CFG::SLOTS, IM = CFG::IM;
    constexpr uint32_t WAVES = CFG::WAVES;
    const BlindRotateArgs& args = ca.base;
    ClusterCtl* ctl = ca.ctl;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    __shared__ uint32_t s_form[4];
    __shared__ uint32_t s_sync[4];        // [0] dead flag, [1], [2] arrival counters of the hand-overs (even / odd epochs)
    uint32_t& s_dead = s_sync[0];

    const int tid = threadIdx.x;
    if (tid == 0) cluster_join_per_cu<C>(ctl, ca.status, ca.spin_limit, s_form, s_sync);
    __syncthreads();
    const uint32_t cluster = s_form[0], member = s_form[1], n_clusters = s_form[2];
    if (cluster == 0xFFFFFFFFu) return;          // not part of a complete cluster: the whole workgroup leaves

    // ---- thread roles (wave-uniform: wave, owner, pair) ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, tau = lane & 15;                 // transform group inside the wave, thread inside the group
    const bool owner = (wave & 1) != 0;
    const int pair = wave >> 1;
    // phases 1 and 3: the pair's waves hold column b = member COLS + g of polynomial p = pair; the owner transforms the
    // digits the decomposition yields first (it = 0), its partner the second level's
    const int pA = pair, itA = owner ? 0 : 1, b = (int)member * CFG::COLS + g;
    // phase 2, forward: wave = T row, group = digit polynomial u = it K1 + p
    const int rowpF = (int)member * CFG::ROWS + wave, rowAF = CFG::row_unperm(rowpF);
    const int uF = g, itF = uF / K1, polyF = uF % K1;
    // phase 2, inverse: groups 0, 1 finish the wave's row (output columns 0, 1)
    const int rlI = wave, colI = g & 1, rowpI = rowpF;

    double* are = lds + (size_t)(tid / TA) * SLOTS;            // this group's exchange planes (both transform sizes are 128)
    double* aim = are + IM;
    // level-two digits, owner -> partner: the partner wave's (idle) real planes, [m][lane] int2
    int2* dig = reinterpret_cast<int2*>(lds + (size_t)((wave & ~1) * 4) * SLOTS);
    static_assert((size_t)4 * SLOTS * 8 >= (size_t)R * 64 * 8, "digit hand-off fits the partner's planes");

    double2* e1 = reinterpret_cast<double2*>(smem + CFG::LDS_PLANES);
    double2* tw2 = e1 + P1;
    double2* tw3 = tw2 + CFG::COLS * P1;
    double2* twa = tw3 + CFG::ROWS * P2;
    double2* twb = twa + FftTwiddleTable<PA>::ENTRIES;
    uint32_t* lds_d = reinterpret_cast<uint32_t*>(smem + CFG::LDS_BYTES);     // [n] modulus-switched mask
    FftTwiddleTable<PA>::fill(twa, tid, NT);
    FftTwiddleTable<PB>::fill(twb, tid, NT);
    const FftTwiddleTable<PA> fca{twa, tau};
    const FftTwiddleTable<PB> fcb{twb, tau};
    // constant factors as in blind_rotate_cluster_kernel: twist row part E1[a], the column part folded into the
    // four-step twiddles tw2[bl][rho][tau] (after the forward column transform) and tw3[rl][m][tau] (after the inverse
    // row transform); angles as integers mod 2N so that sincospi sees an exact argument
    for (int e = tid; e < P1; e += NT) {
        double sn, cs;
        sincospi((double)((uint32_t)e * P2) / (double)N, &sn, &cs);
        e1[e] = make_double2(cs, sn);
    }
    for (int e = tid; e < CFG::COLS * P1; e += NT) {
        const int bl = e / P1, rho = (e / TA) % R, t = e % TA;
        const uint32_t bb = member * CFG::COLS + bl;
        const uint32_t q1 = (uint32_t)freq_of_addr<PA>(slot_addr<PA>(t, rho));
        const uint32_t ang = (bb - 4u * q1 * bb) & (2u * N - 1u);
        double sn, cs;
        sincospi((double)ang / (double)N, &sn, &cs);
        tw2[e] = make_double2(cs, sn);
    }
    for (int e = tid; e < CFG::ROWS * P2; e += NT) {
        const int rl = e / P2, bb = (e % P2) / TB * TB + e % TB;      // [rl][m][tau] with bb = tau + TB m
        const uint32_t q1 = (uint32_t)freq_of_addr<PA>(CFG::row_unperm((int)member * CFG::ROWS + rl));
        const uint32_t ang = (4u * q1 * (uint32_t)bb - (uint32_t)bb) & (2u * N - 1u);
        double sn, cs;
        sincospi((double)ang / (double)N, &sn, &cs);
        tw3[e] = make_double2(cs, sn);
    }
    const double2* my_e1 = e1 + tau;                               // + TA m
    const double2* my_tw2 = tw2 + (size_t)g * P1 + tau;            // + TA rho
    const double2* my_tw3 = tw3 + (size_t)rlI * P2 + tau;          // + TB m

    unsigned char* ws = ca.workspace + (size_t)cluster * CFG::WS_BYTES;
    const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)CFG::WS_T, 0x00020000);
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + CFG::WS_T, 0, (int)CFG::WS_ACC, 0x00020000);
    // byte offsets of this thread inside the cluster's matrices; the rest of every address is a compile-time constant
    const uint32_t voff_t1 = (uint32_t)((((itA * K1 + pA) * P1 + tau) * PITCH + b) * 16);        // phase 1 store: + 16 rho PITCH
    const uint32_t voff_t2f = (uint32_t)(((uF * P1 + rowpF) * PITCH + tau) * 16);                // phase 2 load: + TB m
    const uint32_t voff_t2i = (uint32_t)((((colI * UH) * P1 + rowpI) * PITCH + tau) * 16);       // phase 2 store: + TB m
    const uint32_t voff_t3 = (uint32_t)((((pA * UH) * P1 + tau) * PITCH + b) * 16);              // phase 3 load: + 16 rho PITCH
    const uint32_t voff_pub = (uint32_t)(((pA * P2 + b) * (2 * P1) + tau) * 8);                  // publish: + (h P1 + TA m)
    const uint32_t voff_key = (uint32_t)(((((L - 1 - itF) * K1 + polyF) * K1) * P + rowAF * P2 + tau) * 16);   // + (col P + rho TB)
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

        // acc <- LUT * X^{-ms(body)}: an owner thread's 2R coefficients j = h P + (tau + TA m) P2 + b
        uint64_t own[2 * R];
#pragma unroll
        for (int q = 0; q < 2 * R; q++) own[q] = 0;
        if (owner) {
            const uint32_t d = modulus_switch(lwe[n], LOGN);
            const uint32_t rem = d & (N - 1);
            const bool odd = (d >> LOGN) & 1;
#pragma unroll
            for (int m = 0; m < R; m++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t j = (uint32_t)h * P + (uint32_t)(tau + TA * m) * P2 + (uint32_t)b;
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
        cluster_sync<C, WAVES>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);

        FHE_STAMP_DECL;
        FHE_STAMP(-1);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t d = lds_d[i];
            if (d == 0xFFFFFFFFu) continue;       // a_i == 0 (bootstrap.rs:281): the same for the whole cluster

            // The GGSW rows this group multiplies by in phase 2 (16 x 16 bytes per thread: 2 MB per step and cluster, from the
            // Infinity Cache or HBM) are requested inside phase 1 -- by the partner waves at once, by the owners as soon as their
            // gather has been consumed (vector-memory loads return in order: nothing the phase waits for may queue behind them).
            // Requested just before hand-over 1 instead, they held every wave's flag polls back for 4 us; requested by the partner
            // waves in the PREVIOUS step's phase 3 (where they idle) they sit in front of the owners' T' loads -- a CU serves its
            // waves' vector-memory requests in order: 13.4 instead of 11.0 ms per 8 LWEs (profiles/r04_xcd_history.txt).
            const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(args.fbsk)) + (size_t)i * GGSW_BYTES, 0, (int)GGSW_BYTES,
                0x00020000);
            double2 bv[K1][R];
            auto issue_key = [&]() {
#pragma unroll
                for (int col = 0; col < K1; col++) {
#pragma unroll
                    for (int rho = 0; rho < R; rho++) {
                        const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (int)voff_key, (col * P + rho * TB) * 16, FHESTR_XCD_KEY_AUX);
                        __builtin_memcpy(&bv[col][rho], &raw, 16);
                    }
                }
            };

            // ---- phase 1: rotate, subtract, decompose (owner), twist, column transforms, twiddle -> T ----
            {
                int32_t zr[R], zi[R];             // this wave's digits of (column b, rows tau + TA m), halves lo / hi
                if (owner) {
                    // coefficient j = e P2 + b with e = h P1 + a; rem = rq P2 + rb: (j - rem) mod N sits in column
                    // (b - rb) mod P2 at e' = (e - rq - [b < rb]) mod 2 P1, negated iff that difference wrapped (xor odd)
                    const uint32_t rem = d & (N - 1);
                    const int32_t oddmask = -(int32_t)((d >> LOGN) & 1);
                    const uint32_t rb = rem & (P2 - 1);
                    const int32_t shift = (int32_t)tau - (int32_t)(rem >> LOGP2) - ((uint32_t)b < rb ? 1 : 0);
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
                        uint32_t st_lo = decomp_init_state(ct[0], bL), st_hi = decomp_init_state(ct[1], bL);
                        zr[m] = decomp_next_digit(st_lo, args.base_log);
                        zi[m] = decomp_next_digit(st_hi, args.base_log);
                        int2 second;
                        second.x = decomp_next_digit(st_lo, args.base_log);
                        second.y = decomp_next_digit(st_hi, args.base_log);
                        dig[m * 64 + lane] = second;
                    }
                }
                // key rows: the owners' gather has been consumed, the partners have waited so far.  (Measured, profiles/r04_xcd_history.txt:
                // requested just before hand-over 1 they hold every wave's flag polls back for 4 us -- vector-memory loads return in
                // order; requested by the partners only after the digits arrived they sit in front of BOTH waves' T stores: slower still.)
                asm volatile("" ::: "memory");
                issue_key();
                asm volatile("" ::: "memory");
                lds_barrier();                    // the second level's digits are in LDS
                if (!owner) {
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const int2 second = dig[m * 64 + lane];
                        zr[m] = second.x;
                        zi[m] = second.y;
                    }
                    wave_local_fence();           // the reads above are issued before this wave's transform reuses the planes
                }
                FHE_STAMP(0);
                cplx x[R];
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double2 w = my_e1[TA * m];
                    const double fr = (double)zr[m], fi = (double)zi[m];
                    x[m].re = fr * w.x - fi * w.y;
                    x[m].im = fr * w.y + fi * w.x;
                }
                fft_forward<PA>(x, fca, are, aim, tau);
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 w = my_tw2[TA * rho];
                    double2 v;
                    v.x = x[rho].re * w.x - x[rho].im * w.y;
                    v.y = x[rho].re * w.y + x[rho].im * w.x;
                    u32x4_t raw;
                    __builtin_memcpy(&raw, &v, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, t_rsrc, (int)voff_t1, (16 * rho * PITCH) * 16, 0);
                }
            }
            FHE_STAMP(1);
            cluster_sync<C, WAVES>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);
            FHE_STAMP(2);

            // ---- phase 2: row transforms, multiply by the GGSW row, reduce, inverse row transforms, in place ----
            {
                cplx x[R];
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, (int)voff_t2f, (TB * m) * 16, FHESTR_CL_XCHG_AUX);
                    double2 v;
                    __builtin_memcpy(&v, &raw, 16);
                    x[m].re = v.x; x[m].im = v.y;
                }
                fft_forward<PB>(x, fcb, are, aim, tau);
                FHE_STAMP(3);
                // products with the row of output column 0 and 1, then the sum over the wave's four digit polynomials:
                // v_permlane16_swap pairs u with u ^ 1 (the even group keeps column 0, the odd one column 1),
                // v_permlane32_swap adds the other pair's sum -- groups 0 / 2 end with column 0, groups 1 / 3 with column 1
                cplx sum[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const double2 k0 = bv[0][rho], k1 = bv[1][rho];
                    double a_re = k0.x * x[rho].re - k0.y * x[rho].im, a_im = k0.x * x[rho].im + k0.y * x[rho].re;
                    double b_re = k1.x * x[rho].re - k1.y * x[rho].im, b_im = k1.x * x[rho].im + k1.y * x[rho].re;
                    swap_halves<true>(a_re, b_re);
                    swap_halves<true>(a_im, b_im);
                    double s_re = a_re + b_re, s_im = a_im + b_im;
                    double t_re = s_re, t_im = s_im;
                    swap_halves<false>(s_re, t_re);
                    swap_halves<false>(s_im, t_im);
                    sum[rho].re = s_re + t_re;
                    sum[rho].im = s_im + t_im;
                }
                FHE_STAMP(7);
                // every wave finishes its own row: groups 0 / 1 hold the sums of columns 0 / 1 (groups 2 / 3 the same values:
                // they run along and store nothing).  Handing the odd rows to the even waves through LDS instead (all four
                // groups of an even wave busy, the odd wave idle) saves a quarter of the wave-instructions of a step but
                // puts an LDS hand-off, a barrier and the partner's transform on the step's critical path.
                fft_inverse<PB>(sum, fcb, are, aim, tau);
                if (lane < 32) {
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const double2 w = my_tw3[TB * m];
                        double2 v;
                        v.x = sum[m].re * w.x - sum[m].im * w.y;
                        v.y = sum[m].re * w.y + sum[m].im * w.x;
                        u32x4_t raw;
                        __builtin_memcpy(&raw, &v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(raw, t_rsrc, (int)voff_t2i, (TB * m) * 16, 0);
                    }
                }
            }
            FHE_STAMP(8);
            cluster_sync<C, WAVES>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch);
            FHE_STAMP(4);

            // ---- phase 3 (owner waves): inverse column transforms, untwist, torus rounding, accumulate, publish ----
            if (owner) {
                cplx x[R];
#pragma unroll
                for (int rho = 0; rho < R; rho++) {
                    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(t_rsrc, (int)voff_t3, (16 * rho * PITCH) * 16, FHESTR_CL_XCHG_AUX);
                    double2 v;
                    __builtin_memcpy(&v, &raw, 16);
                    x[rho].re = v.x; x[rho].im = v.y;
                }
                fft_inverse<PA>(x, fca, are, aim, tau);
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
            // hand-over 3 guards the next step's rotation gather only: an owner wave waits for the (at most two) members that own
            // its source columns, a partner wave for nobody (it meets its owner at the LDS barrier of phase 1); the last step of
            // an LWE needs nobody (pbs_cluster_kernels.hip.h: cluster_sync, `need`)
            uint32_t need3 = 0;
            if (owner) {
                uint32_t j = i + 1;
                while (j < n && lds_d[j] == 0xFFFFFFFFu) j++;
                if (j < n) {
                    const uint32_t rbn = lds_d[j] & (P2 - 1);
                    const uint32_t mine = 1u << ((((uint32_t)b - rbn) & (P2 - 1)) / (uint32_t)CFG::COLS);
                    need3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine) | (uint32_t)__builtin_amdgcn_readlane((int)mine, 63);
                }
            }
            cluster_sync<C, WAVES>(flags, member, epoch, &s_dead, ctl, ca.status, ca.spin_limit, mute_epoch, need3);
            FHE_STAMP(6);
        }
#ifdef FHESTR_STAMPS
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && sample == cluster) {
            uint32_t hw, xcc;       // where this workgroup runs: slot 9 = cluster | member | XCC | HW_ID (which CU hosts which clusters)
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            stamp_acc[9] = ((unsigned long long)cluster << 48) | ((unsigned long long)member << 40) | ((unsigned long long)(xcc & 7u) << 32) | hw;
            for (int sg = 0; sg < STAMP_SEGS; sg++)
                g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * STAMP_SEGS + sg] = stamp_acc[sg];
        }
#endif

        // sample extraction at degree 0 (glwe_sample_extraction.rs:121-146), straight from the owners' registers
        if (owner) {
            uint64_t* out = args.lwe_out + (size_t)sample * ((size_t)(K1 - 1) * N + 1);
#pragma unroll
            for (int m = 0; m < R; m++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t j = (uint32_t)h * P + (uint32_t)(tau + TA * m) * P2 + (uint32_t)b;
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
}

}  // namespace fhe
