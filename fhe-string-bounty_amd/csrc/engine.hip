// engine.hip -- host side of libfhestr.so: device memory, key residency, kernel dispatch.
//
// The reference keeps a ServerKey {key_switching_key, bootstrapping_key (Fourier)} in host memory
// and runs one ciphertext at a time through thread-local scratch
// (shortint/server_key/mod.rs:783-857, shortint/engine/mod.rs:184-234).  Here the keys are made
// resident in HBM once (BSK 48.6 MB + KSK 60.9 MB for PARAM_MESSAGE_2_CARRY_2: both sit in the
// 256 MB Infinity Cache) and whole batches of LWEs go through three launches on one HIP stream:
// memset -> keyswitch (ks_decompose + keyswitch_mfma_kernel) -> blind rotation.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "keygen_kernels.hip.h"   // first: asserts fp contract(off); the FFT header turns fusion on after it
#include "lwe_kernels.hip.h"
#include "ks_mfma_kernels.hip.h"
#include "pbs_kernels.hip.h"
#include "pbs_dense_kernels.hip.h"
#include "pbs_large_kernels.hip.h"
#include "pbs_cluster_kernels.hip.h"
#include "pbs_xcd_kernels.hip.h"
#include "pbs_multibit_kernels.hip.h"
#include "pbs_seq_kernels.hip.h"
#include "seeded_kernels.hip.h"

namespace fhe {

thread_local std::string g_last_error;

int fail(const std::string& msg) {
    g_last_error = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

// ---- blind-rotation variant registry ----------------------------------------------------------
struct BrVariant {
    int logN, k1, L, logR;
    bool wide;          // true: every thread carries all k+1 polynomials (blind_rotate_wide_kernel)
    bool large;         // true: four-step FFT through an HBM workspace (blind_rotate_large_kernel)
    int grouping = 1;   // > 1: multi-bit PBS kernel for that grouping factor
    int lds_per_n = 4;  // dynamic LDS bytes per small-LWE coefficient (modulus-switched mask / degrees)
    size_t ws_bytes;    // per-LWE workspace (large only)
    size_t convert_ws;  // per-workgroup workspace of the conversion kernel (large only)
    int threads;
    int convert_threads;
    size_t lds_bytes;
    size_t convert_lds;
    const void* rotate_fn;
    const void* convert_fn;
    bool convert_one_per_block = false;   // the conversion kernel takes one polynomial per workgroup (K1 otherwise)
    const void* rotate_keypf_fn = nullptr; // wide layout with the whole key of a step prefetched (single launches only, see BrWideCfg)
    bool own_plan = false;                // wide layout on a plan of its own: when it is not the engine's primary variant it reads the
                                          // dense kernel's copy of the key (same plan), not the primary variant's
    // multi-bit, small batches: build every (LWE, group) GGSW on the whole GPU first, then rotate against them
    const void* combine_fn = nullptr;
    const void* rotate_combined_fn = nullptr;
    size_t combine_lds = 0;
    int combine_grid_y = 1;
    int combine_chunk = 1;
    size_t combined_bytes = 0;    // one combined GGSW
    // multi-bit on every other shape: two-kernel path only (generic combine + the classic kernel's EXTPROD mode)
    const void* extprod_fn = nullptr;
    const void* combine_generic_fn = nullptr;
    // N >= 16384: several compute units of one XCD per LWE (pbs_cluster_kernels.hip.h); same Fourier key as rotate_fn
    const void* cluster_fn = nullptr;
    int cluster_size = 0;         // workgroups per LWE
    size_t cluster_ws = 0;        // workspace bytes per cluster
    size_t cluster_lds = 0;
    // N = 32768, two levels: all CUs of an XCD per LWE, two LWEs in flight per XCD (pbs_xcd_kernels.hip.h)
    const void* xcd_fn = nullptr;
    int xcd_size = 0, xcd_threads = 0;
    size_t xcd_ws = 0, xcd_lds = 0, xcd_lds_one_per_cu = 0;
    // dense layout (pbs_dense_kernels.hip.h): four workgroups per CU, for batches beyond two LWEs per CU
    const void* dense_fn = nullptr;
    size_t dense_lds = 0;
    const void* dense_convert_fn = nullptr;     // its Fourier key is in its own plan's order: a second copy of the key
    int dense_convert_threads = 0;
};

template <int LOGN, int LOGR, int K1, int L>
BrVariant make_variant() {
    using CFG = BrCfg<LOGN, LOGR, K1, L>;
    BrVariant v;
    v.logN = LOGN; v.k1 = K1; v.L = L; v.logR = LOGR; v.wide = false; v.large = false;
    v.ws_bytes = 0; v.convert_ws = 0;
    v.threads = CFG::THREADS;
    v.convert_threads = CFG::THREADS;
    v.lds_bytes = CFG::LDS_FIXED;   // + 4*n for the modulus-switched mask
    v.convert_lds = (size_t)K1 * CFG::GROUP_SLOTS * 8;
    v.rotate_fn = reinterpret_cast<const void*>(&blind_rotate_kernel<LOGN, LOGR, K1, L>);
    v.convert_fn = reinterpret_cast<const void*>(&bsk_convert_kernel<LOGN, LOGR, K1, L>);
    return v;
}

template <int LOGN, int LOGR, int K1, int L>
BrVariant make_wide_variant() {
    using CFG = BrWideCfg<LOGN, LOGR, K1, L>;
    BrVariant v = make_variant<LOGN, LOGR, K1, L>();   // same Fourier key layout + conversion kernel
    v.wide = true;
    v.threads = CFG::THREADS;
    v.lds_bytes = CFG::LDS_FIXED;
    v.rotate_fn = reinterpret_cast<const void*>(&blind_rotate_wide_kernel<LOGN, LOGR, K1, L>);
    if constexpr (CFG::OWN_PLAN) {        // FftSwap11 / FftSwap9: the key in that plan's order, one polynomial per workgroup
        v.convert_fn = reinterpret_cast<const void*>(&bsk_convert_wide_kernel<LOGN, LOGR, K1, L>);
        v.convert_threads = CFG::THREADS;
        v.convert_lds = (size_t)CFG::GROUP_SLOTS * 8;
        v.convert_one_per_block = true;
        v.own_plan = true;
    }
    if constexpr (LOGN == 10 && LOGR == 2 && K1 == 3 && L == 1)         // N = 1024, k = 2: single launches of 257 ... 512 LWEs
        v.rotate_keypf_fn = reinterpret_cast<const void*>(&blind_rotate_wide_kernel<LOGN, LOGR, K1, L, true>);
    if constexpr (LOGN == 10 && LOGR == 2 && K1 == 3 && L == 1) {       // N = 1024, k = 2 (pbs_dense_kernels.hip.h)
        using DC = BrDenseCfg<LOGN, K1>;
        static_assert(DC::THREADS == CFG::THREADS, "same launch shape as the wide kernel");
        v.dense_fn = reinterpret_cast<const void*>(&blind_rotate_dense_kernel<LOGN, K1>);
        v.dense_lds = DC::LDS_FIXED;
        v.dense_convert_fn = reinterpret_cast<const void*>(&bsk_convert_dense_kernel<LOGN, K1>);
        v.dense_convert_threads = DC::THREADS;
    }
    return v;
}

template <int LOGN, int K1, int L>
BrVariant make_large_variant() {
    using CFG = BrLargeCfg<LOGN, K1, L>;
    BrVariant v;
    v.logN = LOGN; v.k1 = K1; v.L = L; v.logR = 3; v.wide = false; v.large = true;
    v.lds_per_n = 0;
    v.threads = CFG::THREADS;
    v.convert_threads = CFG::THREADS;
    v.lds_bytes = CFG::LDS_BYTES;
    v.convert_lds = CFG::LDS_BYTES;
    v.ws_bytes = CFG::WS_BYTES;
    v.convert_ws = (size_t)CFG::P * 16;
    v.rotate_fn = reinterpret_cast<const void*>(&blind_rotate_large_kernel<LOGN, K1, L>);
    v.convert_fn = reinterpret_cast<const void*>(&bsk_convert_large_kernel<LOGN, K1, L>);
    if constexpr (K1 == 2) {
        using CC = BrClusterCfg<LOGN, K1, L>;
        v.cluster_fn = reinterpret_cast<const void*>(&blind_rotate_cluster_kernel<LOGN, K1, L>);
        v.cluster_size = CC::C;
        v.cluster_ws = CC::WS_BYTES;
        v.cluster_lds = CC::LDS_BYTES;
        if constexpr (L == 2 && LOGN == 15) {
            using XC = BrXcdCfg<LOGN, K1, L>;
            v.xcd_fn = reinterpret_cast<const void*>(&blind_rotate_xcd_kernel<LOGN, K1, L>);
            v.xcd_size = XC::C;
            v.xcd_threads = XC::THREADS;
            v.xcd_ws = XC::WS_BYTES;
            v.xcd_lds = XC::LDS_BYTES;
            v.xcd_lds_one_per_cu = XC::LDS_TWO_PER_CU + 1024;     // more than half of a CU's LDS: one workgroup per CU
        }
    }
    return v;
}

// N = 8192: transforms in LDS one polynomial at a time, accumulator in a cache-resident workspace (pbs_seq_kernels.hip.h)
template <int LOGN, int K1, int L>
BrVariant make_seq_variant() {
    using CFG = BrSeqCfg<LOGN, K1, L>;
    BrVariant v;
    v.logN = LOGN; v.k1 = K1; v.L = L; v.logR = 3; v.wide = false; v.large = true;
    v.lds_per_n = 4;
    v.threads = CFG::THREADS;
    v.convert_threads = CFG::THREADS;
    v.lds_bytes = CFG::LDS_FIXED;
    v.convert_lds = CFG::LDS_CONVERT;
    v.ws_bytes = CFG::WS_BYTES;
    v.convert_ws = 16;
    v.rotate_fn = reinterpret_cast<const void*>(&blind_rotate_seq_kernel<LOGN, K1, L>);
    v.convert_fn = reinterpret_cast<const void*>(&bsk_convert_seq_kernel<LOGN, K1, L>);
    return v;
}

template <int LOGN, int LOGR, int K1, int G>
BrVariant make_multibit_variant() {
    using CFG = BrMultiBitCfg<LOGN, LOGR, K1, G>;
    BrVariant v = make_variant<LOGN, LOGR, K1, 1>();   // same Fourier key slot order + conversion kernel
    v.grouping = G;
    v.lds_bytes = CFG::LDS_FIXED;
    v.lds_per_n = 4 * ((1 << G) - 1) / G + 4;         // (n/G) * (2^G - 1) degrees, rounded up
    v.rotate_fn = reinterpret_cast<const void*>(&blind_rotate_multibit_kernel<LOGN, LOGR, K1, G>);
    v.combine_fn = reinterpret_cast<const void*>(&multibit_combine_kernel<LOGN, LOGR, K1, G>);
    v.rotate_combined_fn = reinterpret_cast<const void*>(&blind_rotate_multibit_kernel<LOGN, LOGR, K1, G, true>);
    v.combine_lds = CFG::LDS_ROOTS;
    v.combine_grid_y = CFG::R / CFG::COMBINE_SLOTS;
    v.combine_chunk = CFG::COMBINE_CHUNK;
    v.combined_bytes = (size_t)K1 * K1 * CFG::P * 16;
    return v;
}

// Multi-bit PBS for a shape served by the classic split kernel / the large-N kernel: same Fourier key layout and
// conversion, rotation = EXTPROD mode against GGSWs prepared by multibit_combine_generic_kernel.
template <int LOGN, int LOGR, int K1, int L>
BrVariant make_multibit_generic_variant(int G) {
    BrVariant v = make_variant<LOGN, LOGR, K1, L>();
    v.grouping = G;
    v.extprod_fn = reinterpret_cast<const void*>(&blind_rotate_kernel<LOGN, LOGR, K1, L, true>);
    v.rotate_fn = v.extprod_fn;
    v.combine_generic_fn = G == 2 ? reinterpret_cast<const void*>(&multibit_combine_generic_kernel<2>)
                                  : reinterpret_cast<const void*>(&multibit_combine_generic_kernel<3>);
    v.combined_bytes = (size_t)L * K1 * K1 * (size_t)(1 << (LOGN - 1)) * 16;
    return v;
}

template <int LOGN, int K1, int L>
BrVariant make_multibit_seq_variant(int G) {
    BrVariant v = make_seq_variant<LOGN, K1, L>();
    v.grouping = G;
    v.extprod_fn = reinterpret_cast<const void*>(&blind_rotate_seq_kernel<LOGN, K1, L, true>);
    v.rotate_fn = v.extprod_fn;
    v.combine_generic_fn = G == 2 ? reinterpret_cast<const void*>(&multibit_combine_generic_kernel<2>)
                                  : reinterpret_cast<const void*>(&multibit_combine_generic_kernel<3>);
    v.combined_bytes = (size_t)L * K1 * K1 * (size_t)(1 << (LOGN - 1)) * 16;
    return v;
}

template <int LOGN, int K1, int L>
BrVariant make_multibit_large_variant(int G) {
    BrVariant v = make_large_variant<LOGN, K1, L>();
    v.grouping = G;
    v.extprod_fn = reinterpret_cast<const void*>(&blind_rotate_large_kernel<LOGN, K1, L, true>);
    v.rotate_fn = v.extprod_fn;
    v.combine_generic_fn = G == 2 ? reinterpret_cast<const void*>(&multibit_combine_generic_kernel<2>)
                                  : reinterpret_cast<const void*>(&multibit_combine_generic_kernel<3>);
    v.combined_bytes = (size_t)L * K1 * K1 * (size_t)(1 << (LOGN - 1)) * 16;
    return v;
}

static const std::vector<BrVariant>& variants() {
    static const std::vector<BrVariant> v = {
        // PARAM_MESSAGE_2_CARRY_2_KS_PBS: N=2048, k=1, l=1  (first entry of a shape = default)
        make_variant<11, 2, 2, 1>(), make_variant<11, 3, 2, 1>(), make_variant<11, 4, 2, 1>(),
        make_wide_variant<11, 2, 2, 1>(), make_wide_variant<11, 3, 2, 1>(),
        // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS: multi-bit PBS, grouping factor 2
        make_multibit_variant<11, 2, 2, 2>(),
        // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS: grouping factor 3 (8 GGSWs per group)
        make_multibit_variant<11, 2, 2, 3>(),
        // PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_{2,3}_KS_PBS (N = 512, k = 3) and
        // PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_{2,3}_KS_PBS (N = 8192, two levels): two-kernel path
        make_multibit_generic_variant<9, 2, 4, 1>(2), make_multibit_generic_variant<9, 2, 4, 1>(3),
        make_multibit_seq_variant<13, 2, 2>(2), make_multibit_seq_variant<13, 2, 2>(3),
        // toy shapes of the multi-bit tests (N = 256, k = 1, two levels; N = 128, k = 2)
        make_multibit_generic_variant<8, 2, 2, 2>(2), make_multibit_generic_variant<8, 2, 2, 2>(3),
        make_multibit_generic_variant<7, 2, 3, 1>(2), make_multibit_generic_variant<7, 2, 3, 1>(3),
        // N=1024, k=2, l=1 family (PARAM_MESSAGE_2_CARRY_1_KS_PBS ...)
        // (4 points per thread, 384 threads: 2.77 ms per 256 LWEs; 8 per thread, 192 threads: 3.38 ms)
        make_variant<10, 2, 3, 1>(), make_variant<10, 3, 3, 1>(), make_wide_variant<10, 2, 3, 1>(),
        // PARAM_MESSAGE_1_CARRY_1_KS_PBS: N=512, k=3, l=1
        make_variant<9, 2, 4, 1>(),
        // toy shapes used by the fast tests
        make_variant<8, 2, 2, 2>(), make_variant<7, 2, 3, 1>(),
        make_wide_variant<8, 2, 2, 2>(), make_wide_variant<7, 2, 3, 1>(),
        // the remaining *_KS_PBS shapes of shortint/parameters/mod.rs
        make_variant<8, 2, 6, 1>(),                                  // N = 256, k = 5  (1_CARRY_0)
        make_variant<9, 2, 3, 2>(),                                  // N = 512, k = 2, 2 levels (2_CARRY_0)
        make_wide_variant<12, 2, 2, 1>(), make_wide_variant<12, 2, 2, 2>(),   // N = 4096 (2_CARRY_3 ..., 1_CARRY_4)
        // polynomial sizes beyond the LDS: four-step FFT through an HBM workspace
        make_seq_variant<13, 2, 1>(), make_seq_variant<13, 2, 2>(),           // N = 8192  (5_CARRY_1 ..., 3_CARRY_3 ...)
        make_large_variant<14, 2, 2>(),                                       // N = 16384 (3_CARRY_4 ...)
        make_large_variant<15, 2, 2>(),                                       // N = 32768 (4_CARRY_4 ...)
        make_large_variant<14, 2, 3>(), make_large_variant<15, 2, 3>(),       // 3 levels of base 2^11 (1_CARRY_6, 3_CARRY_5 ...)
    };
    return v;
}

// selector: 0 = default; otherwise log2(points per thread) + 16 if the "wide" layout is wanted
static const BrVariant* find_variant(const fhe_params_t& p, int selector) {
    int logN = 0;
    while ((1u << logN) < p.N) logN++;
    const int logR = selector & 15;
    const bool wide = (selector & 16) != 0;
    const int grouping = p.grouping_factor > 1 ? (int)p.grouping_factor : 1;
    for (const auto& v : variants())
        if (v.logN == logN && v.k1 == (int)p.k + 1 && v.L == (int)p.pbs_level && v.grouping == grouping &&
            (selector == 0 || (v.logR == logR && v.wide == wide)))
            return &v;
    return nullptr;
}

// ---- Engine -----------------------------------------------------------------------------------
// Everything Engine::create checks before it touches a device: shapes the kernels are instantiated for, decomposition
// ranges of the keyswitch paths (any level count: more than 16 levels take the byte-plane kernel, ks_mfma_supported).
int params_supported(const fhe_params_t& p, int selector, const BrVariant** out_v) {
    if ((p.N & (p.N - 1)) || p.N < 128) return fail("polynomial size must be a power of two >= 128");
    if (p.pbs_base_log < 1 || p.pbs_base_log > 31 || p.pbs_base_log * p.pbs_level > (p.pbs_level >= 3 ? 62u : 31u))
        return fail("unsupported PBS decomposition (base_log * level must be <= 31, or <= 62 with >= 3 levels)");
    if (p.ks_level < 1 || p.ks_base_log < 1 || p.ks_base_log > 7 || p.ks_base_log * p.ks_level > 62)
        return fail("unsupported keyswitch decomposition (base_log 1..7, base_log * level <= 62)");
    if (p.msg_mod * p.carry_mod == 0 || (p.N % (p.msg_mod * p.carry_mod)) != 0)
        return fail("msg_mod * carry_mod must divide N");
    const BrVariant* v = find_variant(p, selector);
    if (p.grouping_factor > 1 && (p.n % p.grouping_factor) != 0) return fail("grouping factor must divide n");
    if (!v) return fail("no blind-rotation kernel instantiated for this (N, k, level, grouping factor)");
    if (out_v) *out_v = v;
    return 0;
}
int params_supported(const fhe_params_t& p) { return params_supported(p, 0, nullptr); }

// Every environment switch of the library in one place, read when an engine is created.  They exist for diagnostics and A/B
// measurements (scripts/); the product interface is the API: fhe_engine_set_variant / _set_pipeline / _set_keep_busy /
// _set_cluster_mode / _set_multibit_combine_max.
EngineEnv EngineEnv::read() {
    EngineEnv v;
    auto num = [](const char* name, int& out) { if (const char* t = getenv(name)) out = atoi(t); };
    num("FHESTR_LOG2_POINTS", v.log2_points);
    num("FHESTR_WIDE_FAIR", v.wide_fair);
    num("FHESTR_DENSE_PER_CU", v.dense_per_cu);
    num("FHESTR_CLUSTER_FALLBACK", v.cluster_fallback);
    num("FHESTR_KEEP_BUSY", v.keep_busy);
    num("FHESTR_OVERLAP_STREAMS", v.overlap_streams);
    num("FHESTR_KS_MFMA", v.ks_mfma);
    num("FHESTR_KS_CHUNKS", v.ks_chunks);
    num("FHESTR_CLUSTER", v.cluster_mode);
    num("FHESTR_CLUSTER_SPIN_LIMIT", v.cluster_spin_limit);
    num("FHESTR_MULTIBIT_COMBINE_MAX", v.multibit_combine_max);
    num("FHESTR_CLUSTER_TEST_FAULT", v.cluster_test_fault);
    return v;
}

int Engine::create(const fhe_params_t& p, int device, Engine** out) {
    const int env_logr = EngineEnv::read().log2_points;
    const BrVariant* v = nullptr;
    if (params_supported(p, env_logr, &v)) return 1;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) return fail("no HIP device: libfhestr has no CPU fallback");
    if (device < 0 || device >= count) return fail("bad device index");
    HIP_TRY(hipSetDevice(device));
    Engine* e = new Engine();
    e->p = p;
    e->device = device;
    e->variant = v;
    e->variant_large = v;
    const EngineEnv env = EngineEnv::read();
    if (env.wide_fair >= 0) e->wide_fair_shift = (uint32_t)std::min(20, env.wide_fair);
    if (env.dense_per_cu >= 0) e->dense_per_cu = (uint32_t)env.dense_per_cu;
    if (env.cluster_fallback >= 0) e->cluster_fallback = env.cluster_fallback != 0;
    if (env.keep_busy >= 0) e->keep_busy = env.keep_busy != 0;
    if (env.overlap_streams >= 0) e->ovl_streams = std::min((int)Engine::OVL_MAX, std::max(2, env.overlap_streams));
    if (env.ks_mfma >= 0) e->ks_mfma_enabled = env.ks_mfma != 0;
    if (env.ks_chunks >= 0) e->ks_chunks_override = (uint32_t)env.ks_chunks;
    if (env.cluster_spin_limit >= 0) e->cluster_spin_limit = (uint32_t)std::max(64, env.cluster_spin_limit);
    if (env.cluster_mode > -2) e->cluster_mode = std::min(2, std::max(-1, env.cluster_mode));
    if (env.multibit_combine_max >= 0) e->multibit_combine_max = (uint32_t)std::min(1024, env.multibit_combine_max);
#ifdef FHESTR_TEST_HOOKS      // fault injection exists only in the test build (make testhooks), never in the product library
    if (env.cluster_test_fault >= 0) e->cluster_test_fault = (uint32_t)env.cluster_test_fault;
#endif
    if (env_logr == 0) {   // automatic: "wide" twin (same points per thread => same key layout) for big batches
        const BrVariant* w = find_variant(p, v->logR | 16);
        if (w) e->variant_large = w;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    e->cu_count = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    for (auto& ev : e->ev) HIP_TRY(hipEventCreate(&ev));
    *out = e;
    return 0;
}

Engine::~Engine() {
    (void)hipSetDevice(device);
    if (stream) (void)sync_all_streams();
    auto rel = [](void* ptr) { if (ptr) (void)hipFree(ptr); };
    rel(d_ksk_packed); rel(d_ksk_rowsum); rel(d_fbsk); rel(d_fbsk_dense); rel(d_luts); rel(d_in); rel(d_small); rel(d_small2); rel(d_out); rel(d_idx);
    rel(d_pool); rel(d_meta); rel(d_ws); rel(d_slot_exp); rel(d_cluster_ws); rel(d_cluster_ctl); rel(d_ksk_mfma); rel(d_ks_digits);
    for (int q = 0; q < OVL_MAX; q++) { rel(ovl_digits[q]); rel(ovl_small[q]); if (ovl_done[q]) (void)hipEventDestroy(ovl_done[q]); if (q >= 2 && ovl_stream[q]) (void)hipStreamDestroy(ovl_stream[q]); }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ring) if (e) (void)hipEventDestroy(e);
    for (auto& e : pipe_ev) if (e) (void)hipEventDestroy(e);
    if (ks_stream) (void)hipStreamDestroy(ks_stream);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

int Engine::set_stream(hipStream_t s, bool use_own) {
    if (use()) return 1;
    if (ks_stream) HIP_TRY(hipStreamSynchronize(ks_stream));
    HIP_TRY(hipStreamSynchronize(stream));
    stream = use_own ? own_stream : s;   // s == nullptr is HIP's default (null) stream
    return 0;
}

int Engine::lut_upload_dedup(const std::vector<uint64_t>& acc, uint32_t* id) {
    auto it = lut_dedup.find(acc);
    if (it != lut_dedup.end()) { *id = it->second; return 0; }
    if (lut_upload(acc.data(), id)) return 1;
    lut_dedup[acc] = *id;
    return 0;
}

int Engine::use() { HIP_TRY(hipSetDevice(device)); return 0; }

static int ensure(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(ptr, bytes));
    *cap = bytes;
    return 0;
}

int Engine::set_variant(int logR) {
    const BrVariant* v = find_variant(p, logR);
    if (!v) return fail("no such blind-rotation variant for these parameters");
    if (d_fbsk && v->logR != variant->logR)
        return fail("variant must be chosen before fhe_engine_load_keys (Fourier key layout depends on it)");
    variant = v;
    variant_large = v;
    shadow_fit = -1;
    if (logR == 0) {
        const BrVariant* w = find_variant(p, v->logR | 16);
        if (w) variant_large = w;
    }
    if (d_fbsk) {
        HIP_TRY(hipFuncSetAttribute(variant->rotate_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(variant->lds_bytes + (size_t)p.n * variant->lds_per_n)));
        HIP_TRY(hipFuncSetAttribute(variant_large->rotate_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(variant_large->lds_bytes + (size_t)p.n * variant_large->lds_per_n)));
        if (variant->rotate_combined_fn)
            HIP_TRY(hipFuncSetAttribute(variant->rotate_combined_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(variant->lds_bytes + (size_t)p.n * variant->lds_per_n)));
    }
    return 0;
}

int Engine::load_keys(const uint64_t* bsk_std, const uint64_t* ksk) {
    if (use()) return 1;
    const size_t ksk_len = (size_t)p.k * p.N * p.ks_level * (p.n + 1);
    const size_t bsk_len = (size_t)n_ggsw(p) * p.pbs_level * (p.k + 1) * (p.k + 1) * p.N;
    uint64_t *d_ksk_std = nullptr, *d_bsk_std = nullptr;
    HIP_TRY(hipMalloc((void**)&d_ksk_std, ksk_len * 8));
    if (hipMalloc((void**)&d_bsk_std, bsk_len * 8) != hipSuccess) {
        (void)hipFree(d_ksk_std);
        return fail("hipMalloc of the standard-domain bootstrap key failed");
    }
    hipError_t e = hipMemcpyAsync(d_ksk_std, ksk, ksk_len * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bsk_std, bsk_std, bsk_len * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
        (void)hipFree(d_ksk_std); (void)hipFree(d_bsk_std);
        return fail(std::string("key upload: ") + hipGetErrorString(e));
    }
    return install_keys(d_ksk_std, d_bsk_std);
}

// A tfhe-rs client's CompressedServerKey (shortint/server_key/compressed.rs): upload the bodies, expand the masks from
// the two compression seeds on the device (seeded_kernels.hip.h), then install as usual.
int Engine::load_seeded_keys(const uint8_t ksk_seed[16], const uint64_t* ksk_bodies, const uint8_t bsk_seed[16],
                             const uint64_t* bsk_bodies, uint64_t* bsk_std_out, uint64_t* ksk_out) {
    if (use()) return 1;
    const uint64_t ksk_rows = (uint64_t)p.k * p.N * p.ks_level, bsk_rows = (uint64_t)n_ggsw(p) * p.pbs_level * (p.k + 1);
    const size_t ksk_len = ksk_rows * (p.n + 1), bsk_len = bsk_rows * (p.k + 1) * p.N;
    const size_t bsk_body_words = bsk_rows * p.N;
    uint64_t *d_ksk_std = nullptr, *d_bsk_std = nullptr, *d_bodies = nullptr;
    uint8_t* d_sbox = nullptr;
    auto cleanup = [&] { (void)hipFree(d_ksk_std); (void)hipFree(d_bsk_std); (void)hipFree(d_bodies); (void)hipFree(d_sbox); };
    hipError_t e = hipMalloc((void**)&d_ksk_std, ksk_len * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bsk_std, bsk_len * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bodies, (bsk_body_words + ksk_rows) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_sbox, 256);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sbox, aes_sbox(), 256, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bodies, bsk_bodies, bsk_body_words * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bodies + bsk_body_words, ksk_bodies, ksk_rows * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { cleanup(); return fail(std::string("load_seeded_keys: ") + hipGetErrorString(e)); }
    SeededExpandArgs ka{}, ba{};
    aes128_round_keys(ksk_seed, ka.round_keys);
    ka.out = d_ksk_std; ka.rows = ksk_rows; ka.mask_per_row = p.n; ka.row_words = p.n + 1;
    aes128_round_keys(bsk_seed, ba.round_keys);
    ba.out = d_bsk_std; ba.rows = bsk_rows; ba.mask_per_row = p.k * p.N; ba.row_words = (p.k + 1) * p.N;
    const unsigned grid = (unsigned)cu_count * 8;
    hipLaunchKernelGGL(seeded_expand_kernel, dim3(grid), dim3(256), 0, stream, ka, d_sbox);
    hipLaunchKernelGGL(seeded_expand_kernel, dim3(grid), dim3(256), 0, stream, ba, d_sbox);
    hipLaunchKernelGGL(seeded_scatter_bodies_kernel, dim3(grid), dim3(256), 0, stream, d_bodies + bsk_body_words, d_ksk_std, ksk_rows,
                       p.n, p.n + 1);
    hipLaunchKernelGGL(seeded_scatter_bodies_kernel, dim3(grid), dim3(256), 0, stream, d_bodies, d_bsk_std, bsk_rows, p.k * p.N,
                       (p.k + 1) * p.N);
    e = hipGetLastError();
    if (e == hipSuccess && ksk_out) e = hipMemcpyAsync(ksk_out, d_ksk_std, ksk_len * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && bsk_std_out) e = hipMemcpyAsync(bsk_std_out, d_bsk_std, bsk_len * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_bodies); d_bodies = nullptr;
    (void)hipFree(d_sbox); d_sbox = nullptr;
    if (e != hipSuccess) { cleanup(); return fail(std::string("load_seeded_keys: ") + hipGetErrorString(e)); }
    return install_keys(d_ksk_std, d_bsk_std);
}

// Compressed big-key ciphertexts -> full ciphertexts in HBM (d_out, count x (kN+1) words) and/or host_out.
int Engine::expand_seeded_lwe(const uint8_t* seeds, const uint64_t* bodies, uint32_t count, uint64_t* d_out, uint64_t* host_out) {
    if (use()) return 1;
    if (count == 0) return 0;
    const uint32_t dim = p.k * p.N;
    const size_t words = (size_t)count * (dim + 1);
    uint8_t *d_seeds = nullptr, *d_sbox = nullptr;
    uint64_t *d_bodies = nullptr, *d_tmp = nullptr;
    auto cleanup = [&] { (void)hipFree(d_seeds); (void)hipFree(d_sbox); (void)hipFree(d_bodies); (void)hipFree(d_tmp); };
    hipError_t e = hipMalloc((void**)&d_seeds, (size_t)count * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&d_sbox, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bodies, (size_t)count * 8);
    if (e == hipSuccess && !d_out) e = hipMalloc((void**)&d_tmp, words * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_seeds, seeds, (size_t)count * 16, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sbox, aes_sbox(), 256, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bodies, bodies, (size_t)count * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { cleanup(); return fail(std::string("expand_seeded_lwe: ") + hipGetErrorString(e)); }
    uint64_t* target = d_out ? d_out : d_tmp;
    hipLaunchKernelGGL(seeded_lwe_expand_kernel, dim3(count), dim3(64), 0, stream, d_seeds, d_bodies, d_sbox, target, dim);
    e = hipGetLastError();
    if (e == hipSuccess && host_out) e = hipMemcpyAsync(host_out, target, words * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    cleanup();
    if (e != hipSuccess) return fail(std::string("expand_seeded_lwe: ") + hipGetErrorString(e));
    return 0;
}

// Server-key generation on the device (keygen_kernels.hip.h); replaces ServerKey::new
// (shortint/engine/server_side.rs:54-160) for callers that hold the secret keys next to the GPU.
int Engine::generate_keys(const uint64_t* glwe_sk, const uint64_t* small_sk, const uint8_t seed[32],
                          uint64_t* bsk_std_out, uint64_t* ksk_out) {
    if (use()) return 1;
    const size_t in_dim = (size_t)p.k * p.N;
    const size_t ksk_len = in_dim * p.ks_level * (p.n + 1);
    const size_t bsk_len = (size_t)n_ggsw(p) * p.pbs_level * (p.k + 1) * (p.k + 1) * p.N;
    for (size_t i = 0; i < in_dim; i++)
        if (glwe_sk[i] > 1) return fail("glwe_sk must be binary");
    for (size_t i = 0; i < p.n; i++)
        if (small_sk[i] > 1) return fail("small_sk must be binary");
    // plaintext bit of every GGSW: the key bits themselves, or their per-group products (multi-bit)
    const uint32_t ng = n_ggsw(p), gf = p.grouping_factor > 1 ? p.grouping_factor : 1;
    std::vector<uint64_t> bits(ng);
    for (uint32_t i = 0; i < ng; i++)
        bits[i] = gf == 1 ? small_sk[i] : multi_bit_key_bit(small_sk + (size_t)(i >> gf) * gf, gf, i & ((1u << gf) - 1));
    uint64_t *d_gsk = nullptr, *d_ssk = nullptr, *d_bits = nullptr, *d_ksk_std = nullptr, *d_bsk_std = nullptr;
    auto cleanup = [&] {
        (void)hipFree(d_gsk); (void)hipFree(d_ssk); (void)hipFree(d_bits); (void)hipFree(d_ksk_std); (void)hipFree(d_bsk_std);
    };
    hipError_t e = hipMalloc((void**)&d_gsk, in_dim * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_ssk, (size_t)p.n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bits, (size_t)ng * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bits, bits.data(), (size_t)ng * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMalloc((void**)&d_ksk_std, ksk_len * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_bsk_std, bsk_len * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gsk, glwe_sk, in_dim * 8, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ssk, small_sk, (size_t)p.n * 8, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { cleanup(); return fail(std::string("generate_keys: ") + hipGetErrorString(e)); }
    KeygenArgs a{d_gsk, d_ssk, d_bits, d_ksk_std, d_bsk_std, seed_from_bytes(seed), p.n, p.k, p.N,
                 p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level, p.lwe_std, p.glwe_std};
    hipLaunchKernelGGL(ksk_gen_kernel, dim3((unsigned)((in_dim + 63) / 64)), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(bsk_gen_kernel, dim3(ng), dim3(256), 0, stream, a);
    e = hipGetLastError();
    if (e == hipSuccess && ksk_out) e = hipMemcpyAsync(ksk_out, d_ksk_std, ksk_len * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && bsk_std_out) e = hipMemcpyAsync(bsk_std_out, d_bsk_std, bsk_len * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_gsk); d_gsk = nullptr;
    (void)hipFree(d_ssk); d_ssk = nullptr;
    (void)hipFree(d_bits); d_bits = nullptr;
    if (e != hipSuccess) { cleanup(); return fail(std::string("generate_keys: ") + hipGetErrorString(e)); }
    return install_keys(d_ksk_std, d_bsk_std);
}

// Takes ownership of the two standard-domain device buffers: repacks the KSK into byte planes and
// converts the BSK to the active variant's Fourier layout, then releases them.
// Standard-domain polynomials -> the variant's Fourier layout (bsk_convert_kernel / bsk_convert_large_kernel).
int Engine::convert_polys(const uint64_t* d_std, double* d_out, uint32_t n_polys) {
    const uint32_t k1 = p.k + 1;
    HIP_TRY(hipFuncSetAttribute(variant->convert_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)variant->convert_lds));
    if (variant->large) {
        const uint32_t blocks = n_polys < (uint32_t)cu_count ? n_polys : (uint32_t)cu_count;
        void* d_cws = nullptr;
        HIP_TRY(hipMalloc(&d_cws, (size_t)blocks * variant->convert_ws));
        void* args[] = {(void*)&d_std, (void*)&d_out, (void*)&n_polys, (void*)&d_cws};
        hipError_t e = hipLaunchKernel(variant->convert_fn, dim3(blocks), dim3(variant->convert_threads), args,
                                       variant->convert_lds, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_cws);
        if (e != hipSuccess) return fail(std::string("bsk conversion: ") + hipGetErrorString(e));
        return 0;
    }
    void* args[] = {(void*)&d_std, (void*)&d_out, (void*)&n_polys};
    HIP_TRY(hipLaunchKernel(variant->convert_fn, dim3(variant->convert_one_per_block ? n_polys : (n_polys + k1 - 1) / k1),
                            dim3(variant->convert_threads), args, variant->convert_lds, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

// Which power of w = e^{i pi / N} each slot of the Fourier layout evaluates at: transform the monomial X with the
// variant's own conversion kernel and read the angles off (pbs_multibit_kernels.hip.h, "any other shape").
int Engine::probe_slot_exponents() {
    const uint32_t k1 = p.k + 1, P = p.N / 2;
    std::vector<uint64_t> mono((size_t)k1 * p.N, 0);
    for (uint32_t r = 0; r < k1; r++) mono[(size_t)r * p.N + 1] = 1;
    uint64_t* d_mono = nullptr;
    double* d_spec = nullptr;
    HIP_TRY(hipMalloc((void**)&d_mono, mono.size() * 8));
    HIP_TRY(hipMalloc((void**)&d_spec, mono.size() * 8));
    HIP_TRY(hipMemcpy(d_mono, mono.data(), mono.size() * 8, hipMemcpyHostToDevice));
    int rc = convert_polys(d_mono, d_spec, k1);
    std::vector<double> spec((size_t)P * 2);
    if (!rc && hipMemcpy(spec.data(), d_spec, spec.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail("slot probe: copy");
    (void)hipFree(d_mono);
    (void)hipFree(d_spec);
    if (rc) return rc;
    std::vector<uint32_t> expo(P);
    const double pi = 3.14159265358979323846;
    for (uint32_t s = 0; s < P; s++) {
        const double turns = std::atan2(spec[2 * s + 1], spec[2 * s]) / pi * (double)p.N;   // in units of pi / N
        const long e = std::lround(turns);
        expo[s] = (uint32_t)(((e % (long)(2 * p.N)) + 2 * p.N) % (2 * p.N));
        if (std::fabs(turns - (double)e) > 0.01 || (expo[s] & 1) == 0)     // roots of X^N + 1 are the odd powers of w
            return fail("slot probe: the conversion kernel did not return a root of X^N + 1");
    }
    if (d_slot_exp) { HIP_TRY(hipFree(d_slot_exp)); d_slot_exp = nullptr; }
    HIP_TRY(hipMalloc((void**)&d_slot_exp, (size_t)P * 4));
    HIP_TRY(hipMemcpy(d_slot_exp, expo.data(), (size_t)P * 4, hipMemcpyHostToDevice));
    return 0;
}

int Engine::install_keys(uint64_t* d_ksk_std, uint64_t* d_std) {
    struct Guard {
        uint64_t*& a; uint64_t*& b;
        ~Guard() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } guard{d_ksk_std, d_std};
    const size_t bsk_len = (size_t)n_ggsw(p) * p.pbs_level * (p.k + 1) * (p.k + 1) * p.N;
    if (d_fbsk) { HIP_TRY(hipFree(d_fbsk)); d_fbsk = nullptr; }
    if (d_ksk_packed) { HIP_TRY(hipFree(d_ksk_packed)); d_ksk_packed = nullptr; }
    {   // repack into byte planes once; the 64-bit layout is then released
        const uint32_t rows = p.k * p.N * p.ks_level, osz = p.n + 1;
        HIP_TRY(hipMalloc((void**)&d_ksk_packed, (size_t)(rows / 4) * 8 * osz * 4));
        hipLaunchKernelGGL(ksk_pack_kernel, dim3((osz + 255) / 256, rows / 4), dim3(256), 0, stream,
                           d_ksk_std, d_ksk_packed, rows, osz);
        const uint32_t tiles = (p.k * p.N + KS_IC - 1) / KS_IC;
        if (d_ksk_rowsum) { HIP_TRY(hipFree(d_ksk_rowsum)); d_ksk_rowsum = nullptr; }
        HIP_TRY(hipMalloc((void**)&d_ksk_rowsum, (size_t)tiles * osz * 8));
        hipLaunchKernelGGL(ksk_rowsum_kernel, dim3((osz + 255) / 256, tiles), dim3(256), 0, stream,
                           d_ksk_std, d_ksk_rowsum, rows, osz, (uint32_t)KS_IC * p.ks_level);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (d_ksk_mfma) { HIP_TRY(hipFree(d_ksk_mfma)); d_ksk_mfma = nullptr; }
    // balanced base-256 digit planes in MFMA fragment order (ks_mfma_kernels.hip.h); a 16-slot group holds whole mask
    // elements with all their levels, so more than 16 levels (PARAM_MESSAGE_3_CARRY_4_COMPACT_PK_PBS_KS: 22) stay on the
    // byte-plane dot4 kernel
    if (ks_mfma_enabled && ks_mfma_supported(p.ks_level)) {
        const KsMfmaGeom g = ks_mfma_geom(p.k * p.N, p.n + 1, p.ks_level, p.ks_base_log);
        const size_t bytes = (size_t)g.col_groups * g.steps * 8 * 1024;
        HIP_TRY(hipMalloc((void**)&d_ksk_mfma, bytes));
        hipLaunchKernelGGL(ksk_repack_mfma_kernel, dim3(g.col_groups, g.steps), dim3(64), 0, stream, d_ksk_std, d_ksk_mfma, g);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
    }
    HIP_TRY(hipMalloc((void**)&d_fbsk, bsk_len * 8));   // N u64 -> N/2 c64: same byte count
    if (convert_polys(d_std, d_fbsk, (uint32_t)(bsk_len / p.N))) return 1;
    if (d_fbsk_dense) { HIP_TRY(hipFree(d_fbsk_dense)); d_fbsk_dense = nullptr; }
    if (variant_large->dense_convert_fn) {      // the copy of the key in FftSwap9's order (N = 1024, k = 2: 54.7 MB): dense and wide kernels
        HIP_TRY(hipMalloc((void**)&d_fbsk_dense, bsk_len * 8));
        uint32_t n_polys = (uint32_t)(bsk_len / p.N);
        void* cargs[] = {(void*)&d_std, (void*)&d_fbsk_dense, (void*)&n_polys};
        HIP_TRY(hipLaunchKernel(variant_large->dense_convert_fn, dim3(n_polys), dim3(variant_large->dense_convert_threads), cargs, 0, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (variant->combine_generic_fn && probe_slot_exponents()) return 1;
    HIP_TRY(hipFuncSetAttribute(variant->rotate_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(variant->lds_bytes + (size_t)p.n * variant->lds_per_n)));
    HIP_TRY(hipFuncSetAttribute(variant_large->rotate_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(variant_large->lds_bytes + (size_t)p.n * variant_large->lds_per_n)));
    if (variant->rotate_combined_fn)
        HIP_TRY(hipFuncSetAttribute(variant->rotate_combined_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(variant->lds_bytes + (size_t)p.n * variant->lds_per_n)));
    return 0;
}

// fill_accumulator: shortint/engine/mod.rs:72-128.  box_values[i] = the torus value the table returns on box i
void fill_accumulator_torus(const fhe_params_t& p, const uint64_t* box_values, std::vector<uint64_t>& acc) {
    const uint32_t N = p.N, k = p.k;
    acc.assign((size_t)(k + 1) * N, 0);
    uint64_t* body = acc.data() + (size_t)k * N;
    const uint32_t modulus_sup = p.msg_mod * p.carry_mod;
    const uint32_t box = N / modulus_sup;
    std::vector<uint64_t> tmp(N);
    for (uint32_t i = 0; i < modulus_sup; i++)
        for (uint32_t j = 0; j < box; j++) tmp[(size_t)i * box + j] = box_values[i];
    const uint32_t half = box / 2;
    for (uint32_t j = 0; j < half; j++) tmp[j] = 0 - tmp[j];
    for (uint32_t j = 0; j < N; j++) body[j] = tmp[(j + half) % N];   // rotate_left(half)
}

uint64_t fill_accumulator(const fhe_params_t& p, const uint64_t* table, std::vector<uint64_t>& acc) {
    const uint32_t modulus_sup = p.msg_mod * p.carry_mod;
    const uint64_t delta = (1ull << 63) / modulus_sup;
    uint64_t maxv = 0;
    std::vector<uint64_t> values(modulus_sup);
    for (uint32_t i = 0; i < modulus_sup; i++) {
        maxv = table[i] > maxv ? table[i] : maxv;
        values[i] = table[i] * delta;
    }
    fill_accumulator_torus(p, values.data(), acc);
    return maxv;
}

int Engine::lut_upload(const uint64_t* acc, uint32_t* id) {
    if (use()) return 1;
    const size_t glwe = (size_t)(p.k + 1) * p.N;
    if ((size_t)(n_luts + 1) * glwe * 8 > luts_cap) {
        size_t new_cap = luts_cap ? luts_cap * 2 : 64 * glwe * 8;
        uint64_t* nl = nullptr;
        HIP_TRY(hipMalloc((void**)&nl, new_cap));
        if (d_luts) {
            if (sync_all_streams()) return 1;      // pipelined calls on the other streams may still read the old table array
            HIP_TRY(hipMemcpy(nl, d_luts, (size_t)n_luts * glwe * 8, hipMemcpyDeviceToDevice));
            HIP_TRY(hipFree(d_luts));
        }
        d_luts = nl;
        luts_cap = new_cap;
    }
    HIP_TRY(hipMemcpyAsync(d_luts + (size_t)n_luts * glwe, acc, glwe * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *id = n_luts++;
    return 0;
}

int Engine::lut_download(uint32_t id, uint64_t* acc) {
    if (use()) return 1;
    if (id >= n_luts) return fail("bad LUT id");
    const size_t glwe = (size_t)(p.k + 1) * p.N;
    HIP_TRY(hipMemcpyAsync(acc, d_luts + (size_t)id * glwe, glwe * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

int Engine::ensure_batch(uint32_t count) {
    const size_t big = (size_t)p.k * p.N + 1, small = (size_t)p.n + 1;
    if (ensure((void**)&d_in, &cap_in, count * big * 8)) return 1;
    if (ensure((void**)&d_out, &cap_out, count * big * 8)) return 1;
    if (ensure((void**)&d_small, &cap_small, count * small * 8)) return 1;
    if (ensure((void**)&d_idx, &cap_idx, (size_t)count * 4)) return 1;
    return 0;
}

int Engine::launch_keyswitch(const uint64_t* d_big, uint64_t* d_sm, uint32_t count, hipStream_t on, bool shadow, int digits_slot) {
    if (!d_ksk_packed) return fail("keys not loaded");
    hipStream_t s = on ? on : stream;
    // grid.y = sample tiles; HIP caps grid.y at 65535
    if (count > 65535u * KSD_S) return fail("batch too large for one keyswitch launch (max 524280 LWEs)");
    const uint32_t in_dim = p.k * p.N, out_size = p.n + 1;
    HIP_TRY(hipMemsetAsync(d_sm, 0, (size_t)count * out_size * 8, s));
    if (d_ksk_mfma && !shadow) {
        // int8 matrix product on the matrix cores: digits -> A fragments, then 32x32x32 tiles over (samples, columns x planes, rows)
        const KsMfmaGeom g = ks_mfma_geom(in_dim, out_size, p.ks_level, p.ks_base_log);
        const uint32_t row_tiles = (count + 31) / 32;
        const size_t need = (size_t)row_tiles * g.steps * 1024;
        int8_t*& digits = digits_slot ? ovl_digits[digits_slot] : d_ks_digits;     // own buffer per overlapped stream
        size_t& cap_digits = digits_slot ? ovl_cap_digits[digits_slot] : cap_ks_digits;
        if (need > cap_digits) {
            if (digits) { if (sync_all_streams()) return 1; HIP_TRY(hipFree(digits)); }
            digits = nullptr; cap_digits = 0;
            HIP_TRY(hipMalloc((void**)&digits, need));
            HIP_TRY(hipMemsetAsync(digits, 0, need, s));
            cap_digits = need;
        }
        KsDecomposeArgs da{d_big, digits, g, count};
        hipLaunchKernelGGL(ks_decompose_kernel, dim3((2 * g.steps + 255) / 256, count), dim3(256), 0, s, da);
        uint32_t mt = 1;
        while (mt < 8 && mt < row_tiles) mt *= 2;
        const uint32_t gy = (row_tiles + mt - 1) / mt;
        // K split over workgroups: enough waves to fill the 1024 SIMDs about one and a half times, no more -- every extra
        // chunk adds batch x columns 64-bit atomics (measured at 256 LWEs: 8 chunks 55 us, 32 chunks 89 us for memset +
        // digits + product; scripts/ks_bench.py)
        uint32_t chunks = (6u * (uint32_t)cu_count + g.col_groups * gy * mt / 2) / (g.col_groups * gy * mt);
        if (ks_chunks_override) chunks = ks_chunks_override;
        chunks = std::max(1u, std::min(chunks, (g.steps + 7) / 8));
        chunks = std::max(chunks, (g.steps + ks_mfma_max_steps(p.ks_base_log) - 1) / ks_mfma_max_steps(p.ks_base_log));   // int32 accumulators
        const uint32_t spc = (g.steps + chunks - 1) / chunks;
        chunks = (g.steps + spc - 1) / spc;
        KsMfmaArgs ma{d_big, d_ksk_mfma, digits, d_sm, g, count, row_tiles, spc};
        const dim3 grid(g.col_groups, gy, chunks);
        switch (mt) {
            case 1: hipLaunchKernelGGL(keyswitch_mfma_kernel<1>, grid, dim3(64), 0, s, ma); break;
            case 2: hipLaunchKernelGGL(keyswitch_mfma_kernel<2>, grid, dim3(128), 0, s, ma); break;
            case 4: hipLaunchKernelGGL(keyswitch_mfma_kernel<4>, grid, dim3(256), 0, s, ma); break;
            default: hipLaunchKernelGGL(keyswitch_mfma_kernel<8>, grid, dim3(512), 0, s, ma); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (d_ksk_packed) {
        KeyswitchPackedArgs pa{d_big, d_ksk_packed, d_ksk_rowsum, d_sm, in_dim, out_size, p.ks_base_log, p.ks_level, count};
        if (shadow) {     // small-register variant: co-resident with the blind rotation of the previous batch
            constexpr int S = 4;
            dim3 pgrid((out_size + KS_COLS - 1) / KS_COLS, (count + S - 1) / S, (in_dim + KS_IC - 1) / KS_IC);
            hipLaunchKernelGGL((keyswitch_dot4_kernel<S, 8>), pgrid, dim3(KS_COLS), (size_t)KS_IC * p.ks_level * S, s, pa);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        dim3 pgrid((out_size + KS_COLS - 1) / KS_COLS, (count + KSD_S - 1) / KSD_S, (in_dim + KS_IC - 1) / KS_IC);
        hipLaunchKernelGGL((keyswitch_dot4_kernel<KSD_S, 2>), pgrid, dim3(KS_COLS), (size_t)KS_IC * p.ks_level * KSD_S, s, pa);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    return fail("keyswitch key not installed");
}

int Engine::launch_blind_rotate(const uint64_t* d_sm, const uint32_t* d_lut_idx, uint64_t* d_big,
                                uint32_t count, hipStream_t on, bool two_per_cu) {
    if (!d_fbsk) return fail("keys not loaded");
    if (n_luts == 0) return fail("no lookup table uploaded");
    BlindRotateArgs a{d_sm, d_lut_idx, d_luts, d_fbsk, d_big, p.n, p.pbs_base_log, count};
    a.grouping = 0;
    // two-LWEs-per-CU kernel: fair time-sliced priorities when every CU gets an even number of workgroups (pbs_kernels.hip.h)
    a.fair_shift = (!two_per_cu && wide_fair_shift && count > (uint32_t)cu_count && (((count + (uint32_t)cu_count - 1) / (uint32_t)cu_count) & 1u) == 0) ? wide_fair_shift : 0u;
    void* args[] = {(void*)&a};
    if (two_per_cu) {          // overlapped throughput mode: the compact layout whatever the batch size, on the given stream
        const BrVariant* w = variant_large;
        a.fair_shift = wide_fair_shift;      // the two launches that share the GPU progress at the same rate (110 k -> 122 k PBS/s)
        if (w->own_plan && w != variant) {
            if (!d_fbsk_dense) return fail("wide kernel: the key copy in its plan's order is missing");
            a.fbsk = d_fbsk_dense;
        }
        HIP_TRY(hipLaunchKernel(w->rotate_fn, dim3(count), dim3(w->threads), args, w->lds_bytes + (size_t)p.n * w->lds_per_n, on ? on : stream));
        return 0;
    }
    // one LWE per CU or fewer: spread it over more threads; above that: the compact layout that
    // lets two LWEs share a CU
    const BrVariant* v = count > (uint32_t)cu_count ? variant_large : variant;
    if (v->combine_fn && count <= multibit_combine_max) {
        // far fewer LWEs than CUs: the idle CUs prepare the groups' GGSWs (lwe_multi_bit_programmable_bootstrapping.rs
        // splits the same way over CPU threads), the rotation then runs n/G plain external products
        const size_t groups = p.n / p.grouping_factor;
        if (ensure(&d_ws, &cap_ws, (size_t)count * groups * v->combined_bytes)) return 1;
        MultiBitCombineArgs ca{a, reinterpret_cast<double2*>(d_ws)};
        void* cargs[] = {(void*)&ca};
        HIP_TRY(hipLaunchKernel(v->combine_fn, dim3((unsigned)groups, (unsigned)v->combine_grid_y, (count + v->combine_chunk - 1) / v->combine_chunk),
                                dim3(v->threads), cargs,
                                v->combine_lds, stream));
        BlindRotateArgs b = a;
        b.fbsk = reinterpret_cast<const double*>(d_ws);
        void* bargs[] = {(void*)&b};
        HIP_TRY(hipLaunchKernel(v->rotate_combined_fn, dim3(count), dim3(v->threads), bargs,
                                v->lds_bytes + (size_t)p.n * v->lds_per_n, stream));
        return 0;
    }
    if (v->extprod_fn) {
        // multi-bit PBS on a shape without a fused kernel: prepare the (LWE, group) GGSWs, then n/G external
        // products per LWE; sub-batches keep the prepared GGSWs within a fixed workspace
        const size_t groups = p.n / p.grouping_factor, per_lwe = groups * v->combined_bytes;
        const size_t rot_ws = v->large ? v->ws_bytes : 0;
        size_t cap = multibit_workspace_cap;
        if (cap == 0) {       // automatic: half of what is free now (plus what the workspace already holds), at most 64 GB
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            cap = std::min<size_t>((size_t)64 << 30, (free_b + cap_ws) / 2);
        }
        const uint32_t sub_max = (uint32_t)std::max<size_t>(1, std::min<size_t>(count, cap / (per_lwe + rot_ws)));
        if (ensure(&d_ws, &cap_ws, (size_t)sub_max * (per_lwe + rot_ws))) return 1;
        unsigned char* rot_base = reinterpret_cast<unsigned char*>(d_ws) + (size_t)sub_max * per_lwe;
        uint32_t logN = 0;
        while ((1u << logN) < p.N) logN++;
        const uint32_t ggsw_elems = (uint32_t)(v->combined_bytes / 16);
        const size_t combine_lds = ((size_t)(1u << ((logN + 1) / 2)) + (size_t)(1u << (logN + 1 - (logN + 1) / 2))) * 16;
        const size_t big = (size_t)p.k * p.N + 1;
        for (uint32_t first = 0; first < count; first += sub_max) {
            const uint32_t sub = std::min(sub_max, count - first);
            MultiBitCombineGenericArgs ca{d_sm + (size_t)first * (p.n + 1), reinterpret_cast<const double2*>(d_fbsk), d_slot_exp,
                                          reinterpret_cast<double2*>(d_ws), p.n, logN, p.N / 2, ggsw_elems, sub};
            void* cargs[] = {(void*)&ca};
            HIP_TRY(hipLaunchKernel(v->combine_generic_fn, dim3((unsigned)groups, (ggsw_elems + 511) / 512, (sub + 7) / 8), dim3(256),
                                    cargs, combine_lds, stream));
            BlindRotateArgs b{d_sm + (size_t)first * (p.n + 1), d_lut_idx ? d_lut_idx + first : nullptr, d_luts,
                              reinterpret_cast<const double*>(d_ws), d_big + (size_t)first * big, p.n, p.pbs_base_log, sub,
                              p.grouping_factor};
            if (v->large) {
                BlindRotateLargeArgs la{b, rot_base};
                void* largs[] = {(void*)&la};
                HIP_TRY(hipLaunchKernel(v->extprod_fn, dim3(sub), dim3(v->threads), largs,
                                        v->lds_bytes + (size_t)p.n * v->lds_per_n, stream));
            } else {
                void* bargs[] = {(void*)&b};
                HIP_TRY(hipLaunchKernel(v->extprod_fn, dim3(sub), dim3(v->threads), bargs,
                                        v->lds_bytes + (size_t)p.n * v->lds_per_n, stream));
            }
        }
        return 0;
    }
    // (a device with fewer than 8 * C compute units -- e.g. one XCD of a partitioned GPU -- cannot host a grid of the
    // cluster kernel: it takes the one-workgroup kernel below)
    // automatic mode: the whole-XCD kernel up to two LWEs per XCD (one LWE 12.3 ms, 16 LWEs 18.4 ms; the 8-CU clusters: 20.9 /
    // 21.4 ms), the 8-CU clusters above (256 LWEs: 1.15 k PBS/s against 0.85 k -- four LWEs in flight per XCD amortise the
    // hand-over latency better than two; profiles/r04_xcd_history.txt)
    if (v->xcd_fn && cluster_mode != 0 && cluster_mode != 2 && (cluster_mode == 1 || count <= std::min(cluster_max_batch, xcd_auto_max)) &&
        (uint32_t)cu_count >= 8u * (uint32_t)v->xcd_size) {
        // All CUs of an XCD per LWE (pbs_xcd_kernels.hip.h).  Up to 8 LWEs: one cluster per XCD, one 256-thread workgroup
        // per CU (the dynamic LDS request is padded past half a CU's LDS so that no CU takes two).  More: two clusters per
        // XCD, i.e. two workgroups on every CU -- the grid is exactly what the device holds, every workgroup must be resident.
        const uint32_t C = (uint32_t)v->xcd_size, quantum = 8 * C;
        const size_t lds2 = v->xcd_lds + (size_t)p.n * 4;
        if (xcd_per_cu < 0) {       // once per engine: do two of its workgroups fit a CU?  (registers, LDS: asked, not assumed)
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->xcd_fn, v->xcd_threads, lds2));
            xcd_per_cu = per_cu;
        }
        const uint32_t rounds = (count > 8 && xcd_per_cu >= 2) ? 2u : 1u;
        const uint32_t max_clusters = std::min<uint32_t>((uint32_t)CLUSTER_MAX, ((uint32_t)cu_count / quantum) * 8 * rounds);
        const uint32_t want = std::min(count, max_clusters);
        const uint32_t grid = (want + 7) / 8 * quantum;
        const size_t lds = grid <= (uint32_t)cu_count ? std::max(lds2, v->xcd_lds_one_per_cu) : lds2;
        if (ensure(&d_cluster_ws, &cap_cluster_ws, (size_t)max_clusters * v->xcd_ws)) return 1;
        if (!d_cluster_ctl) {
            HIP_TRY(hipMalloc((void**)&d_cluster_ctl, sizeof(ClusterCtl) + sizeof(ClusterStatus)));
            HIP_TRY(hipMemsetAsync(d_cluster_ctl, 0, sizeof(ClusterCtl) + sizeof(ClusterStatus), stream));
        }
        HIP_TRY(hipMemsetAsync(d_cluster_ctl, 0, sizeof(ClusterCtl), stream));
        ClusterCtl* ctl = reinterpret_cast<ClusterCtl*>(d_cluster_ctl);
        BlindRotateClusterArgs ka{a, reinterpret_cast<unsigned char*>(d_cluster_ws), ctl, reinterpret_cast<ClusterStatus*>(ctl + 1),
                                  cluster_spin_limit, cluster_test_fault};
        void* kargs[] = {(void*)&ka};
        HIP_TRY(hipLaunchKernel(v->xcd_fn, dim3(grid), dim3(v->xcd_threads), kargs, lds, stream));
        return cluster_settle(d_sm, d_lut_idx, d_big, count);
    }
    if (v->cluster_fn && cluster_mode != 0 && (cluster_mode >= 1 || count <= cluster_max_batch) &&
        (uint32_t)cu_count >= 8u * (uint32_t)v->cluster_size) {
        // Several CUs per LWE: the grid is a whole number of 8 * C workgroups (the dispatcher deals workgroups
        // round-robin over the 8 XCDs, the kernel forms its clusters from what each XCD actually received),
        // never more than one workgroup per CU -- every workgroup of the grid must be resident at once.
        const uint32_t C = (uint32_t)v->cluster_size, quantum = 8 * C;
        const uint32_t max_clusters = std::min<uint32_t>((uint32_t)CLUSTER_MAX, ((uint32_t)cu_count / quantum) * 8);
        const uint32_t want = std::min(count, max_clusters);
        const uint32_t grid = (want + 7) / 8 * quantum;
        if (ensure(&d_cluster_ws, &cap_cluster_ws, (size_t)max_clusters * v->cluster_ws)) return 1;
        if (!d_cluster_ctl) {
            HIP_TRY(hipMalloc((void**)&d_cluster_ctl, sizeof(ClusterCtl) + sizeof(ClusterStatus)));
            HIP_TRY(hipMemsetAsync(d_cluster_ctl, 0, sizeof(ClusterCtl) + sizeof(ClusterStatus), stream));
        }
        // tickets and flags start from zero; the status words behind them are sticky (read by cluster_status())
        HIP_TRY(hipMemsetAsync(d_cluster_ctl, 0, sizeof(ClusterCtl), stream));
        ClusterCtl* ctl = reinterpret_cast<ClusterCtl*>(d_cluster_ctl);
        BlindRotateClusterArgs ka{a, reinterpret_cast<unsigned char*>(d_cluster_ws), ctl, reinterpret_cast<ClusterStatus*>(ctl + 1),
                                  cluster_spin_limit, cluster_test_fault};
        void* kargs[] = {(void*)&ka};
        HIP_TRY(hipLaunchKernel(v->cluster_fn, dim3(grid), dim3(v->threads), kargs, v->cluster_lds + (size_t)p.n * 4, stream));
        return cluster_settle(d_sm, d_lut_idx, d_big, count);
    }
    if (v->large) {
        if (ensure(&d_ws, &cap_ws, (size_t)count * v->ws_bytes)) return 1;
        BlindRotateLargeArgs la{a, reinterpret_cast<unsigned char*>(d_ws)};
        void* largs[] = {(void*)&la};
        HIP_TRY(hipLaunchKernel(v->rotate_fn, dim3(count), dim3(v->threads), largs,
                                v->lds_bytes + (size_t)p.n * v->lds_per_n, stream));
        return 0;
    }
    // keep-busy mode (fhe_engine_set_keep_busy): a launch that would leave more than half of the CUs idle carries replicas of
    // its workgroups on them (they recompute and store nothing) -- the part then keeps its clock for the large launch that
    // follows (2.22 -> 2.39 GHz over 14 ms otherwise, profiles/r03_after_idle.txt), at the price of the energy
    if (v->dense_fn && d_fbsk_dense && dense_per_cu && count > dense_per_cu * (uint32_t)cu_count) {       // more than two LWEs per CU: the variant that puts four on one
        a.fair_shift = 0;
        a.fbsk = d_fbsk_dense;
        HIP_TRY(hipLaunchKernel(v->dense_fn, dim3(count), dim3(v->threads), args, v->dense_lds + (size_t)p.n * v->lds_per_n, stream));
        return 0;
    }
    uint32_t grid = count;
    if (keep_busy && !v->wide && !v->large && count * 2 <= (uint32_t)cu_count) grid = count * ((uint32_t)cu_count / count);
    if (v->own_plan && v != variant) {
        if (!d_fbsk_dense) return fail("wide kernel: the key copy in its plan's order is missing");
        a.fbsk = d_fbsk_dense;
    }
    HIP_TRY(hipLaunchKernel(v->rotate_keypf_fn && v->wide ? v->rotate_keypf_fn : v->rotate_fn, dim3(grid), dim3(v->threads), args,
                            v->lds_bytes + (size_t)p.n * v->lds_per_n, stream));
    return 0;
}

// Can a 64-register keyswitch wave sit on a SIMD next to the blind rotation's waves?  (one workgroup per CU, its waves
// spread over the four SIMDs, registers allocated in blocks of 8 out of 512 per SIMD lane)
bool Engine::shadow_keyswitch_fits() {
    if (shadow_fit < 0) {
        hipFuncAttributes fa{};
        shadow_fit = 0;
        if (hipFuncGetAttributes(&fa, variant->rotate_fn) == hipSuccess) {
            const int waves_per_simd = (variant->threads / 64 + 3) / 4;
            const int regs = (fa.numRegs + 7) / 8 * 8;
            shadow_fit = waves_per_simd * regs + 64 <= 512 ? 1 : 0;
        }
    }
    return shadow_fit == 1;
}

int Engine::ks_pbs_dev(const uint64_t* d_big_in, const uint32_t* d_lut_idx, uint64_t* d_big_out,
                       uint32_t count, bool allow_pipeline) {
    if (use()) return 1;
    if (count == 0) return 0;
    const size_t small = (size_t)p.n + 1;
    if (ensure((void**)&d_small, &cap_small, count * small * 8)) return 1;
    // HIP events on the launch stream: per-call kernel durations without host synchronisation
    constexpr size_t RING = 1024;
    if (ring.empty()) {
        ring.resize(RING * 4);
        for (auto& e : ring) HIP_TRY(hipEventCreate(&e));
    }
    hipEvent_t* e4 = &ring[(ring_used % RING) * 4];     // keyswitch start / end, blind rotation start / end
    ring_used++;
    if (allow_pipeline && pipeline == 2 && stream == own_stream && variant_large != variant && variant_large->wide &&
        !variant->extprod_fn && !variant->combine_fn && count <= (uint32_t)cu_count) {
        // Overlapped batches (fhe_engine_set_pipeline(2)): consecutive calls alternate between ovl_streams (2) streams and
        // run on the two-LWEs-per-CU kernel with time-sliced priorities (BlindRotateArgs::fair_shift), so two 256-LWE
        // launches share every CU and progress at the same rate -- the pace of a 512-LWE launch (2.09 ms per call,
        // 122 k PBS/s) although every batch has 256 LWEs; the price is each call's latency (4.1 ms).  Three streams
        // measured slower (119 k): the third launch's keyswitch waits for idle CUs.  Same ordering rules as mode 1, plus
        // write-after-write / write-after-read against the calls still in flight on the other streams.  Results are
        // those of the large-batch kernel.
        const int ns = ovl_streams;
        if (!ks_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&ks_stream, hipStreamNonBlocking));
            for (auto& e : pipe_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (int q = 2; q < ns; q++)
            if (!ovl_stream[q]) {
                HIP_TRY(hipStreamCreateWithFlags(&ovl_stream[q], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ovl_done[q], hipEventDisableTiming));
            }
        for (int q = 0; q < 2; q++)
            if (!ovl_done[q]) HIP_TRY(hipEventCreateWithFlags(&ovl_done[q], hipEventDisableTiming));
        const int slot = (int)(pipe_calls % (uint64_t)ns);
        hipStream_t s = slot == 0 ? stream : slot == 1 ? ks_stream : ovl_stream[slot];
        if (ensure((void**)&ovl_small[slot], &ovl_cap_small[slot], count * small * 8)) return 1;
        uint64_t* sm = ovl_small[slot];
        const size_t big = (size_t)p.k * p.N + 1;
        const char *in_lo = (const char*)d_big_in, *in_hi = in_lo + (size_t)count * big * 8;
        const char *out_lo = (const char*)d_big_out, *out_hi = out_lo + (size_t)count * big * 8;
        if (pipe_calls == 0) HIP_TRY(hipEventRecord(pipe_ev[4], stream));       // what the engine stream held before this run
        if (pipe_calls < (uint64_t)ns && slot != 0) HIP_TRY(hipStreamWaitEvent(s, pipe_ev[4], 0));
        if (pipe_input_ready) {
            HIP_TRY(hipStreamWaitEvent(s, pipe_input_ready, 0));
            pipe_input_ready = nullptr;
        }
        auto overlaps = [](const char* a_lo, const char* a_hi, const std::vector<ByteRange>& rs) {
            for (const auto& r : rs)
                if (a_lo < r.hi && r.lo < a_hi) return true;
            return false;
        };
        for (int q = 0; q < ns; q++) {       // every call of this run on the other streams, not only their latest
            if (q == slot || !ovl_done[q] || (ovl_ins[q].empty() && ovl_outs[q].empty())) continue;
            if (overlaps(in_lo, in_hi, ovl_outs[q]) || overlaps(out_lo, out_hi, ovl_outs[q]) || overlaps(out_lo, out_hi, ovl_ins[q]))
                HIP_TRY(hipStreamWaitEvent(s, ovl_done[q], 0));
        }
        auto remember = [](std::vector<ByteRange>& rs, const char* lo, const char* hi) {
            for (auto& r : rs)
                if (r.lo == lo && r.hi == hi) return;
            if (rs.size() < 32) { rs.push_back({lo, hi}); return; }
            rs[0].lo = std::min(rs[0].lo, lo);          // many distinct buffers: fold into one conservative range
            rs[0].hi = std::max(rs[0].hi, hi);
        };
        remember(ovl_ins[slot], in_lo, in_hi);
        remember(ovl_outs[slot], out_lo, out_hi);
        HIP_TRY(hipEventRecord(e4[0], s));
        if (launch_keyswitch(d_big_in, sm, count, s, false, slot)) return 1;
        HIP_TRY(hipEventRecord(e4[1], s));
        HIP_TRY(hipEventRecord(e4[2], s));
        if (launch_blind_rotate(sm, d_lut_idx, d_big_out, count, s, true)) return 1;
        HIP_TRY(hipEventRecord(e4[3], s));
        HIP_TRY(hipEventRecord(ovl_done[slot], s));
        pipe_calls++;
        return 0;
    }
    if (allow_pipeline && pipeline == 1 && stream == own_stream && !variant->large && !variant->wide && !variant->extprod_fn &&
        count <= (uint32_t)cu_count && shadow_keyswitch_fits()) {
        // Pipelined mode (fhe_engine_set_pipeline): the keyswitch of this call runs on a second stream, in a 64-VGPR
        // variant whose waves fit next to the two 220-VGPR waves per SIMD of the blind rotation still running for the
        // previous call, into the other of two small-ciphertext buffers.  Calls are independent unless this call's
        // input overlaps the previous call's output (older outputs are ordered by the buffer hand-over below).
        if (!ks_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&ks_stream, hipStreamNonBlocking));
            for (auto& e : pipe_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (ensure((void**)&d_small2, &cap_small2, count * small * 8)) return 1;
        const int slot = (int)(pipe_calls & 1);
        uint64_t* sm = slot ? d_small2 : d_small;
        const size_t big = (size_t)p.k * p.N + 1;
        const char *in_lo = (const char*)d_big_in, *in_hi = in_lo + (size_t)count * big * 8;
        const char* prev = (const char*)pipe_out[slot ^ 1];
        if (pipe_calls == 0) {
            // first pipelined call after anything else: whatever the engine stream already holds -- a serial call's
            // keyswitch / blind rotation that still use d_small, uploads, a caller's kernels -- comes first
            HIP_TRY(hipEventRecord(pipe_ev[4], stream));
            HIP_TRY(hipStreamWaitEvent(ks_stream, pipe_ev[4], 0));
        }
        if (pipe_input_ready) {       // fhe_engine_pipeline_input_event: the producer of this call's input
            HIP_TRY(hipStreamWaitEvent(ks_stream, pipe_input_ready, 0));
            pipe_input_ready = nullptr;
        }
        if (pipe_calls >= 1 && prev && in_lo < prev + pipe_out_bytes[slot ^ 1] && prev < in_hi)
            HIP_TRY(hipStreamWaitEvent(ks_stream, pipe_ev[2 + (slot ^ 1)], 0));              // chained: input = previous output
        if (pipe_calls >= 2) HIP_TRY(hipStreamWaitEvent(ks_stream, pipe_ev[2 + slot], 0));    // blind rotation that read `sm` two calls ago
        HIP_TRY(hipEventRecord(e4[0], ks_stream));
        if (launch_keyswitch(d_big_in, sm, count, ks_stream, true)) return 1;
        HIP_TRY(hipEventRecord(e4[1], ks_stream));
        HIP_TRY(hipEventRecord(pipe_ev[slot], ks_stream));
        HIP_TRY(hipStreamWaitEvent(stream, pipe_ev[slot], 0));
        HIP_TRY(hipEventRecord(e4[2], stream));
        if (launch_blind_rotate(sm, d_lut_idx, d_big_out, count)) return 1;
        HIP_TRY(hipEventRecord(e4[3], stream));
        HIP_TRY(hipEventRecord(pipe_ev[2 + slot], stream));
        pipe_out[slot] = d_big_out;
        pipe_out_bytes[slot] = (size_t)count * big * 8;
        pipe_calls++;
        return 0;
    }
    if (ks_stream && pipe_calls) {        // a serial call after pipelined ones: the other streams must be done with the shared buffers
        HIP_TRY(hipStreamSynchronize(ks_stream));
        for (int q = 2; q < OVL_MAX; q++) if (ovl_stream[q]) HIP_TRY(hipStreamSynchronize(ovl_stream[q]));
        end_pipeline_run();
    }
    pipe_input_ready = nullptr;           // serial calls are stream-ordered: nothing to wait for
    HIP_TRY(hipEventRecord(e4[0], stream));
    if (launch_keyswitch(d_big_in, d_small, count)) return 1;
    HIP_TRY(hipEventRecord(e4[1], stream));
    HIP_TRY(hipEventRecord(e4[2], stream));
    if (launch_blind_rotate(d_small, d_lut_idx, d_big_out, count)) return 1;
    HIP_TRY(hipEventRecord(e4[3], stream));
    return 0;
}

int Engine::check_lut_idx(const uint32_t* lut_idx, uint32_t count) const {
    if (n_luts == 0) return fail("no lookup table uploaded");
    if (lut_idx)
        for (uint32_t i = 0; i < count; i++)
            if (lut_idx[i] >= n_luts) return fail("lut_idx out of range");
    return 0;
}

int Engine::ks_pbs_host(const uint64_t* in, const uint32_t* lut_idx, uint64_t* out, uint32_t count) {
    if (use()) return 1;
    if (count == 0) return 0;
    if (check_lut_idx(lut_idx, count)) return 1;
    if (ensure_batch(count)) return 1;
    const size_t big = (size_t)p.k * p.N + 1;
    HIP_TRY(hipMemcpyAsync(d_in, in, count * big * 8, hipMemcpyHostToDevice, stream));
    if (lut_idx) HIP_TRY(hipMemcpyAsync(d_idx, lut_idx, (size_t)count * 4, hipMemcpyHostToDevice, stream));
    if (ks_pbs_dev(d_in, lut_idx ? d_idx : nullptr, d_out, count)) return 1;
    HIP_TRY(hipMemcpyAsync(out, d_out, count * big * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return cluster_check();
}

int Engine::keyswitch_host(const uint64_t* in, uint64_t* out_small, uint32_t count) {
    if (use()) return 1;
    if (count == 0) return 0;
    if (ensure_batch(count)) return 1;
    const size_t big = (size_t)p.k * p.N + 1, small = (size_t)p.n + 1;
    HIP_TRY(hipMemcpyAsync(d_in, in, count * big * 8, hipMemcpyHostToDevice, stream));
    if (launch_keyswitch(d_in, d_small, count)) return 1;
    HIP_TRY(hipMemcpyAsync(out_small, d_small, count * small * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

int Engine::pbs_host(const uint64_t* in_small, const uint32_t* lut_idx, uint64_t* out, uint32_t count) {
    if (use()) return 1;
    if (count == 0) return 0;
    if (check_lut_idx(lut_idx, count)) return 1;
    if (ensure_batch(count)) return 1;
    const size_t big = (size_t)p.k * p.N + 1, small = (size_t)p.n + 1;
    HIP_TRY(hipMemcpyAsync(d_small, in_small, count * small * 8, hipMemcpyHostToDevice, stream));
    if (lut_idx) HIP_TRY(hipMemcpyAsync(d_idx, lut_idx, (size_t)count * 4, hipMemcpyHostToDevice, stream));
    if (launch_blind_rotate(d_small, lut_idx ? d_idx : nullptr, d_out, count)) return 1;
    HIP_TRY(hipMemcpyAsync(out, d_out, count * big * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return cluster_check();
}

// Small-key order (programmable_bootstrap_keyswitch_assign, shortint/server_key/mod.rs:859-932):
// ciphertexts live under the small LWE key; bootstrap first, then keyswitch back.
int Engine::pbs_ks_host(const uint64_t* in_small, const uint32_t* lut_idx, uint64_t* out_small, uint32_t count) {
    if (use()) return 1;
    if (count == 0) return 0;
    if (check_lut_idx(lut_idx, count)) return 1;
    if (ensure_batch(count)) return 1;
    const size_t small = (size_t)p.n + 1;
    if (ensure((void**)&d_small2, &cap_small2, count * small * 8)) return 1;
    HIP_TRY(hipMemcpyAsync(d_small, in_small, count * small * 8, hipMemcpyHostToDevice, stream));
    if (lut_idx) HIP_TRY(hipMemcpyAsync(d_idx, lut_idx, (size_t)count * 4, hipMemcpyHostToDevice, stream));
    if (launch_blind_rotate(d_small, lut_idx ? d_idx : nullptr, d_out, count)) return 1;
    if (launch_keyswitch(d_out, d_small2, count)) return 1;
    HIP_TRY(hipMemcpyAsync(out_small, d_small2, count * small * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return cluster_check();
}

int Engine::lincomb_dev(const uint64_t* d_pool_, const uint32_t* d_off, const uint32_t* d_src,
                        const int32_t* d_coeff, const uint64_t* d_cst, uint64_t* d_o, uint32_t jobs) {
    if (jobs == 0) return 0;
    LincombArgs a{d_pool_, d_off, d_src, d_coeff, d_cst, d_o, p.k * p.N + 1, jobs};
    hipLaunchKernelGGL(lincomb_kernel, dim3(jobs), dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int Engine::lincomb_batch_dev(const uint64_t* d_pool_, const uint32_t* d_off, const uint32_t* d_src, const int32_t* d_coeff,
                              const uint64_t* d_cst, uint64_t* d_o, uint32_t jobs, uint32_t instances, uint32_t src_slot, uint32_t src_inst,
                              uint32_t out_job, uint32_t out_inst, const uint32_t* d_lut_in, uint32_t* d_lut_out) {
    if (jobs == 0 || instances == 0) return 0;
    if ((uint64_t)jobs * instances > 0x7FFFFFFFull) return fail("lincomb batch: jobs * instances exceeds the grid limit");
    LincombBatchArgs b{{d_pool_, d_off, d_src, d_coeff, d_cst, d_o, p.k * p.N + 1, jobs}, instances, src_slot, src_inst, out_job, out_inst,
                       d_lut_in, d_lut_out};
    hipLaunchKernelGGL(lincomb_batch_kernel, dim3(jobs * instances), dim3(256), 0, stream, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

int Engine::restride_dev(const uint64_t* d_i, uint64_t* d_o, uint32_t slots, uint32_t instances, uint32_t in_slot, uint32_t in_inst,
                         uint32_t out_slot, uint32_t out_inst) {
    if (slots == 0 || instances == 0) return 0;
    hipLaunchKernelGGL(lwe_restride_kernel, dim3(slots * instances), dim3(256), 0, stream, d_i, d_o, p.k * p.N + 1, instances, in_slot, in_inst,
                       out_slot, out_inst);
    HIP_TRY(hipGetLastError());
    return 0;
}

int Engine::lincomb_host(const uint64_t* pool, uint32_t pool_count, const uint32_t* off,
                         const uint32_t* src, const int32_t* coeff, const uint64_t* cst,
                         uint64_t* out, uint32_t jobs) {
    if (use()) return 1;
    if (jobs == 0) return 0;
    const size_t big = (size_t)p.k * p.N + 1;
    const uint32_t terms = off[jobs];
    for (uint32_t t = 0; t < terms; t++)
        if (src[t] >= pool_count) return fail("lincomb source index out of range");
    if (ensure((void**)&d_pool, &cap_pool, (size_t)pool_count * big * 8)) return 1;
    if (ensure((void**)&d_out, &cap_out, (size_t)jobs * big * 8)) return 1;
    // meta: off | src | coeff | cst (8-byte aligned)
    const size_t o_off = 0, o_src = o_off + ((size_t)(jobs + 1) * 4 + 7) / 8 * 8;
    const size_t o_coeff = o_src + ((size_t)terms * 4 + 7) / 8 * 8;
    const size_t o_cst = o_coeff + ((size_t)terms * 4 + 7) / 8 * 8;
    const size_t meta_bytes = o_cst + (size_t)jobs * 8;
    if (ensure((void**)&d_meta, &cap_meta, meta_bytes)) return 1;
    unsigned char* m = reinterpret_cast<unsigned char*>(d_meta);
    HIP_TRY(hipMemcpyAsync(d_pool, pool, (size_t)pool_count * big * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(m + o_off, off, (size_t)(jobs + 1) * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(m + o_src, src, (size_t)terms * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(m + o_coeff, coeff, (size_t)terms * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(m + o_cst, cst, (size_t)jobs * 8, hipMemcpyHostToDevice, stream));
    if (lincomb_dev(d_pool, (const uint32_t*)(m + o_off), (const uint32_t*)(m + o_src),
                    (const int32_t*)(m + o_coeff), (const uint64_t*)(m + o_cst), d_out, jobs))
        return 1;
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)jobs * big * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

int Engine::last_kernel_ms(float ms[2]) {
    if (use()) return 1;
    if (ring_used == 0) return fail("no ks_pbs call recorded");
    hipEvent_t* e4 = &ring[((ring_used - 1) % 1024) * 4];
    if (sync_all_streams()) return 1;
    HIP_TRY(hipEventSynchronize(e4[3]));
    HIP_TRY(hipEventElapsedTime(&ms[0], e4[0], e4[1]));
    HIP_TRY(hipEventElapsedTime(&ms[1], e4[2], e4[3]));
    return 0;
}

int Engine::kernel_times(double total_ms[2], uint32_t* calls, bool reset) {
    if (use()) return 1;
    if (sync_all_streams()) return 1;       // mode 2 records on every overlap stream
    const size_t nrec = ring_used < 1024 ? ring_used : 1024;
    total_ms[0] = total_ms[1] = 0.0;
    for (size_t c = 0; c < nrec; c++) {
        hipEvent_t* e4 = &ring[((ring_used - 1 - c) % 1024) * 4];
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, e4[0], e4[1]));
        HIP_TRY(hipEventElapsedTime(&b, e4[2], e4[3]));
        total_ms[0] += a;
        total_ms[1] += b;
    }
    *calls = (uint32_t)nrec;
    if (reset) ring_used = 0;
    return 0;
}

#ifdef FHESTR_WALL
extern "C" int fhe_debug_read_wall(unsigned long long* out, size_t count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wall), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#ifdef FHESTR_STAMPS
extern "C" int fhe_debug_read_stamps(unsigned long long* out, size_t count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

int Engine::synchronize() {
    if (use()) return 1;
    if (sync_all_streams()) return 1;
    end_pipeline_run();                   // every stream is idle: the next pipelined call starts a new run
    return cluster_check();
}

int Engine::sync_all_streams() {
    if (ks_stream) HIP_TRY(hipStreamSynchronize(ks_stream));
    for (int q = 2; q < OVL_MAX; q++) if (ovl_stream[q]) HIP_TRY(hipStreamSynchronize(ovl_stream[q]));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

void Engine::end_pipeline_run() {
    pipe_calls = 0;
    pipe_out[0] = pipe_out[1] = nullptr;
    for (int q = 0; q < OVL_MAX; q++) { ovl_ins[q].clear(); ovl_outs[q].clear(); }
}

// The cluster kernel never hangs on a hand-over that does not arrive: it gives up, finishes with garbage and says so
// in its status words.  Every host-visible completion point asks here.
// A launch of one of the multi-CU kernels is settled before the call returns: the host waits for it, reads the
// status words and, when the formation or a hand-over gave up -- a foreign kernel held compute units the grid needed;
// the kernel drained with invalid results instead of hanging -- runs the same batch on the one-workgroup kernel, which
// needs nothing resident but itself.  These kernels take 10 ms and more per launch: the synchronisation costs nothing
// measurable.  FHESTR_CLUSTER_FALLBACK=0: the round-3 behaviour (checked at the next completion point, reported as an error).
int Engine::cluster_settle(const uint64_t* d_sm, const uint32_t* d_lut_idx, uint64_t* d_big, uint32_t count) {
    if (!cluster_fallback) { cluster_unchecked = true; return 0; }
    ClusterStatus st{};
    ClusterStatus* d_st = reinterpret_cast<ClusterStatus*>(reinterpret_cast<ClusterCtl*>(d_cluster_ctl) + 1);
    HIP_TRY(hipMemcpyAsync(&st, d_st, sizeof(st), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    cluster_last = st.clusters;
    if (!st.error) return 0;
    HIP_TRY(hipMemsetAsync(d_st, 0, sizeof(st), stream));
    cluster_fallbacks++;
    cluster_last_error = st.error;
    const int saved = cluster_mode;
    cluster_mode = 0;
    const int rc = launch_blind_rotate(d_sm, d_lut_idx, d_big, count);
    cluster_mode = saved;
    return rc;
}

int Engine::cluster_check() {
    if (!cluster_unchecked || !d_cluster_ctl) return 0;
    ClusterStatus st{};
    HIP_TRY(hipMemcpy(&st, reinterpret_cast<ClusterCtl*>(d_cluster_ctl) + 1, sizeof(st), hipMemcpyDeviceToHost));
    cluster_unchecked = false;
    cluster_last = st.clusters;
    if (st.error) {
        HIP_TRY(hipMemset(reinterpret_cast<ClusterCtl*>(d_cluster_ctl) + 1, 0, sizeof(st)));
        return fail("blind_rotate_cluster_kernel: a cluster hand-over timed out (code " + std::to_string(st.error) +
                    "); results of that launch are invalid");
    }
    return 0;
}

}  // namespace fhe
