// engine.h -- internal C++ interface of the engine (the public surface is include/fhestr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fhestr.h"

// bumped whenever a device kernel changes; profiles/r03_counters.json records the revision its
// rocprofv3 counters were taken on and bench.py only attaches them to a matching build
#define FHESTR_KERNEL_REVISION "r04.1"

namespace fhe {

// GGSWs in the bootstrapping key: n (classic PBS) or n/g * 2^g (multi-bit, grouping factor g)
inline uint32_t n_ggsw(const fhe_params_t& p) {
    return p.grouping_factor > 1 ? p.n / p.grouping_factor * (1u << p.grouping_factor) : p.n;
}
// plaintext bit GGSW `sel` of a group encrypts (lwe_multi_bit_bootstrap_key_generation.rs:401-427):
// product over the group's key bits of (s_b if selector bit g-1-b is set else 1 - s_b)
inline uint64_t multi_bit_key_bit(const uint64_t* group_bits, uint32_t g, uint32_t sel) {
    uint64_t prod = 1;
    for (uint32_t b = 0; b < g; b++) prod *= group_bits[b] ^ (((sel >> (g - 1 - b)) & 1) ^ 1);
    return prod;
}

void aes128_round_keys(const uint8_t key[16], uint8_t rk[11][16]);   // seeded_keys.cpp
const uint8_t* aes_sbox();

extern thread_local std::string g_last_error;
int fail(const std::string& msg);

struct BrVariant;

// fill_accumulator: shortint/engine/mod.rs:72-128 (host side, no device needed)
uint64_t fill_accumulator(const fhe_params_t& p, const uint64_t* table, std::vector<uint64_t>& acc);
// the same from torus values per box (tables whose entries are not multiples of delta: Circuit::pbs_full_box)
void fill_accumulator_torus(const fhe_params_t& p, const uint64_t* box_values, std::vector<uint64_t>& acc);

// Environment switches (diagnostics / A/B measurements only; -1 or -2 = not set).  Read once per engine, engine.hip: EngineEnv::read.
struct EngineEnv {
    int log2_points = 0;              // FHESTR_LOG2_POINTS          blind-rotation variant selector (fhe_engine_set_variant)
    int wide_fair = -1;               // FHESTR_WIDE_FAIR            log2 ticks of the two-LWEs-per-CU kernel's priority slices, 0 = off
    int keep_busy = -1;               // FHESTR_KEEP_BUSY            fhe_engine_set_keep_busy at creation
    int overlap_streams = -1;         // FHESTR_OVERLAP_STREAMS      streams of throughput mode 2 (2 .. 4)
    int ks_mfma = -1;                 // FHESTR_KS_MFMA              0: byte-plane keyswitch kernel everywhere
    int ks_chunks = -1;               // FHESTR_KS_CHUNKS            K chunks of the matrix-core keyswitch
    int cluster_mode = -2;            // FHESTR_CLUSTER              fhe_engine_set_cluster_mode at creation (-1 .. 2)
    int cluster_spin_limit = -1;      // FHESTR_CLUSTER_SPIN_LIMIT   polls before a hand-over wait gives up
    int multibit_combine_max = -1;    // FHESTR_MULTIBIT_COMBINE_MAX fhe_engine_set_multibit_combine_max at creation
    int cluster_test_fault = -1;      // FHESTR_CLUSTER_TEST_FAULT   honoured by the -DFHESTR_TEST_HOOKS build only
    int dense_per_cu = -1;            // FHESTR_DENSE_PER_CU         LWEs per CU beyond which the dense wide kernel runs (0 = never)
    int cluster_fallback = -1;        // FHESTR_CLUSTER_FALLBACK     0: a multi-CU launch that gave up is an error, not re-run
    static EngineEnv read();
};

struct Engine {
    std::recursive_mutex mu;      // taken by every C ABI entry point that touches this engine (c_api.cpp, LOCK_ENGINE)
    fhe_params_t p{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::map<std::vector<uint64_t>, uint32_t> lut_dedup;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // scratch triple (unused slots of the ring below)
    std::vector<hipEvent_t> ring;   // 4 events per recorded ks_pbs call
    size_t ring_used = 0;           // calls recorded since the last reset
    const BrVariant* variant = nullptr;       // layout used up to one LWE per CU
    const BrVariant* variant_large = nullptr; // same Fourier-key layout, used for larger batches (may equal variant)
    int cu_count = 256;
    int pipeline = 0;                         // ks_pbs_dev throughput modes: 1 = keyswitch of call k+1 in the shadow of the blind rotation of call k; 2 = whole calls overlapped on two streams
    hipStream_t ks_stream = nullptr;
    hipEvent_t pipe_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // keyswitch done [slot], blind rotation done [slot], engine stream so far
    hipEvent_t pipe_input_ready = nullptr;    // caller's event the next pipelined keyswitch waits for (one shot)
    const void* pipe_out[2] = {nullptr, nullptr};
    size_t pipe_out_bytes[2] = {0, 0};
    // mode 2 (overlapped batches): calls rotate over ovl_streams streams (slot 0 = the engine stream, 1 = ks_stream)
    static constexpr int OVL_MAX = 4;
    int ovl_streams = 2;
    uint32_t wide_fair_shift = 13;           // two-LWEs-per-CU kernel: log2 ticks (100 MHz) of the priority time slice, 0 = off (FHESTR_WIDE_FAIR)
    uint32_t dense_per_cu = 2;               // N = 1024, k = 2: the four-workgroups-per-CU kernel beyond this many LWEs per CU (0 = never)
    bool keep_busy = false;                  // small launches carry replicas on the idle CUs (fhe_engine_set_keep_busy, FHESTR_KEEP_BUSY)
    hipStream_t ovl_stream[OVL_MAX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ovl_done[OVL_MAX] = {nullptr, nullptr, nullptr, nullptr};
    // byte ranges every call of the current run read / wrote, per stream (a later call on ANOTHER stream that touches one of
    // them waits for that stream's latest ovl_done: stream order then covers all of its earlier calls -- ADVICE r3)
    struct ByteRange { const char* lo; const char* hi; };
    std::vector<ByteRange> ovl_ins[OVL_MAX], ovl_outs[OVL_MAX];
    uint64_t* ovl_small[OVL_MAX] = {nullptr, nullptr, nullptr, nullptr};      // small-ciphertext buffer per stream
    size_t ovl_cap_small[OVL_MAX] = {0, 0, 0, 0};
    int8_t* ovl_digits[OVL_MAX] = {nullptr, nullptr, nullptr, nullptr};       // keyswitch digit fragments per stream (slot 0: d_ks_digits)
    size_t ovl_cap_digits[OVL_MAX] = {0, 0, 0, 0};
    int sync_all_streams();
    void end_pipeline_run();
    uint64_t pipe_calls = 0;
    int shadow_fit = -1;                      // -1 unknown, else whether a 64-VGPR wave fits next to the rotation's
    uint32_t multibit_combine_max = 64;       // multi-bit PBS: batches up to this size prepare their GGSWs on the whole GPU first

    // resident keys / tables
    uint32_t* d_ksk_packed = nullptr; // [rows/4][8][n+1] byte planes for keyswitch_dot4_kernel
    uint64_t* d_ksk_rowsum = nullptr; // [kN / KS_IC][n+1] sums of every tile's key rows (bias removal)
    int8_t* d_ksk_mfma = nullptr;     // balanced base-256 digit planes of the KSK in MFMA B-fragment order (ks_mfma_kernels.hip.h)
    int8_t* d_ks_digits = nullptr;    // A fragments: signed decomposition digits of the batch being keyswitched
    size_t cap_ks_digits = 0;
    bool ks_mfma_enabled = true;      // FHESTR_KS_MFMA=0: byte-plane dot4 kernel everywhere
    double* d_fbsk = nullptr;
    double* d_fbsk_dense = nullptr;          // the same key in the dense wide kernel's Fourier order (N = 1024, k = 2 only)
    uint64_t* d_luts = nullptr;
    size_t luts_cap = 0;
    uint32_t n_luts = 0;

    // staging / scratch (grown on demand)
    uint64_t *d_in = nullptr, *d_small = nullptr, *d_small2 = nullptr, *d_out = nullptr, *d_pool = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t* d_slot_exp = nullptr;   // multi-bit two-kernel path: exponent of w = e^{i pi / N} each Fourier slot evaluates at
    size_t multibit_workspace_cap = 0;   // bytes of prepared GGSWs kept at once, larger batches run in sub-batches (0 = from free memory)
    void* d_meta = nullptr;
    void* d_ws = nullptr;      // per-LWE HBM workspace of the large-N blind rotation
    size_t cap_ws = 0;
    void* d_cluster_ws = nullptr;   // cluster kernel: 1.5 MB of exchange matrices per cluster (L2-resident by design)
    size_t cap_cluster_ws = 0;
    void* d_cluster_ctl = nullptr;  // ClusterCtl (tickets, flags; zeroed per launch) + ClusterStatus (sticky)
    bool cluster_unchecked = false; // a cluster launch whose status words have not been read yet
    uint32_t cluster_last = 0;      // clusters the last checked launch formed
    bool cluster_fallback = true;   // a multi-CU launch that gave up is re-run on the one-workgroup kernel (cluster_settle)
    uint32_t cluster_fallbacks = 0; // how often that happened
    uint32_t cluster_last_error = 0;
    int cluster_mode = -1;          // -1 automatic (by batch size), 0 never, 1 always (FHESTR_CLUSTER)
    uint32_t cluster_max_batch = 0xFFFFFFFFu;
    uint32_t ks_chunks_override = 0; // FHESTR_KS_CHUNKS: K chunks of the matrix-core keyswitch (0 = automatic)
    int xcd_per_cu = -1;             // workgroups of the whole-XCD kernel a CU holds (occupancy query, cached)
    uint32_t xcd_auto_max = 16;     // automatic mode: batches up to this size take the whole-XCD kernel (two LWEs per XCD in flight)
    uint32_t cluster_spin_limit = 1u << 22;   // polls before a hand-over wait gives up (FHESTR_CLUSTER_SPIN_LIMIT)
    uint32_t cluster_test_fault = 0;          // tests only (FHESTR_CLUSTER_TEST_FAULT): epoch one workgroup stays silent at
    int cluster_check();            // after a synchronisation: did a cluster launch give up on a hand-over?
    size_t cap_in = 0, cap_small = 0, cap_small2 = 0, cap_out = 0, cap_idx = 0, cap_pool = 0, cap_meta = 0;

    static int create(const fhe_params_t& p, int device, Engine** out);
    ~Engine();
    int use();
    int set_variant(int logR);
    int load_keys(const uint64_t* bsk_std, const uint64_t* ksk);
    int load_seeded_keys(const uint8_t ksk_seed[16], const uint64_t* ksk_bodies, const uint8_t bsk_seed[16], const uint64_t* bsk_bodies,
                         uint64_t* bsk_std_out, uint64_t* ksk_out);
    int expand_seeded_lwe(const uint8_t* seeds, const uint64_t* bodies, uint32_t count, uint64_t* d_out, uint64_t* host_out);
    int generate_keys(const uint64_t* glwe_sk, const uint64_t* small_sk, const uint8_t seed[32],
                      uint64_t* bsk_std_out, uint64_t* ksk_out);
    int install_keys(uint64_t* d_ksk_std, uint64_t* d_bsk_std);
    int convert_polys(const uint64_t* d_std, double* d_out, uint32_t n_polys);
    int cluster_settle(const uint64_t* d_sm, const uint32_t* d_lut_idx, uint64_t* d_big, uint32_t count);
    int probe_slot_exponents();
    uint64_t fill_accumulator(const uint64_t* table, std::vector<uint64_t>& acc) const { return fhe::fill_accumulator(p, table, acc); }
    int lut_upload_dedup(const std::vector<uint64_t>& acc, uint32_t* id);   // same contents -> same id
    int set_stream(hipStream_t s, bool use_own);   // launch on a caller-owned stream (e.g. the framework's current stream)
    int lut_upload(const uint64_t* acc, uint32_t* id);
    int lut_download(uint32_t id, uint64_t* acc);
    int ensure_batch(uint32_t count);
    int check_lut_idx(const uint32_t* lut_idx, uint32_t count) const;

    int launch_keyswitch(const uint64_t* d_big, uint64_t* d_sm, uint32_t count, hipStream_t on = nullptr, bool shadow = false, int digits_slot = 0);
    int launch_blind_rotate(const uint64_t* d_sm, const uint32_t* d_lut_idx, uint64_t* d_big, uint32_t count, hipStream_t on = nullptr, bool two_per_cu = false);
    bool shadow_keyswitch_fits();
    int ks_pbs_dev(const uint64_t* d_big_in, const uint32_t* d_lut_idx, uint64_t* d_big_out, uint32_t count, bool allow_pipeline = false);
    int ks_pbs_host(const uint64_t* in, const uint32_t* lut_idx, uint64_t* out, uint32_t count);
    int keyswitch_host(const uint64_t* in, uint64_t* out_small, uint32_t count);
    int pbs_host(const uint64_t* in_small, const uint32_t* lut_idx, uint64_t* out, uint32_t count);
    int pbs_ks_host(const uint64_t* in_small, const uint32_t* lut_idx, uint64_t* out_small, uint32_t count);
    int lincomb_dev(const uint64_t* d_pool, const uint32_t* d_off, const uint32_t* d_src,
                    const int32_t* d_coeff, const uint64_t* d_cst, uint64_t* d_out, uint32_t jobs);
    // `instances` copies of a plan level at once (lincomb_batch_kernel); strides in ciphertext rows
    int lincomb_batch_dev(const uint64_t* d_pool, const uint32_t* d_off, const uint32_t* d_src, const int32_t* d_coeff,
                          const uint64_t* d_cst, uint64_t* d_out, uint32_t jobs, uint32_t instances, uint32_t src_slot, uint32_t src_inst,
                          uint32_t out_job, uint32_t out_inst, const uint32_t* d_lut_in, uint32_t* d_lut_out);
    int restride_dev(const uint64_t* d_in, uint64_t* d_out, uint32_t slots, uint32_t instances, uint32_t in_slot, uint32_t in_inst,
                     uint32_t out_slot, uint32_t out_inst);
    int lincomb_host(const uint64_t* pool, uint32_t pool_count, const uint32_t* off, const uint32_t* src,
                     const int32_t* coeff, const uint64_t* cst, uint64_t* out, uint32_t jobs);
    int last_kernel_ms(float ms[2]);
    int kernel_times(double total_ms[2], uint32_t* calls, bool reset);
    int synchronize();
};

int params_supported(const fhe_params_t& p);   // 0, or 1 with the reason in fhe_last_error (no device needed)

}  // namespace fhe

struct fhe_plan;
struct fhe_engine {
    fhe::Engine* impl;
    // plans of the one-call string operations (fhe_str_eq ...), most recently used first: building a
    // plan and allocating its pool costs milliseconds, the same (op, capacities, clear pattern) recurs
    std::vector<std::pair<std::string, fhe_plan*>> str_plans;
};
