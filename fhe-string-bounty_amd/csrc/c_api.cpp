#include <algorithm>
// c_api.cpp -- extern "C" boundary (include/fhestr.h).  Mirrors the reference C API conventions:
// every entry point catches everything and returns 0/1 (c_api/utils.rs:3-12), out-pointers are
// nulled first so unchecked failures are loud (c_api/shortint/server_key/pbs.rs:24-29).
#include <exception>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"

using fhe::fail;

#define API_BEGIN try {
#define API_END                                                   \
    }                                                             \
    catch (const std::exception& e) { return fail(e.what()); }    \
    catch (...) { return fail("unknown exception"); }

#define CHECK_PTR(p) \
    if (!(p)) return fail("null pointer: " #p)
// Calls on one engine from several host threads are serialised here (the reference's ServerKey is Sync: every thread
// brings its own scratch through a thread-local ShortintEngine, shortint/engine/mod.rs:23-25,184-189; this engine owns one
// set of staging buffers and one stream, so it takes a lock instead).  Recursive: fhe_str_* build plans through the
// same entry points.
#define LOCK_ENGINE(e) std::lock_guard<std::recursive_mutex> _engine_lock((e)->impl->mu)
#define LOCK_PLAN(p) \
    std::unique_lock<std::recursive_mutex> _plan_lock; \
    if ((p)->c->engine()) _plan_lock = std::unique_lock<std::recursive_mutex>((p)->c->engine()->mu)

extern "C" {

const char* fhe_last_error(void) { return fhe::g_last_error.c_str(); }
const char* fhe_kernel_revision(void) { return FHESTR_KERNEL_REVISION; }

int fhe_engine_create(const fhe_params_t* params, int device, fhe_engine** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(params);
    fhe::Engine* e = nullptr;
    if (fhe::Engine::create(*params, device, &e)) return 1;
    *out = new fhe_engine{e, {}};
    return 0;
    API_END
}

int fhe_engine_destroy(fhe_engine* eng) {
    API_BEGIN
    if (!eng) return 0;
    for (auto& kv : eng->str_plans) fhe_plan_destroy(kv.second);
    delete eng->impl;
    delete eng;
    return 0;
    API_END
}

int fhe_engine_params(const fhe_engine* eng, fhe_params_t* out) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(out);
    *out = eng->impl->p;
    return 0;
    API_END
}

int fhe_engine_load_keys(fhe_engine* eng, const uint64_t* bsk_std, const uint64_t* ksk) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(bsk_std); CHECK_PTR(ksk);
    return eng->impl->load_keys(bsk_std, ksk);
    API_END
}

int fhe_engine_generate_keys(fhe_engine* eng, const uint64_t* glwe_sk, const uint64_t* small_sk, const uint8_t seed[32],
                             uint64_t* bsk_std_out, uint64_t* ksk_out) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(glwe_sk); CHECK_PTR(small_sk); CHECK_PTR(seed);
    return eng->impl->generate_keys(glwe_sk, small_sk, seed, bsk_std_out, ksk_out);
    API_END
}

void* fhe_engine_stream(fhe_engine* eng) { return eng ? (void*)eng->impl->stream : nullptr; }

int fhe_engine_synchronize(fhe_engine* eng) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    return eng->impl->synchronize();
    API_END
}

int fhe_engine_set_variant(fhe_engine* eng, int log2_points) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    return eng->impl->set_variant(log2_points);
    API_END
}

int fhe_engine_load_seeded_keys(fhe_engine* eng, const uint8_t ksk_seed[16], const uint64_t* ksk_bodies, const uint8_t bsk_seed[16],
                                const uint64_t* bsk_bodies, uint64_t* bsk_std_out, uint64_t* ksk_out) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(ksk_seed); CHECK_PTR(ksk_bodies); CHECK_PTR(bsk_seed); CHECK_PTR(bsk_bodies);
    return eng->impl->load_seeded_keys(ksk_seed, ksk_bodies, bsk_seed, bsk_bodies, bsk_std_out, ksk_out);
    API_END
}

int fhe_engine_expand_seeded_lwe(fhe_engine* eng, const uint8_t* seeds, const uint64_t* bodies, uint32_t count, uint64_t* d_out,
                                 uint64_t* host_out) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    if (count) { CHECK_PTR(seeds); CHECK_PTR(bodies); }
    if (!d_out && !host_out) return fhe::fail("expand_seeded_lwe: no destination");
    return eng->impl->expand_seeded_lwe(seeds, bodies, count, d_out, host_out);
    API_END
}

int fhe_engine_set_pipeline(fhe_engine* eng, int on) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    if (eng->impl->synchronize()) return 1;
    if (on < 0 || on > 2) return fhe::fail("pipeline mode: 0 (off), 1 (keyswitch in the shadow of the previous blind rotation) or 2 (overlapped batches)");
    eng->impl->pipeline = on;
    return 0;
    API_END
}

int fhe_engine_pipeline_input_event(fhe_engine* eng, void* hip_event) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    eng->impl->pipe_input_ready = reinterpret_cast<hipEvent_t>(hip_event);
    return 0;
    API_END
}

int fhe_engine_set_multibit_combine_max(fhe_engine* eng, uint32_t max_batch) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    if (max_batch > 1024) return fhe::fail("multibit_combine_max: at most 1024 (workspace grows by 16 MB per LWE at N = 2048)");
    eng->impl->multibit_combine_max = max_batch;
    return 0;
    API_END
}

int fhe_engine_set_cluster_mode(fhe_engine* eng, int mode, uint32_t max_batch) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    if (mode < -1 || mode > 2) return fhe::fail("cluster mode: -1 (automatic), 0 (never), 1 (always) or 2 (always, the 8-CU clusters of round 3)");
    eng->impl->cluster_mode = mode;
    eng->impl->cluster_max_batch = max_batch;
    return 0;
    API_END
}

int fhe_engine_set_keep_busy(fhe_engine* eng, int on) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    eng->impl->keep_busy = on != 0;
    return 0;
    API_END
}

int fhe_engine_cluster_info(fhe_engine* eng, uint32_t* clusters) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(clusters);
    if (eng->impl->synchronize()) return 1;
    *clusters = eng->impl->cluster_last;
    return 0;
    API_END
}

int fhe_engine_cluster_fallbacks(fhe_engine* eng, uint32_t* count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(count);
    *count = eng->impl->cluster_fallbacks;
    return 0;
    API_END
}

int fhe_lut_generate(fhe_engine* eng, const uint64_t* table, uint32_t* lut_id, uint64_t* degree) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(table); CHECK_PTR(lut_id);
    std::vector<uint64_t> acc;
    uint64_t deg = eng->impl->fill_accumulator(table, acc);
    if (degree) *degree = deg;
    return eng->impl->lut_upload_dedup(acc, lut_id);
    API_END
}

int fhe_lut_upload(fhe_engine* eng, const uint64_t* accumulator, uint32_t* lut_id) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(accumulator); CHECK_PTR(lut_id);
    return eng->impl->lut_upload(accumulator, lut_id);
    API_END
}

int fhe_lut_download(fhe_engine* eng, uint32_t lut_id, uint64_t* accumulator) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(accumulator);
    return eng->impl->lut_download(lut_id, accumulator);
    API_END
}

int fhe_lut_count(const fhe_engine* eng, uint32_t* count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(count);
    *count = eng->impl->n_luts;
    return 0;
    API_END
}

int fhe_keyswitch_batch(fhe_engine* eng, const uint64_t* in, uint64_t* out, uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->keyswitch_host(in, out, count);
    API_END
}

int fhe_pbs_batch(fhe_engine* eng, const uint64_t* in, const uint32_t* lut_idx, uint64_t* out,
                  uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->pbs_host(in, lut_idx, out, count);
    API_END
}

int fhe_pbs_ks_batch(fhe_engine* eng, const uint64_t* in, const uint32_t* lut_idx, uint64_t* out,
                     uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->pbs_ks_host(in, lut_idx, out, count);
    API_END
}

int fhe_ks_pbs_batch(fhe_engine* eng, const uint64_t* in, const uint32_t* lut_idx, uint64_t* out,
                     uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->ks_pbs_host(in, lut_idx, out, count);
    API_END
}

int fhe_ks_pbs_batch_dev(fhe_engine* eng, const uint64_t* d_in, const uint32_t* d_lut_idx,
                         uint64_t* d_out, uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(d_in); CHECK_PTR(d_out);
    return eng->impl->ks_pbs_dev(d_in, d_lut_idx, d_out, count, /*allow_pipeline=*/true);
    API_END
}

int fhe_lwe_lincomb_batch(fhe_engine* eng, const uint64_t* pool, uint32_t pool_count,
                          const uint32_t* off, const uint32_t* src, const int32_t* coeff,
                          const uint64_t* cst, uint64_t* out, uint32_t jobs) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(pool); CHECK_PTR(off); CHECK_PTR(src); CHECK_PTR(coeff);
    CHECK_PTR(cst); CHECK_PTR(out);
    return eng->impl->lincomb_host(pool, pool_count, off, src, coeff, cst, out, jobs);
    API_END
}

int fhe_last_kernel_ms(fhe_engine* eng, float ms[2]) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(ms);
    return eng->impl->last_kernel_ms(ms);
    API_END
}

int fhe_kernel_times(fhe_engine* eng, double total_ms[2], uint32_t* calls, int reset) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(total_ms); CHECK_PTR(calls);
    return eng->impl->kernel_times(total_ms, calls, reset != 0);
    API_END
}

}  // extern "C"

// ---- plans: levelised shortint circuits + FheString operations ----------------------------------
#include "circuit.h"

namespace fhe {
int build_string_op(Circuit& c, const std::string& op, uint32_t a_cap, uint32_t b_cap,
                    const uint8_t* clear, uint32_t clear_len);
int build_integer_op(Circuit& c, const std::string& op, uint32_t n_blocks, uint64_t scalar);
}
#include "noise_model.h"

struct fhe_plan {
    fhe::Circuit* c;
    bool finalized;
};

extern "C" {

int fhe_plan_create(fhe_engine* eng, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    *out = new fhe_plan{new fhe::Circuit(eng->impl->p, eng->impl), false};
    return 0;
    API_END
}

int fhe_plan_create_offline(const fhe_params_t* params, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(params);
    *out = new fhe_plan{new fhe::Circuit(*params, nullptr), false};
    return 0;
    API_END
}

int fhe_engine_set_stream(fhe_engine* eng, void* hip_stream) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    return eng->impl->set_stream((hipStream_t)hip_stream, false);
    API_END
}

int fhe_engine_reset_stream(fhe_engine* eng) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng);
    return eng->impl->set_stream(nullptr, true);
    API_END
}

int fhe_plan_destroy(fhe_plan* p) {
    API_BEGIN
    if (!p) return 0;
    {
        LOCK_PLAN(p);          // the circuit's destructor releases device memory of the engine's GPU
        delete p->c;
    }
    delete p;
    return 0;
    API_END
}

#define PLAN_BUILDING(p)                                   \
    CHECK_PTR(p);                                          \
    LOCK_PLAN(p);                                          \
    if ((p)->finalized) return fail("plan already finalised")
#define PLAN_READY(p)                                      \
    CHECK_PTR(p);                                          \
    LOCK_PLAN(p);                                          \
    if (!(p)->finalized) return fail("plan not finalised")

int fhe_plan_input(fhe_plan* p, uint64_t degree, uint32_t* node) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(node);
    *node = p->c->input(degree);
    return 0;
    API_END
}

int fhe_plan_lut(fhe_plan* p, const uint64_t* table, uint32_t* lut) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(table); CHECK_PTR(lut);
    std::vector<uint64_t> t(table, table + p->c->total_modulus());
    *lut = p->c->lut(t);
    if (p->c->failed()) return fail(p->c->take_error());
    return 0;
    API_END
}

int fhe_plan_lin(fhe_plan* p, const uint32_t* nodes, const int32_t* coeffs, uint32_t n_terms,
                 int64_t constant, uint32_t* node) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(node);
    if (n_terms) { CHECK_PTR(nodes); CHECK_PTR(coeffs); }
    std::vector<fhe::Term> terms;
    for (uint32_t i = 0; i < n_terms; i++) terms.push_back({nodes[i], coeffs[i]});
    *node = p->c->lin(terms, constant);
    if (p->c->failed()) return fail(p->c->take_error());
    return 0;
    API_END
}

int fhe_plan_pbs(fhe_plan* p, uint32_t src, uint32_t lut, uint32_t* node) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(node);
    *node = p->c->pbs(src, lut);
    if (p->c->failed()) return fail(p->c->take_error());
    return 0;
    API_END
}

int fhe_plan_pbs_signed(fhe_plan* p, uint32_t src, uint32_t lut, uint32_t* node) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(node);
    *node = p->c->pbs(src, lut, true);
    if (p->c->failed()) return fail(p->c->take_error());
    return 0;
    API_END
}

int fhe_plan_pbs_full_box(fhe_plan* p, uint32_t src, int all, uint32_t* node) {
    API_BEGIN
    PLAN_BUILDING(p); CHECK_PTR(node);
    *node = p->c->pbs_full_box(src, all != 0);
    if (p->c->failed()) return fail(p->c->take_error());
    return 0;
    API_END
}

int fhe_plan_set_owner_hint(fhe_plan* p, int rank) {
    API_BEGIN
    PLAN_BUILDING(p);
    p->c->set_owner_hint(rank);
    return 0;
    API_END
}

int fhe_plan_output(fhe_plan* p, uint32_t node) {
    API_BEGIN
    PLAN_BUILDING(p);
    p->c->output(node);
    return 0;
    API_END
}

int fhe_plan_finalize(fhe_plan* p, uint32_t world) {
    API_BEGIN
    PLAN_BUILDING(p);
    if (p->c->finalize(world)) return 1;
    p->finalized = true;
    return 0;
    API_END
}

static int str_plan(const fhe_params_t& params, fhe::Engine* eng, const char* op, uint32_t a_cap,
                    uint32_t b_cap, const uint8_t* clear, uint32_t clear_len, uint32_t world, fhe_plan** out) {
    fhe::Circuit* c = new fhe::Circuit(params, eng);
    c->set_build_world(world);
    if (fhe::build_string_op(*c, op, a_cap, b_cap, clear, clear_len) || c->finalize(world)) {
        delete c;
        return 1;
    }
    *out = new fhe_plan{c, true};
    return 0;
}

static int int_plan(const fhe_params_t& params, fhe::Engine* eng, const char* op, uint32_t n_blocks, uint64_t scalar,
                    uint32_t world, fhe_plan** out) {
    fhe::Circuit* c = new fhe::Circuit(params, eng);
    c->set_build_world(world);
    if (fhe::build_integer_op(*c, op, n_blocks, scalar) || c->finalize(world)) {
        delete c;
        return 1;
    }
    *out = new fhe_plan{c, true};
    return 0;
}

int fhe_int_plan_create(fhe_engine* eng, const char* op, uint32_t n_blocks, uint64_t scalar, uint32_t world, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(op);
    return int_plan(eng->impl->p, eng->impl, op, n_blocks, scalar, world, out);
    API_END
}

int fhe_int_plan_create_offline(const fhe_params_t* params, const char* op, uint32_t n_blocks, uint64_t scalar,
                                uint32_t world, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(params); CHECK_PTR(op);
    return int_plan(*params, nullptr, op, n_blocks, scalar, world, out);
    API_END
}

int fhe_str_plan_create_offline(const fhe_params_t* params, const char* op, uint32_t a_cap, uint32_t b_cap,
                                const uint8_t* clear, uint32_t clear_len, uint32_t world, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(params); CHECK_PTR(op);
    return str_plan(*params, nullptr, op, a_cap, b_cap, clear, clear_len, world, out);
    API_END
}

int fhe_plan_lut_count(const fhe_plan* p, uint32_t* count) {
    API_BEGIN
    CHECK_PTR(p); CHECK_PTR(count);
    *count = p->c->n_luts();
    return 0;
    API_END
}

int fhe_plan_export_lut(const fhe_plan* p, uint32_t lut, uint64_t* accumulator) {
    API_BEGIN
    CHECK_PTR(p); CHECK_PTR(accumulator);
    if (lut >= p->c->n_luts()) return fail("bad plan LUT id");
    const auto& acc = p->c->lut_accumulator(lut);
    std::copy(acc.begin(), acc.end(), accumulator);
    return 0;
    API_END
}

int fhe_str_plan_create(fhe_engine* eng, const char* op, uint32_t a_cap, uint32_t b_cap,
                        const uint8_t* clear, uint32_t clear_len, uint32_t world, fhe_plan** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(op);
    return str_plan(eng->impl->p, eng->impl, op, a_cap, b_cap, clear, clear_len, world, out);
    API_END
}

int fhe_plan_info(const fhe_plan* p, uint32_t info[6]) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(info);
    info[0] = p->c->n_inputs(); info[1] = p->c->n_outputs(); info[2] = p->c->n_levels();
    info[3] = p->c->n_pbs(); info[4] = p->c->pool_slots(); info[5] = p->c->world();
    return 0;
    API_END
}

static const fhe::Circuit::Level* plan_level(const fhe_plan* p, uint32_t level) {
    if (level < p->c->n_levels()) return &p->c->level(level);
    if (level == p->c->n_levels()) return &p->c->out_level();
    return nullptr;
}

int fhe_plan_level_info(const fhe_plan* p, uint32_t level, uint32_t info[8]) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(info);
    const auto* lv = plan_level(p, level);
    if (!lv) return fail("bad level");
    info[0] = (uint32_t)lv->off.size() - 1; info[1] = lv->local_base; info[2] = lv->local_size;
    info[3] = lv->e_max; info[4] = lv->recv_base; info[5] = (uint32_t)lv->src.size();
    info[6] = info[7] = 0;
    return 0;
    API_END
}

int fhe_plan_level_rank_info(const fhe_plan* p, uint32_t level, uint32_t rank, uint32_t info[3]) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(info);
    if (level >= p->c->n_levels()) return fail("bad level");
    if (rank >= p->c->world()) return fail("bad rank");
    const auto& lv = p->c->level(level);
    info[0] = lv.rank_off[rank]; info[1] = lv.rank_off[rank + 1]; info[2] = lv.n_export[rank];
    return 0;
    API_END
}

int fhe_plan_noise_info(const fhe_plan* p, double info[4]) {
    API_BEGIN
    CHECK_PTR(p); CHECK_PTR(info);
    const fhe::NoiseModel m = fhe::noise_model(p->c->params());
    info[0] = p->c->max_pbs_input_noise();
    info[1] = p->c->noise_budget();
    info[2] = m.log2_pfail(p->c->max_pbs_input_noise());
    info[3] = (double)p->c->gathered_lwes();
    return 0;
    API_END
}

int fhe_noise_model(const fhe_params_t* params, double out[6]) {
    API_BEGIN
    CHECK_PTR(params); CHECK_PTR(out);
    const fhe::NoiseModel m = fhe::noise_model(*params);
    out[0] = m.v_pbs; out[1] = m.v_ks; out[2] = m.v_ms; out[3] = m.half_box;
    out[4] = fhe::default_noise_budget(*params); out[5] = m.log2_pfail(out[4]);
    return 0;
    API_END
}

int fhe_host_alloc(size_t bytes, void** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    if (bytes == 0) return 0;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return fhe::fail("fhe_host_alloc: hipHostMalloc failed"); }
    return 0;
    API_END
}

int fhe_host_free(void* ptr) {
    API_BEGIN
    if (ptr && hipHostFree(ptr) != hipSuccess) return fhe::fail("fhe_host_free: not a buffer of fhe_host_alloc");
    return 0;
    API_END
}

int fhe_params_supported(const fhe_params_t* params) {
    API_BEGIN
    CHECK_PTR(params);
    return fhe::params_supported(*params);
    API_END
}

int fhe_noise_model_is_calibrated(const fhe_params_t* params) {
    return params && fhe::noise_model_is_calibrated(*params) ? 1 : 0;
}

int fhe_plan_set_noise_budget(fhe_plan* p, double budget) {
    API_BEGIN
    PLAN_BUILDING(p);
    p->c->set_noise_budget(budget);
    return 0;
    API_END
}

int fhe_plan_export_level(const fhe_plan* p, uint32_t level, uint32_t* off, uint32_t* src,
                          int32_t* coeff, uint64_t* cst, uint32_t* lut) {
    API_BEGIN
    PLAN_READY(p);
    const auto* lv = plan_level(p, level);
    if (!lv) return fail("bad level");
    if (off) std::copy(lv->off.begin(), lv->off.end(), off);
    if (src) std::copy(lv->src.begin(), lv->src.end(), src);
    if (coeff) std::copy(lv->coeff.begin(), lv->coeff.end(), coeff);
    if (cst) std::copy(lv->cst.begin(), lv->cst.end(), cst);
    if (lut) std::copy(lv->lut.begin(), lv->lut.end(), lut);
    return 0;
    API_END
}

int fhe_plan_run(fhe_plan* p, const uint64_t* inputs, uint64_t* outputs) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(outputs);
    if (p->c->n_inputs()) CHECK_PTR(inputs);
    return p->c->run_host(inputs, outputs);
    API_END
}

int fhe_plan_run_batch(fhe_plan* p, uint32_t instances, const uint64_t* inputs, uint64_t* outputs) {
    API_BEGIN
    PLAN_READY(p);
    if (instances == 0) return 0;
    CHECK_PTR(outputs);
    if (p->c->n_inputs()) CHECK_PTR(inputs);
    return p->c->run_batch_host(inputs, p->c->n_inputs(), nullptr, outputs, instances);
    API_END
}

int fhe_plan_run_batch_dev(fhe_plan* p, uint32_t instances, const uint64_t* d_inputs, uint64_t* d_outputs) {
    API_BEGIN
    PLAN_READY(p);
    if (instances == 0) return 0;
    CHECK_PTR(d_outputs);
    if (p->c->n_inputs()) CHECK_PTR(d_inputs);
    return p->c->run_batch_dev(d_inputs, d_outputs, instances);
    API_END
}

int fhe_plan_run_level_rank_dev(fhe_plan* p, uint64_t* d_pool, uint32_t level, uint32_t rank) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(d_pool);
    return p->c->run_level_rank(d_pool, level, rank);
    API_END
}

int fhe_plan_gather_outputs_dev(fhe_plan* p, const uint64_t* d_pool, uint64_t* d_out) {
    API_BEGIN
    PLAN_READY(p); CHECK_PTR(d_pool); CHECK_PTR(d_out);
    return p->c->gather_outputs(d_pool, d_out);
    API_END
}

// plan cache of the one-call string operations, most recently used first (at most STR_PLAN_CACHE entries)
static int cached_str_plan(fhe_engine* eng, const std::string& op, uint32_t a_cap, uint32_t b_cap, const uint8_t* clear, uint32_t clear_len,
                           fhe_plan** out) {
    constexpr size_t STR_PLAN_CACHE = 8;
    std::string key = op + "|" + std::to_string(a_cap) + "|" + std::to_string(b_cap) + "|";
    if (clear) key.append(reinterpret_cast<const char*>(clear), clear_len);
    fhe_plan* plan = nullptr;
    auto& cache = eng->str_plans;
    for (size_t i = 0; i < cache.size(); i++)
        if (cache[i].first == key) {
            plan = cache[i].second;
            std::rotate(cache.begin(), cache.begin() + i, cache.begin() + i + 1);
            break;
        }
    if (!plan) {
        if (fhe_str_plan_create(eng, op.c_str(), a_cap, b_cap, clear, clear_len, 1, &plan)) return 1;
        cache.insert(cache.begin(), {key, plan});
        if (cache.size() > STR_PLAN_CACHE) {
            fhe_plan_destroy(cache.back().second);
            cache.pop_back();
        }
    }
    *out = plan;
    return 0;
}

// one-call FheString operations (host buffers): up to three encrypted operands, each `caps[i]` characters
static int str_op_parts(fhe_engine* eng, const std::string& op, const uint64_t* const* operands, const uint32_t* caps,
                        uint32_t n_operands, const uint8_t* clear, uint32_t clear_len, uint64_t* out) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(out);
    uint32_t a_cap = caps[0], b_cap = 0;
    for (uint32_t i = 0; i < n_operands; i++) {
        CHECK_PTR(operands[i]);
        if (i) b_cap += caps[i];
    }
    fhe_plan* plan = nullptr;
    if (cached_str_plan(eng, op, a_cap, b_cap, clear, clear_len, &plan)) return 1;
    const uint32_t bpc = plan->c->n_inputs() / (a_cap + b_cap);      // blocks per character
    uint32_t counts[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n_operands; i++) counts[i] = caps[i] * bpc;
    return plan->c->run_host_parts(operands, counts, n_operands, out);
    API_END
}

static int str_op(fhe_engine* eng, const char* op, const uint64_t* a, uint32_t a_cap, const uint64_t* b,
                  uint32_t b_cap, const uint8_t* clear, uint32_t clear_len, uint64_t* out) {
    if (!a) return fail("null pointer: a");
    const uint64_t* operands[2] = {a, b};
    const uint32_t caps[2] = {a_cap, b_cap};
    return str_op_parts(eng, op, operands, caps, b ? 2 : 1, clear, clear_len, out);
}

// `count` strings against ONE second operand in one pass (Circuit::run_batch_host): level l of all rows is one launch.
int fhe_str_op_many(fhe_engine* eng, const char* op, const uint64_t* rows, uint32_t a_cap, uint32_t count, const uint64_t* b, uint32_t b_cap,
                    const uint8_t* clear, uint32_t clear_len, uint64_t* out, uint32_t* n_outputs) {
    API_BEGIN
    CHECK_PTR(eng); LOCK_ENGINE(eng); CHECK_PTR(op);
    fhe_plan* plan = nullptr;
    if (cached_str_plan(eng, op, a_cap, b_cap, clear, clear_len, &plan)) return 1;
    if (n_outputs) *n_outputs = plan->c->n_outputs();
    if (!out && n_outputs) return 0;                    // query: how many output ciphertexts per row
    CHECK_PTR(rows); CHECK_PTR(out);
    if (b_cap && !b) return fail("null pointer: b");
    if (a_cap + b_cap == 0) return fail("fhe_str_op_many: empty operands");
    const uint32_t bpc = plan->c->n_inputs() / (a_cap + b_cap);
    return plan->c->run_batch_host(rows, a_cap * bpc, b_cap ? b : nullptr, out, count);
    API_END
}

#define STR_BINARY(name)                                                                              \
    int fhe_str_##name(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint64_t* b,         \
                       uint32_t b_cap, uint64_t* out) {                                               \
        if (!b) return fail("null pointer: b");                                                       \
        return str_op(eng, #name, a, a_cap, b, b_cap, nullptr, 0, out);                               \
    }                                                                                                 \
    int fhe_str_##name##_clear(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint8_t* pat, \
                               uint32_t pat_len, uint64_t* out) {                                     \
        return str_op(eng, #name "_clear", a, a_cap, nullptr, 0, pat, pat_len, out);                  \
    }
STR_BINARY(eq)
STR_BINARY(ne)
STR_BINARY(starts_with)
STR_BINARY(ends_with)
STR_BINARY(contains)
STR_BINARY(find)
STR_BINARY(rfind)
STR_BINARY(eq_ignore_case)
STR_BINARY(lt)
STR_BINARY(le)
STR_BINARY(gt)
STR_BINARY(ge)
STR_BINARY(concat)

/* encrypted (zero padded) pattern: out = 1 + a_cap * blocks LWEs, stripped bit first */
int fhe_str_strip_prefix(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint64_t* pat, uint32_t pat_cap, uint64_t* out) {
    if (!pat) return fail("null pointer: pat");
    return str_op(eng, "strip_prefix", a, a_cap, pat, pat_cap, nullptr, 0, out);
}
int fhe_str_strip_suffix(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint64_t* pat, uint32_t pat_cap, uint64_t* out) {
    if (!pat) return fail("null pointer: pat");
    return str_op(eng, "strip_suffix", a, a_cap, pat, pat_cap, nullptr, 0, out);
}
int fhe_str_replace_general(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint64_t* from, uint32_t from_cap,
                            const uint64_t* to, uint32_t to_cap, uint32_t out_cap, uint64_t* out) {
    if (!a) return fail("null pointer: a");
    if (!from || from_cap == 0) return fail("replace: `from` needs a capacity of at least one character");
    if (to_cap && !to) return fail("null pointer: to");
    const std::string op = "replace:" + std::to_string(from_cap) + ":" + std::to_string(out_cap);
    const uint64_t* operands[3] = {a, from, to};
    const uint32_t caps[3] = {a_cap, from_cap, to_cap};
    return str_op_parts(eng, op, operands, caps, to_cap ? 3 : 2, nullptr, 0, out);
}
int fhe_str_replace_clear_general(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint8_t* from, uint32_t from_len,
                                  const uint8_t* to, uint32_t to_len, uint32_t out_cap, uint64_t* out) {
    if ((from_len && !from) || (to_len && !to)) return fail("null pointer: from / to");
    std::vector<uint8_t> both(from, from + from_len);
    both.insert(both.end(), to, to + to_len);
    const std::string op = "replace_clear:" + std::to_string(from_len) + ":" + std::to_string(out_cap);
    return str_op(eng, op.c_str(), a, a_cap, nullptr, 0, both.data(), (uint32_t)both.size(), out);
}

int fhe_str_repeat_clear(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint32_t count, uint64_t* out) {
    if (count == 0 || count > 255) return fail("repeat: count must be in 1..255");
    const uint8_t c = (uint8_t)count;
    return str_op(eng, "repeat_clear", a, a_cap, nullptr, 0, &c, 1, out);
}

int fhe_str_replace(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint64_t* from_to,
                    uint32_t pat_cap, uint64_t* out) {
    if (!from_to) return fail("null pointer: from_to");
    return str_op(eng, "replace", a, a_cap, from_to, 2 * pat_cap, nullptr, 0, out);
}
int fhe_str_replace_clear(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint8_t* from,
                          const uint8_t* to, uint32_t pat_len, uint64_t* out) {
    if (pat_len && (!from || !to)) return fail("null pointer: from / to");
    std::vector<uint8_t> both(from, from + pat_len);
    both.insert(both.end(), to, to + pat_len);
    return str_op(eng, "replace_clear", a, a_cap, nullptr, 0, both.data(), 2 * pat_len, out);
}
int fhe_str_trim_start(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "trim_start", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_trim_end(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "trim_end", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_strip(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "strip", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_len(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "len", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_is_empty(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "is_empty", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_strip_prefix_clear(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint8_t* pat,
                               uint32_t pat_len, uint64_t* out) {
    return str_op(eng, "strip_prefix_clear", a, a_cap, nullptr, 0, pat, pat_len, out);
}
int fhe_str_strip_suffix_clear(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, const uint8_t* pat,
                               uint32_t pat_len, uint64_t* out) {
    return str_op(eng, "strip_suffix_clear", a, a_cap, nullptr, 0, pat, pat_len, out);
}
int fhe_str_to_upper(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "to_upper", a, a_cap, nullptr, 0, nullptr, 0, out);
}
int fhe_str_to_lower(fhe_engine* eng, const uint64_t* a, uint32_t a_cap, uint64_t* out) {
    return str_op(eng, "to_lower", a, a_cap, nullptr, 0, nullptr, 0, out);
}

}  // extern "C"
