// c_api.cpp -- extern "C" boundary (include/fhestr.h).  Mirrors the reference C API conventions:
// every entry point catches everything and returns 0/1 (c_api/utils.rs:3-12), out-pointers are
// nulled first so unchecked failures are loud (c_api/shortint/server_key/pbs.rs:24-29).
#include <exception>
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

extern "C" {

const char* fhe_last_error(void) { return fhe::g_last_error.c_str(); }

int fhe_engine_create(const fhe_params_t* params, int device, fhe_engine** out) {
    API_BEGIN
    CHECK_PTR(out);
    *out = nullptr;
    CHECK_PTR(params);
    fhe::Engine* e = nullptr;
    if (fhe::Engine::create(*params, device, &e)) return 1;
    *out = new fhe_engine{e};
    return 0;
    API_END
}

int fhe_engine_destroy(fhe_engine* eng) {
    API_BEGIN
    if (!eng) return 0;
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
    CHECK_PTR(eng); CHECK_PTR(bsk_std); CHECK_PTR(ksk);
    return eng->impl->load_keys(bsk_std, ksk);
    API_END
}

void* fhe_engine_stream(fhe_engine* eng) { return eng ? (void*)eng->impl->stream : nullptr; }

int fhe_engine_synchronize(fhe_engine* eng) {
    API_BEGIN
    CHECK_PTR(eng);
    return eng->impl->synchronize();
    API_END
}

int fhe_engine_set_variant(fhe_engine* eng, int log2_points) {
    API_BEGIN
    CHECK_PTR(eng);
    return eng->impl->set_variant(log2_points);
    API_END
}

int fhe_lut_generate(fhe_engine* eng, const uint64_t* table, uint32_t* lut_id, uint64_t* degree) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(table); CHECK_PTR(lut_id);
    std::vector<uint64_t> acc;
    uint64_t deg = eng->impl->fill_accumulator(table, acc);
    if (degree) *degree = deg;
    return eng->impl->lut_upload(acc.data(), lut_id);
    API_END
}

int fhe_lut_upload(fhe_engine* eng, const uint64_t* accumulator, uint32_t* lut_id) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(accumulator); CHECK_PTR(lut_id);
    return eng->impl->lut_upload(accumulator, lut_id);
    API_END
}

int fhe_lut_download(fhe_engine* eng, uint32_t lut_id, uint64_t* accumulator) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(accumulator);
    return eng->impl->lut_download(lut_id, accumulator);
    API_END
}

int fhe_lut_count(const fhe_engine* eng, uint32_t* count) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(count);
    *count = eng->impl->n_luts;
    return 0;
    API_END
}

int fhe_keyswitch_batch(fhe_engine* eng, const uint64_t* in, uint64_t* out, uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->keyswitch_host(in, out, count);
    API_END
}

int fhe_pbs_batch(fhe_engine* eng, const uint64_t* in, const uint32_t* lut_idx, uint64_t* out,
                  uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->pbs_host(in, lut_idx, out, count);
    API_END
}

int fhe_ks_pbs_batch(fhe_engine* eng, const uint64_t* in, const uint32_t* lut_idx, uint64_t* out,
                     uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(in); CHECK_PTR(out);
    return eng->impl->ks_pbs_host(in, lut_idx, out, count);
    API_END
}

int fhe_ks_pbs_batch_dev(fhe_engine* eng, const uint64_t* d_in, const uint32_t* d_lut_idx,
                         uint64_t* d_out, uint32_t count) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(d_in); CHECK_PTR(d_out);
    return eng->impl->ks_pbs_dev(d_in, d_lut_idx, d_out, count);
    API_END
}

int fhe_lwe_lincomb_batch(fhe_engine* eng, const uint64_t* pool, uint32_t pool_count,
                          const uint32_t* off, const uint32_t* src, const int32_t* coeff,
                          const uint64_t* cst, uint64_t* out, uint32_t jobs) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(pool); CHECK_PTR(off); CHECK_PTR(src); CHECK_PTR(coeff);
    CHECK_PTR(cst); CHECK_PTR(out);
    return eng->impl->lincomb_host(pool, pool_count, off, src, coeff, cst, out, jobs);
    API_END
}

int fhe_last_kernel_ms(fhe_engine* eng, float ms[2]) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(ms);
    return eng->impl->last_kernel_ms(ms);
    API_END
}

int fhe_kernel_times(fhe_engine* eng, double total_ms[2], uint32_t* calls, int reset) {
    API_BEGIN
    CHECK_PTR(eng); CHECK_PTR(total_ms); CHECK_PTR(calls);
    return eng->impl->kernel_times(total_ms, calls, reset != 0);
    API_END
}

}  // extern "C"
