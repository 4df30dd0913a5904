// wire_format.cpp -- tfhe-rs 0.5 serialized forms of the objects either side of the hot path, so that a
// real tfhe-rs client can hand ciphertexts and server keys to the engine and read results back.
//
// The reference serializes with serde + bincode 1.x: `bincode::serialize` (legacy config) and
// `safe_serialize` (DefaultOptions + with_fixint_encoding, safe_deserialization.rs:30-99) agree on these
// types: little endian, fixed-width integers, usize as u64, u128 as 16 bytes, Vec<T> / String as a u64
// length then the elements, structs as their fields in declaration order, a unit enum variant as its u32
// index.  Field orders are those of the struct definitions:
//   LweCiphertext<Vec<u64>>      { data, ciphertext_modulus }                          entities/lwe_ciphertext.rs:500-507
//   LweKeyswitchKey<Vec<u64>>    { data, decomp_base_log, decomp_level_count, output_lwe_size,
//                                  ciphertext_modulus }                                entities/lwe_keyswitch_key.rs:76-86
//   LweBootstrapKey<Vec<u64>>    { ggsw_list: GgswCiphertextList { data, glwe_size, polynomial_size,
//                                  decomp_base_log, decomp_level_count, ciphertext_modulus } }
//                                                     entities/lwe_bootstrap_key.rs:98-106, ggsw_ciphertext_list.rs:9-20
//   CiphertextModulus<u64>       { modulus: u128 (0 = native 2^64), scalar_bits: usize }
//                                                     commons/ciphertext_modulus.rs:41-64
//   shortint::Ciphertext         { ct, degree, noise_level, message_modulus, carry_modulus, pbs_order }
//                                                     shortint/ciphertext/mod.rs:261-270 (all usize newtypes; PBSOrder:
//                                                     commons/parameters.rs:233-245)
//   safe_serialize framing       String "0.1", String T::NAME ("shortint::Ciphertext"), then the object, each
//                                under its own size limit                              safe_deserialization.rs:16-54
// The data layouts inside `data` are the flat layouts the C ABI already uses (fhestr.h).
// The reference ships no serialized fixture and cannot be run here: byte-level PARITY IS UNPINNED; the
// tests pin the layout against the bincode rules above on hand-built examples and round trips.
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace {

using fhe::fail;

struct Writer {
    uint8_t* out;
    size_t cap, pos = 0;
    bool overflow = false;
    void bytes(const void* p, size_t n) {
        if (out) {
            if (pos + n > cap) overflow = true;
            else std::memcpy(out + pos, p, n);
        }
        pos += n;
    }
    void u32(uint32_t v) { uint8_t b[4]; for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * i)); bytes(b, 4); }
    void u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); bytes(b, 8); }
    void vec_u64(const uint64_t* v, size_t n) {
        u64(n);
        if (out && pos + n * 8 <= cap) {            // little-endian host: one copy
            std::memcpy(out + pos, v, n * 8);
            pos += n * 8;
        } else {
            if (out) overflow = true;
            pos += n * 8;
        }
    }
    void str(const char* s) { const size_t n = std::strlen(s); u64(n); bytes(s, n); }
    void native_modulus_u64() { u64(0); u64(0); u64(64); }   // u128 0 = native, scalar_bits = 64
};

struct Reader {
    const uint8_t* in;
    size_t len, pos = 0;
    std::string err;
    bool need(size_t n) {
        if (!err.empty()) return false;
        if (n > len - pos) { err = "truncated input"; return false; }
        return true;
    }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)in[pos + i] << (8 * i); pos += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)in[pos + i] << (8 * i); pos += 8; return v; }
    // Vec<u64> of exactly / at most `max_words` into dst; returns the element count
    size_t vec_u64(uint64_t* dst, size_t max_words) {
        const uint64_t n = u64();
        if (!err.empty()) return 0;
        if (n > max_words) { err = "vector longer than the destination (" + std::to_string(n) + " words)"; return 0; }
        if (n > (len - pos) / 8) { err = "truncated input"; return 0; }
        std::memcpy(dst, in + pos, (size_t)n * 8);
        pos += (size_t)n * 8;
        return (size_t)n;
    }
    std::string str(uint64_t limit) {
        const uint64_t n = u64();
        if (!err.empty()) return "";
        if (n > limit) { err = "string longer than its size limit"; return ""; }
        if (!need((size_t)n)) return "";
        std::string s(reinterpret_cast<const char*>(in + pos), (size_t)n);
        pos += (size_t)n;
        return s;
    }
    void native_modulus_u64() {
        const uint64_t lo = u64(), hi = u64(), bits = u64();
        if (!err.empty()) return;
        if (bits != 64) err = "expected an unsigned integer with 64 bits, got " + std::to_string(bits);   // ciphertext_modulus.rs:74-80
        else if (lo != 0 || hi != 0) err = "only the native modulus 2^64 is supported";
    }
};

int finish(Writer& w, size_t* written) {
    if (written) *written = w.pos;
    if (w.overflow) return fail("output buffer too small: " + std::to_string(w.pos) + " bytes needed");
    return 0;
}

constexpr const char* kVersion = "0.1";                     // safe_deserialization.rs:16
constexpr const char* kShortintName = "shortint::Ciphertext";   // shortint/ciphertext/mod.rs:272-274

}  // namespace

extern "C" {

int fhe_wire_write_lwe_ciphertext(const uint64_t* ct, size_t lwe_size, uint8_t* out, size_t out_cap, size_t* written) {
    if (!ct || lwe_size == 0) return fail("null / empty ciphertext");
    Writer w{out, out_cap};
    w.vec_u64(ct, lwe_size);
    w.native_modulus_u64();
    return finish(w, written);
}

int fhe_wire_read_lwe_ciphertext(const uint8_t* in, size_t in_len, uint64_t* ct, size_t ct_cap, size_t* lwe_size,
                                 size_t* consumed) {
    if (!in || !ct) return fail("null pointer");
    Reader r{in, in_len};
    const size_t n = r.vec_u64(ct, ct_cap);
    r.native_modulus_u64();
    if (!r.err.empty()) return fail("LweCiphertext: " + r.err);
    if (n == 0) return fail("LweCiphertext: empty container");
    if (lwe_size) *lwe_size = n;
    if (consumed) *consumed = r.pos;
    return 0;
}

int fhe_wire_write_keyswitch_key(const fhe_params_t* p, const uint64_t* ksk, uint8_t* out, size_t out_cap, size_t* written) {
    if (!p || !ksk) return fail("null pointer");
    Writer w{out, out_cap};
    w.vec_u64(ksk, (size_t)p->k * p->N * p->ks_level * (p->n + 1));
    w.u64(p->ks_base_log);
    w.u64(p->ks_level);
    w.u64((uint64_t)p->n + 1);
    w.native_modulus_u64();
    return finish(w, written);
}

int fhe_wire_read_keyswitch_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint64_t* ksk, size_t* consumed) {
    if (!p || !in || !ksk) return fail("null pointer");
    const size_t want = (size_t)p->k * p->N * p->ks_level * (p->n + 1);
    Reader r{in, in_len};
    const size_t n = r.vec_u64(ksk, want);
    const uint64_t base_log = r.u64(), level = r.u64(), out_size = r.u64();
    r.native_modulus_u64();
    if (!r.err.empty()) return fail("LweKeyswitchKey: " + r.err);
    // ParameterSetConformant-style checks (conformance.rs): every dimension must be the parameter set's
    if (base_log != p->ks_base_log || level != p->ks_level || out_size != (uint64_t)p->n + 1 || n != want)
        return fail("LweKeyswitchKey does not match the parameter set (base_log " + std::to_string(base_log) + ", level " +
                    std::to_string(level) + ", output size " + std::to_string(out_size) + ", " + std::to_string(n) + " words)");
    if (consumed) *consumed = r.pos;
    return 0;
}

int fhe_wire_write_bootstrap_key(const fhe_params_t* p, const uint64_t* bsk_std, uint8_t* out, size_t out_cap, size_t* written) {
    if (!p || !bsk_std) return fail("null pointer");
    if (p->grouping_factor > 1) return fail("multi-bit bootstrap keys have their own container: use fhe_wire_{read,write}_multi_bit_bootstrap_key");
    Writer w{out, out_cap};
    w.vec_u64(bsk_std, (size_t)p->n * p->pbs_level * (p->k + 1) * (p->k + 1) * p->N);
    w.u64((uint64_t)p->k + 1);
    w.u64(p->N);
    w.u64(p->pbs_base_log);
    w.u64(p->pbs_level);
    w.native_modulus_u64();
    return finish(w, written);
}

int fhe_wire_read_bootstrap_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint64_t* bsk_std, size_t* consumed) {
    if (!p || !in || !bsk_std) return fail("null pointer");
    if (p->grouping_factor > 1) return fail("multi-bit bootstrap keys have their own container: use fhe_wire_{read,write}_multi_bit_bootstrap_key");
    const size_t want = (size_t)p->n * p->pbs_level * (p->k + 1) * (p->k + 1) * p->N;
    Reader r{in, in_len};
    const size_t n = r.vec_u64(bsk_std, want);
    const uint64_t glwe_size = r.u64(), poly = r.u64(), base_log = r.u64(), level = r.u64();
    r.native_modulus_u64();
    if (!r.err.empty()) return fail("LweBootstrapKey: " + r.err);
    if (glwe_size != (uint64_t)p->k + 1 || poly != p->N || base_log != p->pbs_base_log || level != p->pbs_level || n != want)
        return fail("LweBootstrapKey does not match the parameter set (glwe_size " + std::to_string(glwe_size) + ", N " +
                    std::to_string(poly) + ", base_log " + std::to_string(base_log) + ", level " + std::to_string(level) + ", " +
                    std::to_string(n) + " words)");
    if (consumed) *consumed = r.pos;
    return 0;
}

int fhe_wire_write_shortint_ciphertext(const uint64_t* ct, size_t lwe_size, const fhe_shortint_meta* meta, int safe_framing,
                                       uint8_t* out, size_t out_cap, size_t* written) {
    if (!ct || !meta || lwe_size == 0) return fail("null / empty ciphertext");
    if (meta->pbs_order > 1) return fail("pbs_order must be 0 (KeyswitchBootstrap) or 1 (BootstrapKeyswitch)");
    Writer w{out, out_cap};
    if (safe_framing) {
        w.str(kVersion);
        w.str(kShortintName);
    }
    w.vec_u64(ct, lwe_size);
    w.native_modulus_u64();
    w.u64(meta->degree);
    w.u64(meta->noise_level);
    w.u64(meta->message_modulus);
    w.u64(meta->carry_modulus);
    w.u32(meta->pbs_order);
    return finish(w, written);
}

int fhe_wire_read_shortint_ciphertext(const uint8_t* in, size_t in_len, int safe_framing, uint64_t size_limit, uint64_t* ct,
                                      size_t ct_cap, size_t* lwe_size, fhe_shortint_meta* meta, size_t* consumed) {
    if (!in || !ct || !meta) return fail("null pointer");
    Reader r{in, in_len};
    if (safe_framing) {
        const std::string version = r.str(100);          // VERSION_LENGTH_LIMIT
        if (r.err.empty() && version != kVersion)
            return fail("On deserialization, expected serialization version 0.1, got version " + version);
        const std::string name = r.str(1000);            // TYPE_NAME_LENGTH_LIMIT
        if (r.err.empty() && name != kShortintName)
            return fail("On deserialization, expected type shortint::Ciphertext, got type " + name);
    }
    const size_t body = r.pos;
    const size_t n = r.vec_u64(ct, ct_cap);
    r.native_modulus_u64();
    meta->degree = r.u64();
    meta->noise_level = r.u64();
    meta->message_modulus = r.u64();
    meta->carry_modulus = r.u64();
    meta->pbs_order = r.u32();
    if (!r.err.empty()) return fail("shortint::Ciphertext: " + r.err);
    if (size_limit && r.pos - body > size_limit) return fail("shortint::Ciphertext: serialized object exceeds the size limit");
    if (meta->pbs_order > 1) return fail("shortint::Ciphertext: unknown PBSOrder variant " + std::to_string(meta->pbs_order));
    if (n == 0) return fail("shortint::Ciphertext: empty container");
    if (lwe_size) *lwe_size = n;
    if (consumed) *consumed = r.pos;
    return 0;
}

}  // extern "C"
