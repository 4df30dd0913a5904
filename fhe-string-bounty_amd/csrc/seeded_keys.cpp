// seeded_keys.cpp -- the seeded ("compressed") server keys a tfhe-rs client actually sends
// (shortint/server_key/compressed.rs: CompressedServerKey = SeededLweKeyswitchKey + SeededLweBootstrapKey or
// SeededLweMultiBitBootstrapKey), and the multi-bit bootstrap key container of the uncompressed form.
//
// A seeded ciphertext keeps its body only; the mask is regenerated from a 128-bit compression seed:
//   * byte stream: concrete-csprng's AES-128 counter mode -- key = the seed as 16 native-endian bytes, block a is
//     AES_key(a as a little-endian u128) (generators/implem/soft/block_cipher.rs:14-60), and a generator built
//     without a start index begins at table index SECOND = byte 1 of block 0 (aes_ctr/generic.rs:55-66,
//     index.rs:27-31);
//   * integers: eight consecutive bytes, little endian (commons/math/random/uniform.rs:15-24);
//   * order: MaskRandomGenerator::new(seed) is forked down the container hierarchy with byte counts equal to
//     what each child draws (generators/encryption/mask_random_generator.rs:347-395), so the masks are simply
//     drawn in storage order --
//       keyswitch key   for every input coefficient, every level: n mask words
//                       (seeded_lwe_keyswitch_key_decompression.rs:10-30, seeded_lwe_ciphertext_list_decompression.rs)
//       bootstrap key   for every GGSW, level, row: k polynomials of N mask words
//                       (seeded_lwe_bootstrap_key_decompression.rs, seeded_ggsw_ciphertext_list_decompression.rs:10-55,
//                        seeded_ggsw_ciphertext_decompression.rs:11-60); multi-bit: the same over all
//                       n/g * 2^g GGSWs (seeded_lwe_multi_bit_bootstrap_key_decompression.rs:11-70).
// bincode layouts (field order of the struct definitions, rules as in wire_format.cpp):
//   SeededLweKeyswitchKey        { data, decomp_base_log, decomp_level_count, output_lwe_size, compression_seed: u128,
//                                  ciphertext_modulus }                       entities/seeded_lwe_keyswitch_key.rs:11-21
//   SeededLweBootstrapKey        { ggsw_list: SeededGgswCiphertextList { data, glwe_size, polynomial_size, decomp_base_log,
//                                  decomp_level_count, compression_seed: u128, ciphertext_modulus } }
//                                                                             entities/seeded_ggsw_ciphertext_list.rs:12-23
//   SeededLweMultiBitBootstrapKey{ ggsw_list: SeededGgswCiphertextList, grouping_factor }   entities/seeded_lwe_multi_bit_bootstrap_key.rs:16-25
//   LweMultiBitBootstrapKey      { ggsw_list: GgswCiphertextList, grouping_factor }         entities/lwe_multi_bit_bootstrap_key.rs:11-20
// The AES block function is pinned by the FIPS-197 vector the reference's own test uses (implem/soft/block_cipher.rs:
// 89-91); the stream position, integer packing and draw order follow the files above and are checked here against an
// independent restatement in the tests (tests/test_seeded_keys.py) -- the reference holds no seeded-key fixture, so
// byte-level parity with a real tfhe-rs client IS UNPINNED.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace {

using fhe::fail;

// ---- AES-128 (encryption only; table-free S-box lookups -- this is key expansion of public masks, not secret data)
const uint8_t kSbox[256] = {
    0x63, 0x7c, 0x77, 0x7b, 0xf2, 0x6b, 0x6f, 0xc5, 0x30, 0x01, 0x67, 0x2b, 0xfe, 0xd7, 0xab, 0x76, 0xca, 0x82, 0xc9, 0x7d, 0xfa, 0x59,
    0x47, 0xf0, 0xad, 0xd4, 0xa2, 0xaf, 0x9c, 0xa4, 0x72, 0xc0, 0xb7, 0xfd, 0x93, 0x26, 0x36, 0x3f, 0xf7, 0xcc, 0x34, 0xa5, 0xe5, 0xf1,
    0x71, 0xd8, 0x31, 0x15, 0x04, 0xc7, 0x23, 0xc3, 0x18, 0x96, 0x05, 0x9a, 0x07, 0x12, 0x80, 0xe2, 0xeb, 0x27, 0xb2, 0x75, 0x09, 0x83,
    0x2c, 0x1a, 0x1b, 0x6e, 0x5a, 0xa0, 0x52, 0x3b, 0xd6, 0xb3, 0x29, 0xe3, 0x2f, 0x84, 0x53, 0xd1, 0x00, 0xed, 0x20, 0xfc, 0xb1, 0x5b,
    0x6a, 0xcb, 0xbe, 0x39, 0x4a, 0x4c, 0x58, 0xcf, 0xd0, 0xef, 0xaa, 0xfb, 0x43, 0x4d, 0x33, 0x85, 0x45, 0xf9, 0x02, 0x7f, 0x50, 0x3c,
    0x9f, 0xa8, 0x51, 0xa3, 0x40, 0x8f, 0x92, 0x9d, 0x38, 0xf5, 0xbc, 0xb6, 0xda, 0x21, 0x10, 0xff, 0xf3, 0xd2, 0xcd, 0x0c, 0x13, 0xec,
    0x5f, 0x97, 0x44, 0x17, 0xc4, 0xa7, 0x7e, 0x3d, 0x64, 0x5d, 0x19, 0x73, 0x60, 0x81, 0x4f, 0xdc, 0x22, 0x2a, 0x90, 0x88, 0x46, 0xee,
    0xb8, 0x14, 0xde, 0x5e, 0x0b, 0xdb, 0xe0, 0x32, 0x3a, 0x0a, 0x49, 0x06, 0x24, 0x5c, 0xc2, 0xd3, 0xac, 0x62, 0x91, 0x95, 0xe4, 0x79,
    0xe7, 0xc8, 0x37, 0x6d, 0x8d, 0xd5, 0x4e, 0xa9, 0x6c, 0x56, 0xf4, 0xea, 0x65, 0x7a, 0xae, 0x08, 0xba, 0x78, 0x25, 0x2e, 0x1c, 0xa6,
    0xb4, 0xc6, 0xe8, 0xdd, 0x74, 0x1f, 0x4b, 0xbd, 0x8b, 0x8a, 0x70, 0x3e, 0xb5, 0x66, 0x48, 0x03, 0xf6, 0x0e, 0x61, 0x35, 0x57, 0xb9,
    0x86, 0xc1, 0x1d, 0x9e, 0xe1, 0xf8, 0x98, 0x11, 0x69, 0xd9, 0x8e, 0x94, 0x9b, 0x1e, 0x87, 0xe9, 0xce, 0x55, 0x28, 0xdf, 0x8c, 0xa1,
    0x89, 0x0d, 0xbf, 0xe6, 0x42, 0x68, 0x41, 0x99, 0x2d, 0x0f, 0xb0, 0x54, 0xbb, 0x16};

inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

struct Aes128 {
    uint8_t rk[11][16];
    explicit Aes128(const uint8_t key[16]) {
        std::memcpy(rk[0], key, 16);
        uint8_t rcon = 1;
        for (int r = 1; r <= 10; r++) {
            const uint8_t* prev = rk[r - 1];
            uint8_t t[4] = {kSbox[prev[13]], kSbox[prev[14]], kSbox[prev[15]], kSbox[prev[12]]};   // RotWord + SubWord
            t[0] ^= rcon;
            rcon = xtime(rcon);
            for (int c = 0; c < 4; c++) {
                for (int b = 0; b < 4; b++) {
                    const uint8_t left = c == 0 ? t[b] : rk[r][4 * (c - 1) + b];
                    rk[r][4 * c + b] = prev[4 * c + b] ^ left;
                }
            }
        }
    }
    void encrypt(const uint8_t in[16], uint8_t out[16]) const {
        uint8_t s[16];
        for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[0][i];
        for (int r = 1; r <= 10; r++) {
            uint8_t t[16];
            for (int c = 0; c < 4; c++)            // SubBytes + ShiftRows (state is column major: s[4c + row])
                for (int row = 0; row < 4; row++) t[4 * c + row] = kSbox[s[4 * ((c + row) & 3) + row]];
            if (r < 10) {
                for (int c = 0; c < 4; c++) {      // MixColumns
                    const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                    s[4 * c] = (uint8_t)(xtime(a0) ^ (xtime(a1) ^ a1) ^ a2 ^ a3);
                    s[4 * c + 1] = (uint8_t)(a0 ^ xtime(a1) ^ (xtime(a2) ^ a2) ^ a3);
                    s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xtime(a2) ^ (xtime(a3) ^ a3));
                    s[4 * c + 3] = (uint8_t)((xtime(a0) ^ a0) ^ a1 ^ a2 ^ xtime(a3));
                }
            } else {
                std::memcpy(s, t, 16);
            }
            for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
        }
        std::memcpy(out, s, 16);
    }
};

// the mask generator's byte stream from its first byte on (table index SECOND)
struct MaskStream {
    Aes128 aes;
    uint64_t block_lo = 0, block_hi = 0;     // 128-bit block counter
    uint8_t buf[16];
    int pos;
    explicit MaskStream(const uint8_t seed[16]) : aes(seed) {
        refill();
        pos = 1;
    }
    void refill() {
        uint8_t ctr[16];
        for (int i = 0; i < 8; i++) { ctr[i] = (uint8_t)(block_lo >> (8 * i)); ctr[8 + i] = (uint8_t)(block_hi >> (8 * i)); }
        aes.encrypt(ctr, buf);
        if (++block_lo == 0) ++block_hi;
        pos = 0;
    }
    uint8_t byte() {
        if (pos == 16) refill();
        return buf[pos++];
    }
    uint64_t word() {
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)byte() << (8 * i);
        return v;
    }
    void words(uint64_t* dst, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = word(); }
};

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
    void u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); bytes(b, 8); }
    void vec_u64(const uint64_t* v, size_t n) { u64(n); bytes(v, n * 8); }     // little-endian host
    void native_modulus_u64() { u64(0); u64(0); u64(64); }
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
    uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)in[pos + i] << (8 * i); pos += 8; return v; }
    void raw(uint8_t* dst, size_t n) { if (!need(n)) return; std::memcpy(dst, in + pos, n); pos += n; }
    size_t vec_u64(uint64_t* dst, size_t max_words) {
        const uint64_t n = u64();
        if (!err.empty()) return 0;
        if (n > max_words) { err = "vector longer than the destination (" + std::to_string(n) + " words)"; return 0; }
        if (n > (len - pos) / 8) { err = "truncated input"; return 0; }
        std::memcpy(dst, in + pos, (size_t)n * 8);
        pos += (size_t)n * 8;
        return (size_t)n;
    }
    void native_modulus_u64() {
        const uint64_t lo = u64(), hi = u64(), bits = u64();
        if (!err.empty()) return;
        if (bits != 64) err = "expected an unsigned integer with 64 bits, got " + std::to_string(bits);
        else if (lo != 0 || hi != 0) err = "only the native modulus 2^64 is supported";
    }
};

int finish(Writer& w, size_t* written) {
    if (written) *written = w.pos;
    if (w.overflow) return fail("output buffer too small: " + std::to_string(w.pos) + " bytes needed");
    return 0;
}

size_t bsk_rows(const fhe_params_t& p) { return (size_t)fhe::n_ggsw(p) * p.pbs_level * (p.k + 1); }   // GLWE ciphertexts in the key
size_t ksk_rows(const fhe_params_t& p) { return (size_t)p.k * p.N * p.ks_level; }                       // LWE ciphertexts in the key

// GgswCiphertextList fields after `data`
void write_ggsw_list_meta(Writer& w, const fhe_params_t& p) {
    w.u64((uint64_t)p.k + 1);
    w.u64(p.N);
    w.u64(p.pbs_base_log);
    w.u64(p.pbs_level);
}
bool read_ggsw_list_meta(Reader& r, const fhe_params_t& p, std::string& why) {
    const uint64_t glwe_size = r.u64(), poly = r.u64(), base_log = r.u64(), level = r.u64();
    if (!r.err.empty()) return false;
    if (glwe_size != (uint64_t)p.k + 1 || poly != p.N || base_log != p.pbs_base_log || level != p.pbs_level) {
        why = "glwe_size " + std::to_string(glwe_size) + ", N " + std::to_string(poly) + ", base_log " + std::to_string(base_log) +
              ", level " + std::to_string(level);
        return false;
    }
    return true;
}

}  // namespace

namespace fhe {
// for the device-side expansion (seeded_kernels.hip.h)
void aes128_round_keys(const uint8_t key[16], uint8_t rk[11][16]) {
    const Aes128 a(key);
    std::memcpy(rk, a.rk, sizeof(a.rk));
}
const uint8_t* aes_sbox() { return kSbox; }
}  // namespace fhe

extern "C" {

int fhe_aes128_encrypt_block(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]) {
    if (!key || !in || !out) return fail("null pointer");
    Aes128(key).encrypt(in, out);
    return 0;
}

int fhe_seeded_mask_words(const uint8_t seed[16], uint64_t* out, size_t count) {
    if (!seed || (!out && count)) return fail("null pointer");
    MaskStream s(seed);
    s.words(out, count);
    return 0;
}

int fhe_seeded_decompress_keyswitch_key(const fhe_params_t* p, const uint8_t seed[16], const uint64_t* bodies, uint64_t* ksk) {
    if (!p || !seed || !bodies || !ksk) return fail("null pointer");
    MaskStream s(seed);
    const size_t rows = ksk_rows(*p), n = p->n;
    for (size_t r = 0; r < rows; r++) {
        s.words(ksk + r * (n + 1), n);
        ksk[r * (n + 1) + n] = bodies[r];
    }
    return 0;
}

int fhe_seeded_decompress_bootstrap_key(const fhe_params_t* p, const uint8_t seed[16], const uint64_t* bodies, uint64_t* bsk_std) {
    if (!p || !seed || !bodies || !bsk_std) return fail("null pointer");
    MaskStream s(seed);
    const size_t rows = bsk_rows(*p), N = p->N, k = p->k;
    for (size_t r = 0; r < rows; r++) {
        uint64_t* glwe = bsk_std + r * (k + 1) * N;
        s.words(glwe, k * N);
        std::memcpy(glwe + k * N, bodies + r * N, N * 8);
    }
    return 0;
}

int fhe_seeded_split_keyswitch_key(const fhe_params_t* p, const uint64_t* ksk, uint64_t* bodies) {
    if (!p || !ksk || !bodies) return fail("null pointer");
    const size_t rows = ksk_rows(*p), n = p->n;
    for (size_t r = 0; r < rows; r++) bodies[r] = ksk[r * (n + 1) + n];
    return 0;
}

int fhe_seeded_split_bootstrap_key(const fhe_params_t* p, const uint64_t* bsk_std, uint64_t* bodies) {
    if (!p || !bsk_std || !bodies) return fail("null pointer");
    const size_t rows = bsk_rows(*p), N = p->N, k = p->k;
    for (size_t r = 0; r < rows; r++) std::memcpy(bodies + r * N, bsk_std + r * (k + 1) * N + k * N, N * 8);
    return 0;
}

int fhe_wire_write_seeded_keyswitch_key(const fhe_params_t* p, const uint8_t seed[16], const uint64_t* bodies, uint8_t* out,
                                        size_t out_cap, size_t* written) {
    if (!p || !seed || !bodies) return fail("null pointer");
    Writer w{out, out_cap};
    w.vec_u64(bodies, ksk_rows(*p));
    w.u64(p->ks_base_log);
    w.u64(p->ks_level);
    w.u64((uint64_t)p->n + 1);
    w.bytes(seed, 16);
    w.native_modulus_u64();
    return finish(w, written);
}

int fhe_wire_read_seeded_keyswitch_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint8_t seed[16], uint64_t* bodies,
                                       size_t* consumed) {
    if (!p || !in || !seed || !bodies) return fail("null pointer");
    Reader r{in, in_len};
    const size_t want = ksk_rows(*p);
    const size_t n = r.vec_u64(bodies, want);
    const uint64_t base_log = r.u64(), level = r.u64(), out_size = r.u64();
    r.raw(seed, 16);
    r.native_modulus_u64();
    if (!r.err.empty()) return fail("SeededLweKeyswitchKey: " + r.err);
    if (base_log != p->ks_base_log || level != p->ks_level || out_size != (uint64_t)p->n + 1 || n != want)
        return fail("SeededLweKeyswitchKey does not match the parameter set (base_log " + std::to_string(base_log) + ", level " +
                    std::to_string(level) + ", output size " + std::to_string(out_size) + ", " + std::to_string(n) + " bodies)");
    if (consumed) *consumed = r.pos;
    return 0;
}

// SeededLweBootstrapKey, or SeededLweMultiBitBootstrapKey when the parameter set has a grouping factor
int fhe_wire_write_seeded_bootstrap_key(const fhe_params_t* p, const uint8_t seed[16], const uint64_t* bodies, uint8_t* out,
                                        size_t out_cap, size_t* written) {
    if (!p || !seed || !bodies) return fail("null pointer");
    Writer w{out, out_cap};
    w.vec_u64(bodies, bsk_rows(*p) * p->N);
    write_ggsw_list_meta(w, *p);
    w.bytes(seed, 16);
    w.native_modulus_u64();
    if (p->grouping_factor > 1) w.u64(p->grouping_factor);
    return finish(w, written);
}

int fhe_wire_read_seeded_bootstrap_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint8_t seed[16], uint64_t* bodies,
                                       size_t* consumed) {
    if (!p || !in || !seed || !bodies) return fail("null pointer");
    Reader r{in, in_len};
    const size_t want = bsk_rows(*p) * p->N;
    const size_t n = r.vec_u64(bodies, want);
    std::string why;
    const bool dims_ok = read_ggsw_list_meta(r, *p, why);
    r.raw(seed, 16);
    r.native_modulus_u64();
    const uint64_t grouping = p->grouping_factor > 1 ? r.u64() : 0;
    if (!r.err.empty()) return fail("SeededLweBootstrapKey: " + r.err);
    if (!dims_ok || n != want) return fail("SeededLweBootstrapKey does not match the parameter set (" + why + ", " + std::to_string(n) + " words)");
    if (p->grouping_factor > 1 && grouping != p->grouping_factor) return fail("SeededLweMultiBitBootstrapKey: grouping factor " + std::to_string(grouping));
    if (consumed) *consumed = r.pos;
    return 0;
}

// LweMultiBitBootstrapKey (uncompressed): GgswCiphertextList + grouping_factor
int fhe_wire_write_multi_bit_bootstrap_key(const fhe_params_t* p, const uint64_t* bsk_std, uint8_t* out, size_t out_cap,
                                           size_t* written) {
    if (!p || !bsk_std) return fail("null pointer");
    if (p->grouping_factor < 2) return fail("not a multi-bit parameter set");
    Writer w{out, out_cap};
    w.vec_u64(bsk_std, bsk_rows(*p) * (p->k + 1) * p->N);
    write_ggsw_list_meta(w, *p);
    w.native_modulus_u64();
    w.u64(p->grouping_factor);
    return finish(w, written);
}

int fhe_wire_read_multi_bit_bootstrap_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint64_t* bsk_std, size_t* consumed) {
    if (!p || !in || !bsk_std) return fail("null pointer");
    if (p->grouping_factor < 2) return fail("not a multi-bit parameter set");
    Reader r{in, in_len};
    const size_t want = bsk_rows(*p) * (p->k + 1) * p->N;
    const size_t n = r.vec_u64(bsk_std, want);
    std::string why;
    const bool dims_ok = read_ggsw_list_meta(r, *p, why);
    r.native_modulus_u64();
    const uint64_t grouping = r.u64();
    if (!r.err.empty()) return fail("LweMultiBitBootstrapKey: " + r.err);
    if (!dims_ok || n != want || grouping != p->grouping_factor)
        return fail("LweMultiBitBootstrapKey does not match the parameter set (" + why + ", grouping factor " + std::to_string(grouping) +
                    ", " + std::to_string(n) + " words)");
    if (consumed) *consumed = r.pos;
    return 0;
}

// shortint CompressedServerKey (shortint/server_key/compressed.rs:44-55): { key_switching_key: SeededLweKeyswitchKey,
// bootstrapping_key: enum { Classic(SeededLweBootstrapKey) = 0, MultiBit { seeded_bsk, deterministic_execution: bool } = 1 }
// (:11-17), message_modulus, carry_modulus, max_degree, ciphertext_modulus, pbs_order (unit enum, u32) } -- the object a
// client serializes for the server.
int fhe_wire_write_compressed_server_key(const fhe_params_t* p, const uint8_t ksk_seed[16], const uint64_t* ksk_bodies,
                                         const uint8_t bsk_seed[16], const uint64_t* bsk_bodies, uint64_t max_degree,
                                         uint32_t pbs_order, uint8_t* out, size_t out_cap, size_t* written) {
    if (!p || !ksk_seed || !ksk_bodies || !bsk_seed || !bsk_bodies) return fail("null pointer");
    size_t pos = 0, n = 0;
    bool overflow = false;
    auto room = [&](size_t at) { return out && at < out_cap ? out_cap - at : 0; };
    auto part = [&](int rc) { if (rc) overflow = true; pos += n; };
    part(fhe_wire_write_seeded_keyswitch_key(p, ksk_seed, ksk_bodies, out ? out + std::min(pos, out_cap) : nullptr, room(pos), &n));
    Writer w{out ? out + std::min(pos, out_cap) : nullptr, room(pos)};
    const uint32_t tag = p->grouping_factor > 1 ? 1u : 0u;
    for (int i = 0; i < 4; i++) { const uint8_t b = (uint8_t)(tag >> (8 * i)); w.bytes(&b, 1); }
    overflow |= w.overflow;
    pos += w.pos;
    part(fhe_wire_write_seeded_bootstrap_key(p, bsk_seed, bsk_bodies, out ? out + std::min(pos, out_cap) : nullptr, room(pos), &n));
    Writer t{out ? out + std::min(pos, out_cap) : nullptr, room(pos)};
    if (tag == 1) { const uint8_t deterministic = 1; t.bytes(&deterministic, 1); }
    t.u64(p->msg_mod);
    t.u64(p->carry_mod);
    t.u64(max_degree);
    t.native_modulus_u64();
    for (int i = 0; i < 4; i++) { const uint8_t b = (uint8_t)(pbs_order >> (8 * i)); t.bytes(&b, 1); }
    overflow |= t.overflow;
    pos += t.pos;
    if (written) *written = pos;
    if (out && overflow) return fail("output buffer too small: " + std::to_string(pos) + " bytes needed");
    return 0;
}

int fhe_wire_read_compressed_server_key(const fhe_params_t* p, const uint8_t* in, size_t in_len, uint8_t ksk_seed[16],
                                        uint64_t* ksk_bodies, uint8_t bsk_seed[16], uint64_t* bsk_bodies, uint64_t* max_degree,
                                        uint32_t* pbs_order, size_t* consumed) {
    if (!p || !in || !ksk_seed || !ksk_bodies || !bsk_seed || !bsk_bodies) return fail("null pointer");
    size_t pos = 0, n = 0;
    if (fhe_wire_read_seeded_keyswitch_key(p, in, in_len, ksk_seed, ksk_bodies, &n)) return 1;
    pos += n;
    Reader r{in + pos, in_len - pos};
    uint8_t tb[4] = {0, 0, 0, 0};
    r.raw(tb, 4);
    if (!r.err.empty()) return fail("CompressedServerKey: " + r.err);
    const uint32_t tag = (uint32_t)tb[0] | (uint32_t)tb[1] << 8 | (uint32_t)tb[2] << 16 | (uint32_t)tb[3] << 24;
    if (tag > 1) return fail("CompressedServerKey: unknown bootstrapping key variant " + std::to_string(tag));
    if ((tag == 1) != (p->grouping_factor > 1))
        return fail(tag == 1 ? "CompressedServerKey holds a multi-bit bootstrapping key, the parameter set is classic"
                             : "CompressedServerKey holds a classic bootstrapping key, the parameter set is multi-bit");
    pos += 4;
    if (fhe_wire_read_seeded_bootstrap_key(p, in + pos, in_len - pos, bsk_seed, bsk_bodies, &n)) return 1;
    pos += n;
    Reader t{in + pos, in_len - pos};
    if (tag == 1) { uint8_t deterministic = 0; t.raw(&deterministic, 1); if (deterministic > 1 && t.err.empty()) t.err = "invalid bool"; }
    const uint64_t msg = t.u64(), carry = t.u64(), deg = t.u64();
    t.native_modulus_u64();
    uint8_t ob[4] = {0, 0, 0, 0};
    t.raw(ob, 4);
    if (!t.err.empty()) return fail("CompressedServerKey: " + t.err);
    const uint32_t order = (uint32_t)ob[0] | (uint32_t)ob[1] << 8 | (uint32_t)ob[2] << 16 | (uint32_t)ob[3] << 24;
    if (msg != p->msg_mod || carry != p->carry_mod)
        return fail("CompressedServerKey: message / carry modulus " + std::to_string(msg) + " / " + std::to_string(carry) +
                    " do not match the parameter set");
    if (order > 1) return fail("CompressedServerKey: unknown PBSOrder " + std::to_string(order));
    if (max_degree) *max_degree = deg;
    if (pbs_order) *pbs_order = order;
    if (consumed) *consumed = pos + t.pos;
    return 0;
}

// Seeded LWE ciphertexts: every ciphertext has its own compression seed (seeded_lwe_ciphertext_decompression.rs:11-50).
int fhe_seeded_decompress_lwe_batch(uint32_t lwe_dim, const uint8_t* seeds, const uint64_t* bodies, uint32_t count, uint64_t* out) {
    if (count && (!seeds || !bodies || !out)) return fail("null pointer");
    for (uint32_t c = 0; c < count; c++) {
        MaskStream s(seeds + (size_t)c * 16);
        uint64_t* ct = out + (size_t)c * (lwe_dim + 1);
        s.words(ct, lwe_dim);
        ct[lwe_dim] = bodies[c];
    }
    return 0;
}

// shortint CompressedCiphertext { ct: SeededLweCiphertext { data: u64, lwe_size, compression_seed, ciphertext_modulus },
// degree, message_modulus, carry_modulus, pbs_order, noise_level }   shortint/ciphertext/mod.rs:471-478,
// entities/seeded_lwe_ciphertext.rs:13-18 -- note the field order differs from shortint::Ciphertext's
int fhe_wire_write_compressed_ciphertext(uint64_t body, size_t lwe_size, const uint8_t seed[16], const fhe_shortint_meta* meta,
                                         uint8_t* out, size_t out_cap, size_t* written) {
    if (!seed || !meta) return fail("null pointer");
    Writer w{out, out_cap};
    w.u64(body);
    w.u64(lwe_size);
    w.bytes(seed, 16);
    w.native_modulus_u64();
    w.u64(meta->degree);
    w.u64(meta->message_modulus);
    w.u64(meta->carry_modulus);
    for (int i = 0; i < 4; i++) { const uint8_t b = (uint8_t)(meta->pbs_order >> (8 * i)); w.bytes(&b, 1); }
    w.u64(meta->noise_level);
    return finish(w, written);
}

int fhe_wire_read_compressed_ciphertext(const uint8_t* in, size_t in_len, uint64_t* body, size_t* lwe_size, uint8_t seed[16],
                                        fhe_shortint_meta* meta, size_t* consumed) {
    if (!in || !body || !seed || !meta) return fail("null pointer");
    Reader r{in, in_len};
    *body = r.u64();
    const uint64_t size = r.u64();
    r.raw(seed, 16);
    r.native_modulus_u64();
    meta->degree = r.u64();
    meta->message_modulus = r.u64();
    meta->carry_modulus = r.u64();
    uint8_t ob[4] = {0, 0, 0, 0};
    r.raw(ob, 4);
    meta->noise_level = r.u64();
    if (!r.err.empty()) return fail("CompressedCiphertext: " + r.err);
    meta->pbs_order = (uint32_t)ob[0] | (uint32_t)ob[1] << 8 | (uint32_t)ob[2] << 16 | (uint32_t)ob[3] << 24;
    if (meta->pbs_order > 1) return fail("CompressedCiphertext: unknown PBSOrder " + std::to_string(meta->pbs_order));
    if (size == 0) return fail("CompressedCiphertext: lwe_size 0");
    if (lwe_size) *lwe_size = (size_t)size;
    if (consumed) *consumed = r.pos;
    return 0;
}

// integer RadixCiphertext / CompressedRadixCiphertext = BaseRadixCiphertext<Block> { blocks: Vec<Block> }, least significant
// block first (integer/ciphertext/mod.rs:18-21,30,45): a u64 count, then the blocks -- an FheUint8 character is four of
// them under PARAM_MESSAGE_2_CARRY_2, a string a sequence of those.
int fhe_wire_write_radix_ciphertext(const uint64_t* cts, size_t lwe_size, const fhe_shortint_meta* metas, size_t n_blocks,
                                    uint8_t* out, size_t out_cap, size_t* written) {
    if ((!cts || !metas) && n_blocks) return fail("null pointer");
    size_t pos = 0;
    bool overflow = false;
    {
        Writer w{out, out_cap};
        w.u64(n_blocks);
        overflow |= w.overflow;
        pos = w.pos;
    }
    for (size_t b = 0; b < n_blocks; b++) {
        size_t n = 0;
        const bool have = out && pos < out_cap;
        if (fhe_wire_write_shortint_ciphertext(cts + b * lwe_size, lwe_size, &metas[b], 0, have ? out + pos : nullptr,
                                               have ? out_cap - pos : 0, &n)) overflow = true;
        if (out && !have) overflow = true;
        pos += n;
    }
    if (written) *written = pos;
    if (out && overflow) return fail("output buffer too small: " + std::to_string(pos) + " bytes needed");
    return 0;
}

int fhe_wire_read_radix_ciphertext(const uint8_t* in, size_t in_len, uint64_t* cts, size_t lwe_size, size_t max_blocks,
                                   fhe_shortint_meta* metas, size_t* n_blocks, size_t* consumed) {
    if (!in || !cts || !metas || !n_blocks) return fail("null pointer");
    Reader r{in, in_len};
    const uint64_t n = r.u64();
    if (!r.err.empty()) return fail("RadixCiphertext: " + r.err);
    if (n > max_blocks) return fail("RadixCiphertext: " + std::to_string(n) + " blocks, room for " + std::to_string(max_blocks));
    size_t pos = r.pos;
    for (uint64_t b = 0; b < n; b++) {
        size_t size = 0, used = 0;
        if (fhe_wire_read_shortint_ciphertext(in + pos, in_len - pos, 0, 0, cts + b * lwe_size, lwe_size, &size, &metas[b], &used)) return 1;
        if (size != lwe_size) return fail("RadixCiphertext: block " + std::to_string(b) + " has LWE size " + std::to_string(size));
        pos += used;
    }
    *n_blocks = (size_t)n;
    if (consumed) *consumed = pos;
    return 0;
}

int fhe_wire_write_compressed_radix_ciphertext(const uint64_t* bodies, const uint8_t* seeds, size_t lwe_size,
                                               const fhe_shortint_meta* metas, size_t n_blocks, uint8_t* out, size_t out_cap,
                                               size_t* written) {
    if ((!bodies || !seeds || !metas) && n_blocks) return fail("null pointer");
    size_t pos = 0;
    bool overflow = false;
    {
        Writer w{out, out_cap};
        w.u64(n_blocks);
        overflow |= w.overflow;
        pos = w.pos;
    }
    for (size_t b = 0; b < n_blocks; b++) {
        size_t n = 0;
        const bool have = out && pos < out_cap;
        if (fhe_wire_write_compressed_ciphertext(bodies[b], lwe_size, seeds + b * 16, &metas[b], have ? out + pos : nullptr,
                                                 have ? out_cap - pos : 0, &n)) overflow = true;
        if (out && !have) overflow = true;
        pos += n;
    }
    if (written) *written = pos;
    if (out && overflow) return fail("output buffer too small: " + std::to_string(pos) + " bytes needed");
    return 0;
}

int fhe_wire_read_compressed_radix_ciphertext(const uint8_t* in, size_t in_len, uint64_t* bodies, uint8_t* seeds, size_t* lwe_size,
                                              size_t max_blocks, fhe_shortint_meta* metas, size_t* n_blocks, size_t* consumed) {
    if (!in || !bodies || !seeds || !metas || !n_blocks) return fail("null pointer");
    Reader r{in, in_len};
    const uint64_t n = r.u64();
    if (!r.err.empty()) return fail("CompressedRadixCiphertext: " + r.err);
    if (n > max_blocks) return fail("CompressedRadixCiphertext: " + std::to_string(n) + " blocks, room for " + std::to_string(max_blocks));
    size_t pos = r.pos, common = 0;
    for (uint64_t b = 0; b < n; b++) {
        size_t size = 0, used = 0;
        if (fhe_wire_read_compressed_ciphertext(in + pos, in_len - pos, &bodies[b], &size, seeds + b * 16, &metas[b], &used)) return 1;
        if (b == 0) common = size;
        else if (size != common) return fail("CompressedRadixCiphertext: blocks of different LWE sizes");
        pos += used;
    }
    *n_blocks = (size_t)n;
    if (lwe_size) *lwe_size = common;
    if (consumed) *consumed = pos;
    return 0;
}

}  // extern "C"
