// client.cpp -- client side of the engine: secret keys, encryption, decryption, server-key
// generation.  CPU code by design: in the reference these are ClientKey operations that never run
// on the evaluation path (shortint/engine/client_side.rs:13-128, shortint/client_key/mod.rs:281-337,
// shortint/engine/server_side.rs:54-160) and SURVEY.md section 8(f) ranks device-side key generation
// as a later row.  Randomness: ChaCha20 keystream under a 256-bit seed, one stream per purpose and key
// row (det_math.h; stands in for the reference's AES-128-CTR concrete-csprng with forked generators);
// noise follows the reference's Gaussian sampler
// (core_crypto/commons/math/random/gaussian.rs:17-47, polar method on two signed 64-bit draws) --
// both live in det_math.h, shared with the device-side key generation kernels.
#include <errno.h>
#include <sys/random.h>

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "det_math.h"
#include "engine.h"

namespace fhe {

struct ClientKey {
    fhe_params_t p;
    Seed256 seed;
    std::vector<uint64_t> glwe_sk;    // k*N bits; also the big LWE key (client_side.rs:29)
    std::vector<uint64_t> small_sk;   // n bits
    Rng enc_rng;

    ClientKey(const fhe_params_t& params, const Seed256& seed_)
        : p(params), seed(seed_), glwe_sk((size_t)params.k * params.N), small_sk(params.n), enc_rng(seed_, 3) {
        fill_binary(glwe_sk, 1);
        fill_binary(small_sk, 2);
    }
    void fill_binary(std::vector<uint64_t>& key, uint64_t stream) {   // client_side.rs:13-27
        Rng r(seed, stream);
        for (size_t i = 0; i < key.size(); i += 64) {
            const uint64_t w = r.next();
            for (size_t b = 0; b < 64 && i + b < key.size(); b++) key[i + b] = (w >> b) & 1;
        }
    }
    uint64_t delta() const { return (1ull << 63) / ((uint64_t)p.msg_mod * p.carry_mod); }

    // core_crypto/algorithms/lwe_encryption.rs:61-110
    static void lwe_encrypt(const uint64_t* sk, size_t dim, uint64_t pt, double std_dev, Rng& r, uint64_t* ct) {
        uint64_t acc = 0;
        for (size_t i = 0; i < dim; i++) {
            ct[i] = r.next();
            acc += ct[i] * sk[i];
        }
        ct[dim] = acc + gaussian_torus(r, std_dev) + pt;
    }
    uint64_t phase(const uint64_t* ct) const {
        const size_t dim = glwe_sk.size();
        uint64_t acc = 0;
        for (size_t i = 0; i < dim; i++) acc += ct[i] * glwe_sk[i];
        return ct[dim] - acc;
    }
    // shortint/client_key/mod.rs:281-303
    uint64_t decode(uint64_t x) const {
        const uint64_t d = delta(), rounding = (x & (d >> 1)) << 1;
        return (x + rounding) / d;
    }

    // lwe_keyswitch_key_generation.rs:65-130
    void gen_ksk(uint64_t* ksk) const {
        const size_t in_dim = glwe_sk.size(), osz = (size_t)p.n + 1;
        for (size_t i = 0; i < in_dim; i++) {
            Rng r(seed, 0x4B534B0000000000ull + i);
            for (uint32_t it = 0; it < p.ks_level; it++) {
                const uint32_t level = p.ks_level - it;
                const uint64_t pt = glwe_sk[i] << (64 - p.ks_base_log * level);
                lwe_encrypt(small_sk.data(), p.n, pt, p.lwe_std, r, ksk + (i * p.ks_level + it) * osz);
            }
        }
    }
    // glwe_encryption.rs:17-60 with a binary key: body += e + sum_q A_q * S_q (signed shifts)
    void glwe_encrypt_assign(uint64_t* glwe, Rng& r) const {
        const uint32_t N = p.N, k = p.k;
        uint64_t* body = glwe + (size_t)k * N;
        for (size_t j = 0; j < (size_t)k * N; j++) glwe[j] = r.next();
        for (uint32_t j = 0; j < N; j++) body[j] += gaussian_torus(r, p.glwe_std);
        for (uint32_t q = 0; q < k; q++) {
            const uint64_t *a = glwe + (size_t)q * N, *s = glwe_sk.data() + (size_t)q * N;
            for (uint32_t t = 0; t < N; t++) {
                if (!s[t]) continue;
                for (uint32_t c = 0; c < N - t; c++) body[c + t] += a[c];
                for (uint32_t c = N - t; c < N; c++) body[c + t - N] -= a[c];
            }
        }
    }
    // lwe_bootstrap_key_generation.rs:76-135 + ggsw_encryption.rs:72-151,300-331
    // plaintext bit of every GGSW of the key (classic: the small key; multi-bit: per-group products)
    std::vector<uint64_t> ggsw_bits() const {
        const uint32_t ng = n_ggsw(p), gf = p.grouping_factor > 1 ? p.grouping_factor : 1;
        std::vector<uint64_t> bits(ng);
        for (uint32_t i = 0; i < ng; i++)
            bits[i] = gf == 1 ? small_sk[i]
                              : multi_bit_key_bit(small_sk.data() + (size_t)(i >> gf) * gf, gf, i & ((1u << gf) - 1));
        return bits;
    }
    // (multi-bit: lwe_multi_bit_bootstrap_key_generation.rs:87-173, the same GGSW encryption per entry)
    void gen_bsk_range(uint64_t* bsk, const uint64_t* bits, size_t lo, size_t hi) const {
        const uint32_t N = p.N, k = p.k, k1 = k + 1, L = p.pbs_level;
        const size_t glwe_len = (size_t)k1 * N, ggsw_len = (size_t)L * k1 * glwe_len;
        for (size_t i = lo; i < hi; i++) {
            Rng r(seed, 0x42534B0000000000ull + i);
            uint64_t* ggsw = bsk + i * ggsw_len;
            const uint64_t m = bits[i];
            for (uint32_t li = 0; li < L; li++) {
                const uint64_t factor = (0 - m) * (1ull << (64 - p.pbs_base_log * (li + 1)));
                for (uint32_t row = 0; row < k1; row++) {
                    uint64_t* glwe = ggsw + ((size_t)li * k1 + row) * glwe_len;
                    uint64_t* body = glwe + (size_t)k * N;
                    if (row < k) {
                        const uint64_t* s = glwe_sk.data() + (size_t)row * N;
                        for (uint32_t c = 0; c < N; c++) body[c] = s[c] * factor;
                    } else {
                        std::memset(body, 0, N * sizeof(uint64_t));
                        body[0] = 0 - factor;
                    }
                    glwe_encrypt_assign(glwe, r);
                }
            }
        }
    }
    void gen_bsk(uint64_t* bsk, int threads) const {
        if (threads < 1) threads = 1;
        const std::vector<uint64_t> bits = ggsw_bits();
        const size_t ng = bits.size();
        const uint64_t* b = bits.data();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([=] { gen_bsk_range(bsk, b, ng * t / threads, ng * (t + 1) / threads); });
        for (auto& th : pool) th.join();
    }
};

}  // namespace fhe

struct fhe_client_key {
    fhe::ClientKey* impl;
};

extern "C" {

size_t fhe_params_ksk_len(const fhe_params_t* p) { return (size_t)p->k * p->N * p->ks_level * (p->n + 1); }
size_t fhe_params_bsk_len(const fhe_params_t* p) {
    return (size_t)fhe::n_ggsw(*p) * p->pbs_level * (p->k + 1) * (p->k + 1) * p->N;
}

int fhe_random_seed(uint8_t seed[32]) {
    if (!seed) return fhe::fail("null pointer: seed");
    size_t got = 0;
    while (got < 32) {                       // the kernel's CSPRNG (getrandom(2)); never blocks once initialised
        const ssize_t r = getrandom(seed + got, 32 - got, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            return fhe::fail("getrandom failed: no entropy source");
        }
        got += (size_t)r;
    }
    return 0;
}

int fhe_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    if (!key || !out) return fhe::fail("null pointer");
    fhe::chacha20_block(fhe::seed_from_bytes(key), counter, stream, out);
    return 0;
}

int fhe_client_key_create(const fhe_params_t* params, const uint8_t seed[32], fhe_client_key** out) {
    if (!out) return fhe::fail("null pointer: out");
    *out = nullptr;
    if (!params) return fhe::fail("null pointer: params");
    if (!seed) return fhe::fail("null pointer: seed");
    try {
        *out = new fhe_client_key{new fhe::ClientKey(*params, fhe::seed_from_bytes(seed))};
    } catch (const std::exception& e) {
        return fhe::fail(e.what());
    }
    return 0;
}

int fhe_client_key_destroy(fhe_client_key* ck) {
    if (ck) {
        delete ck->impl;
        delete ck;
    }
    return 0;
}

int fhe_client_encrypt(fhe_client_key* ck, const uint64_t* msgs, uint32_t count, uint64_t* cts) {
    if (!ck || !msgs || !cts) return fhe::fail("null pointer");
    auto& c = *ck->impl;
    const size_t dim = c.glwe_sk.size();
    for (uint32_t i = 0; i < count; i++)
        fhe::ClientKey::lwe_encrypt(c.glwe_sk.data(), dim, msgs[i] * c.delta(), c.p.glwe_std, c.enc_rng,
                                    cts + (size_t)i * (dim + 1));
    return 0;
}

int fhe_client_decrypt(fhe_client_key* ck, const uint64_t* cts, uint32_t count, uint64_t* msgs) {
    if (!ck || !msgs || !cts) return fhe::fail("null pointer");
    auto& c = *ck->impl;
    const size_t dim = c.glwe_sk.size();
    for (uint32_t i = 0; i < count; i++) msgs[i] = c.decode(c.phase(cts + (size_t)i * (dim + 1)));
    return 0;
}

int fhe_client_gen_server_keys(fhe_client_key* ck, uint64_t* bsk_std, uint64_t* ksk, int threads) {
    if (!ck || !bsk_std || !ksk) return fhe::fail("null pointer");
    try {
        ck->impl->gen_ksk(ksk);
        ck->impl->gen_bsk(bsk_std, threads);
    } catch (const std::exception& e) {
        return fhe::fail(e.what());
    }
    return 0;
}

int fhe_client_secret_keys(fhe_client_key* ck, uint64_t* glwe_sk, uint64_t* small_sk) {
    if (!ck) return fhe::fail("null pointer");
    auto& c = *ck->impl;
    if (glwe_sk) std::memcpy(glwe_sk, c.glwe_sk.data(), c.glwe_sk.size() * 8);
    if (small_sk) std::memcpy(small_sk, c.small_sk.data(), c.small_sk.size() * 8);
    return 0;
}

}  // extern "C"
