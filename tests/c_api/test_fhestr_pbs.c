/* Plain-C caller of libfhestr.so: the scenarios the reference's C API test exercises for this path
 * (tfhe/c_api_tests/test_shortint_pbs.c:40-190 — univariate PBS of every 2-bit message, PBS of a
 * PBS output, bivariate PBS of every message pair), on PARAM_MESSAGE_2_CARRY_2_KS_PBS, plus one
 * FheString eq/contains.  No Python, no torch: gcc + the C ABI of include/fhestr.h only.
 *
 *   gcc -O2 -Iinclude tests/c_api/test_fhestr_pbs.c -Lfhe-string-bounty_amd -lfhestr \
 *       -Wl,-rpath,$PWD/fhe-string-bounty_amd -o /tmp/test_fhestr_pbs && /tmp/test_fhestr_pbs
 */
#include "fhestr.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        if ((call) != 0) {                                                                           \
            fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, fhe_last_error()); \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)
#define EXPECT(cond)                                                                                 \
    do {                                                                                             \
        if (!(cond)) {                                                                               \
            fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond);          \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

static const fhe_params_t P22 = {742, 1, 2048, 23, 1, 3, 5, 4, 4, 0.000007069849454709433,
                                 0.00000000000000029403601535432533, 0 /* classic PBS */};
enum { M = 16, MSG = 4 };

static uint64_t twice(uint64_t x) { return (x * 2) % MSG; }
static uint64_t product(uint64_t l, uint64_t r) { return (l * r) % MSG; }

typedef struct {
    fhe_client_key *ck;
    fhe_engine *eng;
    size_t big;
} ctx_t;

static ctx_t setup(void) {
    ctx_t c;
    /* a fixed 256-bit seed: tests only -- production callers take it from fhe_random_seed() */
    static const uint8_t seed[32] = {0xEE, 0xFF, 0xC0};
    CHECK(fhe_client_key_create(&P22, seed, &c.ck));
    uint64_t *bsk = malloc(fhe_params_bsk_len(&P22) * sizeof(uint64_t));
    uint64_t *ksk = malloc(fhe_params_ksk_len(&P22) * sizeof(uint64_t));
    EXPECT(bsk && ksk);
    CHECK(fhe_client_gen_server_keys(c.ck, bsk, ksk, 8));
    CHECK(fhe_engine_create(&P22, 0, &c.eng));
    CHECK(fhe_engine_load_keys(c.eng, bsk, ksk));
    free(bsk);
    free(ksk);
    c.big = (size_t)P22.k * P22.N + 1;
    return c;
}

static void teardown(ctx_t *c) {
    CHECK(fhe_engine_destroy(c->eng));
    CHECK(fhe_client_key_destroy(c->ck));
}

/* test_shortint_pbs.c:40-112: PBS with f(x) = 2x mod 4 on every message, then again on the output */
static void test_univariate(ctx_t *c) {
    uint64_t table[M], degree = 0, msgs[MSG], got[MSG];
    uint32_t lut;
    for (uint64_t i = 0; i < M; ++i) table[i] = twice(i);
    CHECK(fhe_lut_generate(c->eng, table, &lut, &degree));
    EXPECT(degree == 2); /* max of 2x mod 4 */
    for (uint64_t i = 0; i < MSG; ++i) msgs[i] = i;
    uint64_t *ct = malloc(MSG * c->big * 8), *out = malloc(MSG * c->big * 8);
    uint32_t idx[MSG] = {lut, lut, lut, lut};
    CHECK(fhe_client_encrypt(c->ck, msgs, MSG, ct));
    CHECK(fhe_ks_pbs_batch(c->eng, ct, idx, out, MSG));
    CHECK(fhe_client_decrypt(c->ck, out, MSG, got));
    for (uint64_t i = 0; i < MSG; ++i) EXPECT(got[i] == twice(i));
    CHECK(fhe_ks_pbs_batch(c->eng, out, idx, ct, MSG)); /* the reference's _assign round */
    CHECK(fhe_client_decrypt(c->ck, ct, MSG, got));
    for (uint64_t i = 0; i < MSG; ++i) EXPECT(got[i] == twice(twice(i)));
    free(ct);
    free(out);
}

/* test_shortint_pbs.c:114-183: bivariate PBS = pack left*msg_mod + right (bivariate_pbs.rs:167-182)
 * then one PBS with the wrapped table (bivariate_pbs.rs:80-104) */
static void test_bivariate(ctx_t *c) {
    uint64_t table[M], degree = 0;
    uint32_t lut;
    for (uint64_t i = 0; i < M; ++i) table[i] = product(i / MSG, i % MSG);
    CHECK(fhe_lut_generate(c->eng, table, &lut, &degree));
    EXPECT(degree == 3);
    enum { PAIRS = MSG * MSG };
    uint64_t msgs[2 * PAIRS], got[PAIRS], cst[PAIRS] = {0};
    uint32_t off[PAIRS + 1], src[2 * PAIRS], idx[PAIRS];
    int32_t coeff[2 * PAIRS];
    for (uint32_t p = 0; p < PAIRS; ++p) {
        msgs[2 * p] = p / MSG;
        msgs[2 * p + 1] = p % MSG;
        off[p] = 2 * p;
        src[2 * p] = 2 * p;
        src[2 * p + 1] = 2 * p + 1;
        coeff[2 * p] = MSG;
        coeff[2 * p + 1] = 1;
        idx[p] = lut;
    }
    off[PAIRS] = 2 * PAIRS;
    uint64_t *ct = malloc(2 * PAIRS * c->big * 8), *packed = malloc(PAIRS * c->big * 8),
             *out = malloc(PAIRS * c->big * 8);
    CHECK(fhe_client_encrypt(c->ck, msgs, 2 * PAIRS, ct));
    CHECK(fhe_lwe_lincomb_batch(c->eng, ct, 2 * PAIRS, off, src, coeff, cst, packed, PAIRS));
    CHECK(fhe_ks_pbs_batch(c->eng, packed, idx, out, PAIRS));
    CHECK(fhe_client_decrypt(c->ck, out, PAIRS, got));
    for (uint32_t p = 0; p < PAIRS; ++p) EXPECT(got[p] == product(p / MSG, p % MSG));
    /* second round: (previous output, right) -> product again, as the reference's _assign call */
    for (uint32_t p = 0; p < PAIRS; ++p) {
        memcpy(ct + (size_t)(2 * p) * c->big, out + (size_t)p * c->big, c->big * 8);
    }
    CHECK(fhe_lwe_lincomb_batch(c->eng, ct, 2 * PAIRS, off, src, coeff, cst, packed, PAIRS));
    CHECK(fhe_ks_pbs_batch(c->eng, packed, idx, out, PAIRS));
    CHECK(fhe_client_decrypt(c->ck, out, PAIRS, got));
    for (uint32_t p = 0; p < PAIRS; ++p)
        EXPECT(got[p] == product(product(p / MSG, p % MSG), p % MSG));
    free(ct);
    free(packed);
    free(out);
}

/* an encrypted string: cap chars x 4 two-bit blocks, little endian, zero padded */
static uint64_t *encrypt_string(ctx_t *c, const char *s, uint32_t cap) {
    uint32_t blocks = cap * 4;
    uint64_t *msgs = calloc(blocks, 8), *ct = malloc((size_t)blocks * c->big * 8);
    for (uint32_t i = 0; i < cap && s[i]; ++i)
        for (uint32_t b = 0; b < 4; ++b) msgs[4 * i + b] = ((uint8_t)s[i] >> (2 * b)) & 3;
    CHECK(fhe_client_encrypt(c->ck, msgs, blocks, ct));
    free(msgs);
    return ct;
}

static void test_strings(ctx_t *c) {
    enum { CAP = 8 };
    uint64_t *a = encrypt_string(c, "fhe gpu", CAP), *b = encrypt_string(c, "fhe gpu", CAP),
             *d = encrypt_string(c, "fhe cpu", CAP), *pat = encrypt_string(c, "gpu", 4);
    uint64_t *out = malloc(c->big * 8), bit = 9;
    CHECK(fhe_str_eq(c->eng, a, CAP, b, CAP, out));
    CHECK(fhe_client_decrypt(c->ck, out, 1, &bit));
    EXPECT(bit == 1);
    CHECK(fhe_str_eq(c->eng, a, CAP, d, CAP, out));
    CHECK(fhe_client_decrypt(c->ck, out, 1, &bit));
    EXPECT(bit == 0);
    CHECK(fhe_str_contains(c->eng, a, CAP, pat, 4, out));
    CHECK(fhe_client_decrypt(c->ck, out, 1, &bit));
    EXPECT(bit == 1);
    CHECK(fhe_str_contains(c->eng, d, CAP, pat, 4, out));
    CHECK(fhe_client_decrypt(c->ck, out, 1, &bit));
    EXPECT(bit == 0);
    CHECK(fhe_str_starts_with_clear(c->eng, a, CAP, (const uint8_t *)"fhe", 3, out));
    CHECK(fhe_client_decrypt(c->ck, out, 1, &bit));
    EXPECT(bit == 1);
    free(a);
    free(b);
    free(d);
    free(pat);
    free(out);
}

static void test_errors(ctx_t *c) {
    uint64_t dummy[4] = {0};
    uint32_t bad = 12345;
    EXPECT(fhe_pbs_batch(c->eng, dummy, &bad, dummy, 1) != 0); /* unknown LUT id */
    EXPECT(strlen(fhe_last_error()) > 0);
    fhe_params_t p = P22;
    p.N = 3000; /* not a power of two */
    fhe_engine *e = NULL;
    EXPECT(fhe_engine_create(&p, 0, &e) != 0 && e == NULL);
}

int main(void) {
    ctx_t c = setup();
    test_univariate(&c);
    test_bivariate(&c);
    test_strings(&c);
    test_errors(&c);
    teardown(&c);
    printf("c_api ok\n");
    return 0;
}
