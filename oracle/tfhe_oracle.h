/*
 * tfhe_oracle.h -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * A plain-C restatement of the tfhe-rs 0.5.0 hot path that the reference repository
 * (Lcressot/fhe-string-bounty, read-only at /root/reference) runs for every encrypted
 * string operation: LWE keyswitch -> programmable bootstrap (blind rotation with the
 * f64 twisted-FFT external product) -> sample extraction, plus the shortint lookup-table
 * generation and the minimal client side (keygen / encrypt / decrypt) a test harness needs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The shipped library (fhe-string-bounty_amd/csrc) never links or calls it.
 *
 * Parity status: every integer-only function is pinned by the reference's doctest
 * known-answer vectors (tests/test_oracle_kat.py).  The f64 FFT lives in the un-vendored
 * crate concrete-fft 0.3.0 (tfhe/Cargo.toml:60); its exact output bits are "parity
 * unpinned" -- the reference itself only asserts tolerance / decrypt-level results
 * (fft64/math/fft/tests.rs:40-46,166-173; fft64/crypto/tests.rs:5-13), and so do we.
 *
 * All citations are file:line under /root/reference/tfhe/src unless stated otherwise.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shortint/parameters/mod.rs:61-76 (ClassicPBSParameters), native modulus 2^64 only. */
typedef struct {
    uint32_t n;            /* lwe_dimension (small key)            */
    uint32_t k;            /* glwe_dimension                       */
    uint32_t N;            /* polynomial_size                      */
    uint32_t pbs_base_log; /* pbs_base_log                         */
    uint32_t pbs_level;    /* pbs_level                            */
    uint32_t ks_base_log;  /* ks_base_log                          */
    uint32_t ks_level;     /* ks_level                             */
    uint32_t msg_mod;      /* message_modulus                      */
    uint32_t carry_mod;    /* carry_modulus                        */
    double lwe_std;        /* lwe_modular_std_dev  (small key / KSK noise) */
    double glwe_std;       /* glwe_modular_std_dev (BSK + fresh big-key ciphertexts) */
} orc_params;

/* ---- integer helpers, generic in the scalar width `bits` (8/32/64) so the reference's
 *      u8/u32 doctest vectors can be replayed verbatim ---- */
uint64_t orc_closest_representable(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits);
void orc_decompose(uint64_t x, uint32_t base_log, uint32_t level, uint32_t bits, uint64_t *digits);
uint64_t orc_modulus_switch(uint64_t x, uint32_t log2_poly_size);
void orc_monomial_div(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree, uint32_t bits);
void orc_monomial_mul(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree, uint32_t bits);
void orc_monomial_mul_and_subtract(uint64_t *out, const uint64_t *in, uint32_t N, uint64_t degree,
                                   uint32_t bits);
void orc_slice_sub_scalar_mul(uint64_t *out, const uint64_t *in, uint64_t scalar, size_t len,
                              uint32_t bits);
uint64_t orc_from_torus(double x);
int64_t orc_f64_to_i64(double x);

/* ---- the hot path (u64 torus) ---- */
void orc_keyswitch(const orc_params *p, const uint64_t *ksk, const uint64_t *lwe_big,
                   uint64_t *lwe_small);
void orc_sample_extract(const orc_params *p, const uint64_t *acc, uint64_t *lwe_big);

/* f64 negacyclic FFT plan for polynomial size N (complex size N/2). */
typedef struct orc_fft orc_fft;
orc_fft *orc_fft_new(uint32_t N);
void orc_fft_free(orc_fft *f);
/* out: N/2 complex as interleaved (re,im) doubles */
void orc_fft_forward_as_integer(const orc_fft *f, double *out, const uint64_t *poly);
void orc_fft_forward_as_torus(const orc_fft *f, double *out, const uint64_t *poly);
void orc_fft_add_backward_as_torus(const orc_fft *f, uint64_t *poly, double *fourier_inout);

/* standard-domain BSK (n GGSWs) -> Fourier BSK: n*level*(k+1)*(k+1)*(N/2) complex (2 doubles) */
void orc_bsk_to_fourier(const orc_params *p, const uint64_t *bsk_std, double *fbsk);

void orc_add_external_product_fft(const orc_params *p, const orc_fft *f, uint64_t *out_glwe,
                                  const double *fggsw, const uint64_t *glwe);
void orc_add_external_product_exact(const orc_params *p, uint64_t *out_glwe,
                                    const uint64_t *ggsw_std, const uint64_t *glwe);

/* acc (in/out): (k+1)*N, initialised by the caller with the LUT. */
void orc_blind_rotate_fft(const orc_params *p, const orc_fft *f, const double *fbsk,
                          const uint64_t *lwe_small, uint64_t *acc);
void orc_blind_rotate_exact(const orc_params *p, const uint64_t *bsk_std,
                            const uint64_t *lwe_small, uint64_t *acc);

/* programmable bootstrap: small LWE (n+1) + LUT ((k+1)N) -> big LWE (kN+1). */
void orc_pbs_fft(const orc_params *p, const orc_fft *f, const double *fbsk,
                 const uint64_t *lwe_small, const uint64_t *lut, uint64_t *lwe_big_out);
void orc_pbs_exact(const orc_params *p, const uint64_t *bsk_std, const uint64_t *lwe_small,
                   const uint64_t *lut, uint64_t *lwe_big_out);

/* shortint apply_lookup_table on a batch (KS -> PBS), `threads` OS threads over LWEs.
 * luts: n_luts*(k+1)*N, lut_idx: per-LWE index (NULL => 0). exact != 0 selects the exact
 * integer external product (bsk_std used), else the f64 FFT one (fbsk used). */
/* multi-bit PBS (lwe_multi_bit_programmable_bootstrapping.rs); keys = classic routines run on the
 * n/g * 2^g GGSW list whose plaintext bits orc_multi_bit_key_bits returns */
void orc_multi_bit_key_bits(const uint64_t *small_sk, uint32_t n, uint32_t g, uint64_t *out);
void orc_multi_bit_pbs_fft(const orc_params *p, uint32_t g, const orc_fft *f, const double *fbsk,
                           const uint64_t *lwe, const uint64_t *lut, uint64_t *out);
void orc_multi_bit_pbs_exact(const orc_params *p, uint32_t g, const uint64_t *bsk_std, const uint64_t *lwe,
                             const uint64_t *lut, uint64_t *out);
void orc_ks_pbs_batch(const orc_params *p, const uint64_t *ksk, const double *fbsk,
                      const uint64_t *bsk_std, int exact, const uint64_t *lwe_in,
                      const uint32_t *lut_idx, const uint64_t *luts, uint64_t *lwe_out,
                      size_t count, int threads);

/* shortint LUT: table[i] = f(i) for i < msg_mod*carry_mod; returns max f (the degree). */
uint64_t orc_fill_accumulator(const orc_params *p, const uint64_t *table, uint64_t *lut);
/* trivial-ciphertext shortcut (server_key/mod.rs:763-781): body in, body out. */
uint64_t orc_trivial_pbs_body(const orc_params *p, uint64_t body, const uint64_t *lut);

/* ---- harness: deterministic PRNG, keys, encryption, decryption ---- */
typedef struct {   /* sequential reader of one ChaCha20 stream (key = 256-bit seed, nonce = stream id) */
    uint8_t key[32];
    uint64_t stream, counter;
    uint32_t buf[16];
    int pos;
} orc_rng;
void orc_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t stream, uint32_t out[16]);
void orc_rng_init(orc_rng *r, const uint8_t seed[32], uint64_t stream);
uint64_t orc_rng_next(orc_rng *r);
void orc_rng_gaussian_pair(orc_rng *r, double std, double *a, double *b);

void orc_gen_binary_key(const uint8_t seed[32], uint64_t stream, uint64_t *key, size_t len);
void orc_lwe_encrypt(const uint64_t *sk, size_t dim, uint64_t plaintext, double std, orc_rng *r,
                     uint64_t *ct);
uint64_t orc_lwe_decrypt(const uint64_t *sk, size_t dim, const uint64_t *ct);
void orc_glwe_encrypt_assign(const orc_params *p, const uint64_t *glwe_sk, uint64_t *glwe,
                             double std, orc_rng *r);
void orc_gen_ksk(const orc_params *p, const uint64_t *big_sk, const uint64_t *small_sk,
                 const uint8_t seed[32], uint64_t *ksk);
void orc_gen_bsk(const orc_params *p, const uint64_t *small_sk, const uint64_t *glwe_sk,
                 const uint8_t seed[32], uint64_t *bsk_std, int threads);
uint64_t orc_encode(const orc_params *p, uint64_t msg);
uint64_t orc_decode(const orc_params *p, uint64_t plaintext); /* message AND carry */

size_t orc_ksk_len(const orc_params *p);
size_t orc_bsk_len(const orc_params *p);

#ifdef __cplusplus
}
#endif
#endif
