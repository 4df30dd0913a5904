/*
 * fhestr.h -- C ABI of the MI355X-native FheString engine (libfhestr.so).
 *
 * Drop-in boundary for the hot path of the reference (tfhe-rs 0.5.0 fork, paths under
 * /root/reference/tfhe/src): the batched shortint `apply_lookup_table` = LWE keyswitch +
 * programmable bootstrap, the LWE linear ops around it, the integer-layer comparison loops that
 * call it, and the FheString operations built from those.  Conventions follow the reference's own
 * C API (c_api/utils.rs:3-28, c_api/shortint/server_key/pbs.rs:24-85): opaque handles, `int`
 * return (0 = ok), results through out-pointers, caller-owned flat u64 buffers.
 *
 * Ciphertext layouts are the reference's (entities/lwe_ciphertext.rs:598-625,
 * lwe_keyswitch_key.rs:77-108, lwe_bootstrap_key.rs, glwe_ciphertext.rs:210-222):
 *   big LWE    [a_0 .. a_{kN-1}, b]          kN+1 u64
 *   small LWE  [a_0 .. a_{n-1}, b]           n+1 u64
 *   KSK        [kN][ks_level (level l first)][n+1] u64
 *   BSK (std)  [n][pbs_level (level 1 first)][k+1 rows][k+1 polys][N] u64
 *   LUT / accumulator  [k+1][N] u64 (mask polynomials zero)
 *
 * Pointers named d_* are device (HBM) pointers on the engine's GPU; all others are host pointers.
 * An engine is bound to one GPU and one HIP stream.  Every entry point that touches an engine (directly or through
 * one of its plans) takes that engine's lock, so several host threads may share one engine the way they share the
 * reference's `Sync` ServerKey (shortint/engine/mod.rs:23-25,184-189 gives each thread its own scratch; here the
 * calls are serialised instead: use one engine per thread or per rank for concurrency).  The asynchronous *_dev calls
 * only enqueue work under the lock: ordering between threads is the callers' business, as with any shared stream.
 */
#ifndef FHESTR_H
#define FHESTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shortint/parameters/mod.rs:61-76 (ClassicPBSParameters) and multi_bit.rs (MultiBitPBSParameters);
 * native modulus 2^64, KS->PBS order */
typedef struct fhe_params_t {
    uint32_t n, k, N;
    uint32_t pbs_base_log, pbs_level;
    uint32_t ks_base_log, ks_level;
    uint32_t msg_mod, carry_mod;
    double lwe_std, glwe_std;
    /* 0 or 1: classic PBS.  2: multi-bit PBS with that grouping factor (shortint/parameters/multi_bit.rs,
     * core_crypto/algorithms/lwe_multi_bit_programmable_bootstrapping.rs): the bootstrapping key is then
     * n/g * 2^g GGSWs, standard layout [group][selector][level][row][col][N]. */
    uint32_t grouping_factor;
} fhe_params_t;

typedef struct fhe_engine fhe_engine;

/* Last error message of the calling thread ("" if none). */
const char *fhe_last_error(void);
/* Revision tag of the device kernels in this build (profiles/ counter files are keyed on it). */
const char *fhe_kernel_revision(void);

/* ---- engine + keys ------------------------------------------------------------------------ */
/* replaces ServerKey construction (shortint/engine/server_side.rs:54-160), evaluation side only */
int fhe_engine_create(const fhe_params_t *params, int device, fhe_engine **out);
int fhe_engine_destroy(fhe_engine *eng);
int fhe_engine_params(const fhe_engine *eng, fhe_params_t *out);
/* Upload KSK and standard-domain BSK; the BSK is converted on the GPU to the engine's Fourier
 * layout (replaces par_convert_standard_lwe_bootstrap_key_to_fourier,
 * core_crypto/algorithms/lwe_bootstrap_key_conversion.rs:99-152). */
int fhe_engine_load_keys(fhe_engine *eng, const uint64_t *bsk_std, const uint64_t *ksk);
/* Generate KSK and BSK on the device from the secret keys (k*N and n words, each 0 or 1) and install
 * them: replaces ServerKey::new (shortint/engine/server_side.rs:54-160), i.e.
 * allocate_and_generate_new_lwe_keyswitch_key (core_crypto/algorithms/lwe_keyswitch_key_generation.rs:65-130)
 * and par_allocate_and_generate_new_lwe_bootstrap_key (lwe_bootstrap_key_generation.rs:237-300), followed by
 * the Fourier conversion.  Same randomness as fhe_client_gen_server_keys: for one (secret keys, seed)
 * the keys are bit-identical.  bsk_std_out / ksk_out (either may be NULL) receive the standard-domain
 * keys (fhe_params_{bsk,ksk}_len words). */
int fhe_engine_generate_keys(fhe_engine *eng, const uint64_t *glwe_sk, const uint64_t *small_sk,
                             const uint8_t seed[32], uint64_t *bsk_std_out, uint64_t *ksk_out);
/* The engine's HIP stream (hipStream_t) so callers can order their own work against it. */
void *fhe_engine_stream(fhe_engine *eng);
/* Launch on a caller-owned hipStream_t instead (NULL = HIP's default stream), e.g. the framework
 * stream RCCL collectives are enqueued on, so no host synchronisation is needed between a level's
 * kernels and its all-gather.  fhe_engine_reset_stream goes back to the engine's own stream. */
int fhe_engine_set_stream(fhe_engine *eng, void *hip_stream);
int fhe_engine_reset_stream(fhe_engine *eng);
int fhe_engine_synchronize(fhe_engine *eng);
/* Choose the blind-rotation variant: points per thread = 2^log2_points (0 = automatic). */
int fhe_engine_set_variant(fhe_engine *eng, int log2_points);
/* Throughput mode for back-to-back fhe_ks_pbs_batch_dev calls on the engine's own stream (off by default): the
 * keyswitch of call k+1 runs on a second stream, in a 64-register variant whose waves are co-resident with the blind
 * rotation of call k, so it costs (almost) no time of its own: 99.4 k instead of 93.8 k PBS/s on 256-LWE batches of
 * PARAM_MESSAGE_2_CARRY_2.  Results are bit-identical.  Applies to batches of at most one LWE per CU on the
 * register/LDS-resident kernels; other calls run serially as before.
 * ORDERING CONTRACT (differs from the serial mode, where every call is ordered on the engine stream): calls are
 * treated as independent batches.  The engine orders a pipelined call after (a) everything enqueued on the engine
 * stream before the FIRST pipelined call of a run (a run ends with any serial call or synchronisation), (b) an
 * earlier pipelined call whose OUTPUT range overlaps this call's input, (c) the event given through
 * fhe_engine_pipeline_input_event.  Anything else that produces the input -- a kernel or copy the caller enqueued on
 * fhe_engine_stream() between two pipelined calls -- must either be complete on the host or be handed over as such
 * an event; without it the second-stream keyswitch may read the input before it is written.  Outputs are valid
 * after fhe_engine_synchronize.
 * on = 2, "overlapped batches": consecutive calls alternate between two streams and run on the two-LWEs-per-CU kernel, so
 * two batches share the GPU like the two halves of a 512-LWE launch: 256-LWE batches reach the large-batch rate (about
 * 122 k instead of 100 k PBS/s) for the price of each call's latency (4.1 instead of 2.6 ms).  Same ordering contract,
 * plus: a call is also ordered after the previous one if it WRITES what that call reads or writes.  Results are the
 * large-batch kernel's: decrypt-identical to the serial ones and within the same noise, not bit-identical to mode 0 / 1
 * (two f64 transform schedules round differently); bit-identical to serial calls of more than one LWE per CU. */
int fhe_engine_set_pipeline(fhe_engine *eng, int on);
/* One shot: the keyswitch of the NEXT pipelined fhe_ks_pbs_batch_dev call waits for `hip_event` (a hipEvent_t the caller
 * recorded behind the work that produces that call's input, on any stream).  Ignored by serial calls. */
int fhe_engine_pipeline_input_event(fhe_engine *eng, void *hip_event);
/* Multi-bit PBS only: batches of up to max_batch LWEs build every (LWE, group) GGSW on the whole GPU first
 * (prepare_multi_bit_ggsw, lwe_multi_bit_programmable_bootstrapping.rs:18-83, which the reference runs on
 * separate threads) and then rotate against them; larger batches fuse both into one kernel.  Default 64
 * (env FHESTR_MULTIBIT_COMBINE_MAX); 0 = always fused.  Both paths give bit-identical ciphertexts.  Costs
 * max_batch * (n / grouping_factor) * (k+1)^2 * N * 8 bytes of device workspace when used. */
int fhe_engine_set_multibit_combine_max(fhe_engine *eng, uint32_t max_batch);
/* Polynomial sizes N >= 16384 (PARAM_MESSAGE_4_CARRY_4 ...): the blind rotation of one LWE is spread over a
 * cluster of compute units of one XCD that exchange the four-step transform's matrices through that XCD's L2:
 * all 32 CUs of the XCD with two LWEs in flight per XCD for N = 32768 with two levels (round 4), 8 / 4 CUs
 * (N = 32768 / 16384) otherwise.  mode: -1 automatic (default; env FHESTR_CLUSTER), 0 never (one workgroup per
 * LWE through an HBM workspace), 1 always, 2 always and with round 3's 8-CU clusters even where the whole-XCD
 * kernel exists; max_batch: in automatic mode, batches above it take the one-workgroup kernel.  Automatic mode
 * picks the whole-XCD kernel for up to 16 LWEs (one LWE: 12.3 ms against 20.9 ms) and the 8-CU clusters above
 * (four LWEs in flight per XCD: 1.15 k PBS/s at 256 LWEs against 0.85 k).  All kernels read the same Fourier key
 * and give decrypt-identical results. */
int fhe_engine_set_cluster_mode(fhe_engine *eng, int mode, uint32_t max_batch);
/* Latency knob for dependent chains of small batches (the trailing levels of a string comparison): with `on`, a
 * blind-rotation launch that would leave more than half of the compute units idle carries replicas of its workgroups
 * on them (they recompute an LWE of the batch and store nothing).  Results are unchanged; the part keeps drawing power
 * and therefore keeps its clock for the large launch that follows -- otherwise that launch starts 7 % low in shader
 * clock and needs about 14 ms to ramp back (profiles/r03_after_idle.txt).  Costs the energy of the replicas; off by
 * default (FHESTR_KEEP_BUSY=1 turns it on at engine creation).  No counterpart in the reference. */
int fhe_engine_set_keep_busy(fhe_engine *eng, int on);
/* After a synchronisation: clusters the last cluster launch formed (0 if none ran). */
int fhe_engine_cluster_info(fhe_engine *eng, uint32_t *clusters);
/* The multi-CU kernels need their whole grid resident at once; when another kernel holds compute units for too long
 * the launch drains with a status instead of hanging, and the engine runs the same batch again on the one-workgroup
 * kernel before the call returns (correct results, that launch's time lost).  count = how often this engine did so. */
int fhe_engine_cluster_fallbacks(fhe_engine *eng, uint32_t *count);

/* ---- lookup tables ------------------------------------------------------------------------- */
/* generate_lookup_table (shortint/server_key/mod.rs:383-399, engine/mod.rs:72-128):
 * table[i] = f(i), i < msg_mod*carry_mod.  Returns the LUT id and its degree (max f). */
int fhe_lut_generate(fhe_engine *eng, const uint64_t *table, uint32_t *lut_id, uint64_t *degree);
/* Upload an already-built accumulator ((k+1)*N u64). */
int fhe_lut_upload(fhe_engine *eng, const uint64_t *accumulator, uint32_t *lut_id);
/* Read an accumulator back ((k+1)*N u64). */
int fhe_lut_download(fhe_engine *eng, uint32_t lut_id, uint64_t *accumulator);
int fhe_lut_count(const fhe_engine *eng, uint32_t *count);

/* ---- the hot path -------------------------------------------------------------------------- */
/* keyswitch_lwe_ciphertext (core_crypto/algorithms/lwe_keyswitch.rs:96-170), batched. */
int fhe_keyswitch_batch(fhe_engine *eng, const uint64_t *lwe_big_in, uint64_t *lwe_small_out,
                        uint32_t count);
/* programmable_bootstrap_lwe_ciphertext_mem_optimized
 * (core_crypto/algorithms/lwe_programmable_bootstrapping.rs:1067-1111), batched; lut_idx may be
 * NULL (LUT 0 for all). */
int fhe_pbs_batch(fhe_engine *eng, const uint64_t *lwe_small_in, const uint32_t *lut_idx,
                  uint64_t *lwe_big_out, uint32_t count);
/* apply_lookup_table / keyswitch_programmable_bootstrap_assign
 * (shortint/server_key/mod.rs:457-476,783-857), batched, host buffers. */
int fhe_ks_pbs_batch(fhe_engine *eng, const uint64_t *lwe_big_in, const uint32_t *lut_idx,
                     uint64_t *lwe_big_out, uint32_t count);
/* Small-key order: programmable_bootstrap_keyswitch_assign (shortint/server_key/mod.rs:859-932, the
 * *_PBS_KS parameter sets): small LWEs in, bootstrap, keyswitch, small LWEs out. */
int fhe_pbs_ks_batch(fhe_engine *eng, const uint64_t *lwe_small_in, const uint32_t *lut_idx,
                     uint64_t *lwe_small_out, uint32_t count);
/* Same with everything resident in HBM; asynchronous on the engine stream. */
int fhe_ks_pbs_batch_dev(fhe_engine *eng, const uint64_t *d_lwe_big_in, const uint32_t *d_lut_idx,
                         uint64_t *d_lwe_big_out, uint32_t count);
/* LWE linear combinations (unchecked_add / scalar_mul / scalar_add / bivariate packing,
 * shortint/server_key/add.rs:520-524, scalar_mul.rs:206-208, scalar_add.rs:211-218,
 * bivariate_pbs.rs:167-182): out[j] = sum_{t in [off[j],off[j+1])} coeff[t]*pool[src[t]],
 * then body += cst[j].  Host buffers. */
int fhe_lwe_lincomb_batch(fhe_engine *eng, const uint64_t *pool, uint32_t pool_count,
                          const uint32_t *off, const uint32_t *src, const int32_t *coeff,
                          const uint64_t *cst, uint64_t *out, uint32_t jobs);
/* Kernel-only timing of the last fhe_ks_pbs_batch*_ call (HIP events on the engine stream):
 * ms[0] = keyswitch, ms[1] = blind rotation + sample extraction. Synchronises. */
int fhe_last_kernel_ms(fhe_engine *eng, float ms[2]);
/* Sum of the kernel durations of the (up to 1024) fhe_ks_pbs_batch* calls recorded since the last
 * reset, measured with HIP events on the engine stream: total_ms[0] keyswitch, total_ms[1] blind
 * rotation; *calls = number of calls summed.  Synchronises; reset != 0 clears the record. */
int fhe_kernel_times(fhe_engine *eng, double total_ms[2], uint32_t *calls, int reset);

/* ---- plans: levelised shortint circuits ----------------------------------------------------- */
/* A plan records shortint operations (LWE linear combinations and apply_lookup_table) as a DAG and
 * executes them level by level, one batched KS+PBS launch per level.  It is the batched replacement
 * of the per-block rayon loops of the reference's integer layer
 * (integer/server_key/radix_parallel/comparison.rs:10-83, scalar_comparison.rs:104-558).
 * With `world` > 1 every PBS has an owner rank; ranks exchange only what another rank consumes (one
 * all-gather per level that exports anything, see fhe_plan_level_info). */
typedef struct fhe_plan fhe_plan;
int fhe_plan_create(fhe_engine *eng, fhe_plan **out);
/* Engine-less plan: can be built, finalised and exported (levels, LUT accumulators) on a host
 * without a GPU -- used by planners and checkers; every run entry point refuses it. */
int fhe_plan_create_offline(const fhe_params_t *params, fhe_plan **out);
int fhe_plan_destroy(fhe_plan *plan);
/* building (before fhe_plan_finalize) */
int fhe_plan_input(fhe_plan *plan, uint64_t degree, uint32_t *node);
int fhe_plan_lut(fhe_plan *plan, const uint64_t *table, uint32_t *lut);    /* generate_lookup_table */
int fhe_plan_lin(fhe_plan *plan, const uint32_t *nodes, const int32_t *coeffs, uint32_t n_terms,
                 int64_t constant, uint32_t *node);                        /* unchecked add/scalar ops */
int fhe_plan_pbs(fhe_plan *plan, uint32_t src, uint32_t lut, uint32_t *node); /* apply_lookup_table */
/* Value ranges are tracked two-sided: fhe_plan_pbs refuses an input that may be negative (it would
 * wrap into the padding bit, shortint/engine/client_side.rs:66-74); fhe_plan_pbs_signed declares that
 * the padding bit is used on purpose and the table is read through its negacyclic extension
 * f(x - msg*carry) = -f(x).  Both refuse inputs above the parameter set's noise budget (in units of one
 * nominal ciphertext variance; MaxNoiseLevel::validate, shortint/ciphertext/mod.rs:28-55). */
int fhe_plan_pbs_signed(fhe_plan *plan, uint32_t src, uint32_t lut, uint32_t *node);
/* Reduction of msg*carry = T bits in one lookup: `src` is a sum with value range [0, T]; the result is the bit
 * (sum == T) if `all`, else (sum != 0).  The reference's are_all_comparisons_block_true /
 * is_at_least_one_comparisons_block_true (integer/server_key/radix_parallel/scalar_comparison.rs:147-233) take T - 1
 * bits per lookup; sum = T is the padding bit, which a table with entries -/+ delta/2 answers consistently (csrc/circuit.h).
 * One level less for AND over a 16-char pattern and OR over up to 256 offsets under PARAM_MESSAGE_2_CARRY_2. */
int fhe_plan_pbs_full_box(fhe_plan *plan, uint32_t src, int all, uint32_t *node);
int fhe_plan_set_noise_budget(fhe_plan *plan, double budget);   /* <= 0: no check */
/* PBS nodes created from now on run on `rank` (-1: automatic).  Used with world > 1 to keep a slice of
 * the work and its first reduction levels on one GPU. */
int fhe_plan_set_owner_hint(fhe_plan *plan, int rank);
int fhe_plan_output(fhe_plan *plan, uint32_t node);
int fhe_plan_finalize(fhe_plan *plan, uint32_t world);
/* info[6] = {n_inputs, n_outputs, n_levels, n_pbs, pool_slots, world} */
int fhe_plan_info(const fhe_plan *plan, uint32_t info[6]);
/* info[4] = {largest PBS-input noise in the plan (nominal variances), the budget, log2 of the modelled
 * failure probability of that worst PBS, ciphertexts a rank receives over all all-gathers} */
int fhe_plan_noise_info(const fhe_plan *plan, double info[4]);
/* out[6] = {V_pbs, V_ks, V_ms (variances, torus = 1), delta/2, default budget, log2 p_fail at it} */
int fhe_noise_model(const fhe_params_t *params, double out[6]);
/* Page-locked host memory (hipHostMalloc).  Every *_host entry point and every FheString call takes plain host
 * pointers; pageable ones move at 3-10 GB/s and page-fault on first touch, buffers from fhe_host_alloc move at the
 * full PCIe rate -- worth it for 1024-char strings under PARAM_MESSAGE_4_CARRY_4 (268 MB each way).  The
 * reference has no counterpart (its ciphertexts are Vec<u64> on the heap, entities/lwe_ciphertext.rs:598-625).
 * fhe_host_free(NULL) is a no-op. */
int fhe_host_alloc(size_t bytes, void **out);
int fhe_host_free(void *ptr);
/* 1 if the model's PBS-output variance has been checked against this engine's measured noise for the shape
 * (N, k, level, grouping factor) of `params`; shapes that have not carry a 4x safety factor in the budget. */
int fhe_noise_model_is_calibrated(const fhe_params_t *params);
/* Would fhe_engine_create accept these parameters?  0 = yes; 1 = no, reason in fhe_last_error.  Needs no device: the
 * same checks fhe_engine_create runs before it touches one (a blind-rotation kernel instantiated for (N, k, level,
 * grouping factor); decomposition ranges of the keyswitch -- any level count, more than 16 levels take the byte-plane
 * kernel instead of the matrix-core one).  Covers every ClassicPBSParameters / MultiBitPBSParameters constant of
 * shortint/parameters/{mod,multi_bit,parameters_compact_pk}.rs (tests/test_parameter_tables.py). */
int fhe_params_supported(const fhe_params_t *params);
/* Pool layout of a level.  Every rank runs its own jobs [job_lo, job_hi) of the level (rank_info) and
 * writes job job_lo + i to pool slot local_base + i; if e_max > 0 the level ends with an all-gather of
 * the first e_max slots of every rank's local region into [recv_base, recv_base + world * e_max).
 * info[8] = {jobs, local_base, local_size, e_max, recv_base, n_terms, 0, 0}; level == n_levels describes
 * the output gather (jobs, n_terms only). */
int fhe_plan_level_info(const fhe_plan *plan, uint32_t level, uint32_t info[8]);
/* info[3] = {job_lo, job_hi, n_export} */
int fhe_plan_level_rank_info(const fhe_plan *plan, uint32_t level, uint32_t rank, uint32_t info[3]);
/* CSR description of a level in job order (any pointer may be NULL): off[jobs+1], src/coeff[n_terms]
 * (src = pool slot), cst[jobs] (already scaled by delta), lut[jobs] */
int fhe_plan_export_level(const fhe_plan *plan, uint32_t level, uint32_t *off, uint32_t *src,
                          int32_t *coeff, uint64_t *cst, uint32_t *lut);
/* plan-local LUT ids (the `lut` arrays above) and their accumulators ((k+1)*N u64) */
int fhe_plan_lut_count(const fhe_plan *plan, uint32_t *count);
int fhe_plan_export_lut(const fhe_plan *plan, uint32_t lut, uint64_t *accumulator);
/* single GPU, host buffers: inputs n_inputs x (kN+1), outputs n_outputs x (kN+1) */
int fhe_plan_run(fhe_plan *plan, const uint64_t *inputs, uint64_t *outputs);
/* `instances` independent copies of the plan (finalised for world 1) in ONE pass: level l of all instances is one gather +
 * one keyswitch + one blind-rotation launch over instances x jobs(l) ciphertexts.  This is the reference's throughput shape
 * -- many independent inputs per call (benches/core_crypto/pbs_bench.rs:430-549; rayon over the blocks of an integer,
 * integer/server_key/radix_parallel/comparison.rs:22-28) -- for whole operations: a single FheString::eq pays four dependent
 * single-PBS latencies on a nearly idle GPU (12.8 ms), 32 of them share those four and run at the engine's batch rate.
 * inputs: [instances][n_inputs] ciphertexts, outputs: [instances][n_outputs] (host arrays; _dev: device arrays, ordered on
 * the engine's stream, no host synchronisation).  Every instance's outputs are those of fhe_plan_run on its inputs (the same
 * gathers and tables; batches beyond one LWE per CU run the two-LWEs-per-CU kernel: decrypt-identical, not bit-identical).
 * With several GPUs the instances shard over the ranks -- no collective at all. */
int fhe_plan_run_batch(fhe_plan *plan, uint32_t instances, const uint64_t *inputs, uint64_t *outputs);
int fhe_plan_run_batch_dev(fhe_plan *plan, uint32_t instances, const uint64_t *d_inputs, uint64_t *d_outputs);
/* multi-GPU building blocks (device pool of pool_slots big LWEs; inputs live in slots [0, n_inputs) on
 * every rank): run `rank`'s jobs of one level; gather the outputs once the last level is through. */
int fhe_plan_run_level_rank_dev(fhe_plan *plan, uint64_t *d_pool, uint32_t level, uint32_t rank);
int fhe_plan_gather_outputs_dev(fhe_plan *plan, const uint64_t *d_pool, uint64_t *d_out);

/* ---- radix-integer operations (the reference's integer layer) -------------------------------- */
/* An unsigned radix integer = n_blocks big-key LWE blocks, little endian, log2(msg_mod) bits each
 * (integer/block_decomposition.rs:119-144).  A plan for one operation, run with fhe_plan_run (inputs in
 * the order given) or sharded like any plan:
 *   "add" "sub"                       a, b -> n_blocks blocks (wrapping): unchecked add + parallel one-carry
 *                                     propagation + message_extract (radix_parallel/add.rs:487-624,724-772)
 *   "scalar_add" "scalar_sub"         a, clear `scalar` (radix_parallel/scalar_add.rs:204-222)
 *   "message_extract" "carry_extract" n_blocks blocks with full carries -> their messages / carries
 *   "cmux"                            cond (0/1 block), t, f -> cond ? t : f (radix_parallel/cmux.rs:194-316)
 *   "eq" "ne" "gt" "ge" "lt" "le"     a, b -> one 0/1 block (comparator.rs:193-280, scalar_comparison.rs:147-233)
 *   "scalar_eq" ... "scalar_le"       a, clear `scalar` (at most 64 bits) */
int fhe_int_plan_create(fhe_engine *eng, const char *op, uint32_t n_blocks, uint64_t scalar, uint32_t world,
                        fhe_plan **out);
int fhe_int_plan_create_offline(const fhe_params_t *params, const char *op, uint32_t n_blocks, uint64_t scalar,
                                uint32_t world, fhe_plan **out);

/* ---- FheString operations --------------------------------------------------------------------- */
/* An encrypted string = `cap` characters, zero padded, each character 8/log2(msg_mod) big-key LWE
 * blocks, little endian (integer/block_decomposition.rs:119-144): cap * blocks * (kN+1) u64.
 * Results decrypt to what the clear-text function gives on the unpadded ASCII strings.  The zero characters come
 * after the text and nowhere else (the searches with an encrypted pattern rely on it: csrc/fhe_string.cpp, group_match).
 * op in {"eq","ne","starts_with","ends_with","contains","find"} (+ "_clear" suffix for a clear
 * pattern) or {"to_upper","to_lower","trim_start","trim_end","strip","replace","replace_clear","concat",
 * "concat_clear","repeat_clear"}.  Outputs: one 0/1 block; find: found block then
 * ceil(log_msg_mod(cap+1)) index digits (little endian); case ops: the whole string. */
int fhe_str_plan_create(fhe_engine *eng, const char *op, uint32_t a_cap, uint32_t b_cap,
                        const uint8_t *clear, uint32_t clear_len, uint32_t world, fhe_plan **out);
int fhe_str_plan_create_offline(const fhe_params_t *params, const char *op, uint32_t a_cap,
                                uint32_t b_cap, const uint8_t *clear, uint32_t clear_len,
                                uint32_t world, fhe_plan **out);
#define FHE_STR_BINARY_DECL(name)                                                                    \
    int fhe_str_##name(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint64_t *b,       \
                       uint32_t b_cap, uint64_t *out);                                               \
    int fhe_str_##name##_clear(fhe_engine *eng, const uint64_t *a, uint32_t a_cap,                  \
                               const uint8_t *pat, uint32_t pat_len, uint64_t *out);
FHE_STR_BINARY_DECL(eq)
FHE_STR_BINARY_DECL(ne)
FHE_STR_BINARY_DECL(starts_with)
FHE_STR_BINARY_DECL(ends_with)
FHE_STR_BINARY_DECL(contains)
FHE_STR_BINARY_DECL(find)
FHE_STR_BINARY_DECL(rfind)            /* last occurrence; same outputs as find */
FHE_STR_BINARY_DECL(eq_ignore_case)   /* ASCII case folding on both sides, then eq */
FHE_STR_BINARY_DECL(lt)               /* lexicographic (byte-wise) order, like Rust's str / Python's bytes */
FHE_STR_BINARY_DECL(le)
FHE_STR_BINARY_DECL(gt)
FHE_STR_BINARY_DECL(ge)
/* concat: a followed by b (a's padding removed); out = (a_cap + b_cap) * blocks LWEs, resp. a_cap + pat_len
 * for the clear form.  Plan op names "concat" / "concat_clear". */
FHE_STR_BINARY_DECL(concat)
/* repeat: a repeated `count` (clear, 1..255) times; out = count * a_cap * blocks LWEs.  Plan op name
 * "repeat_clear" with the count as the one clear byte. */
int fhe_str_repeat_clear(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint32_t count, uint64_t *out);
/* whitespace = ASCII 9..13 and 32; results are re-padded with zeros (whole string returned) */
int fhe_str_trim_start(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
int fhe_str_trim_end(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
int fhe_str_strip(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
/* replace every leftmost non-overlapping occurrence of `from` by `to` (Rust's str::replace / Python's
 * bytes.replace).
 * Equal-length forms (the string keeps its shape and is rewritten in place): from_to = pat_cap chars of
 * `from` then pat_cap chars of `to`, neither padded.  Plan op names: "replace" (b_cap = 2*pat_cap) /
 * "replace_clear" (clear = from||to).
 * General forms: any lengths, the result has `out_cap` characters (longer results are cut, so give
 * a_cap + max(0, |to| - |from|) * (a_cap / |from|) to be safe).  Encrypted `from` / `to` may be zero
 * padded (hidden lengths; to_cap == 0 deletes); an encrypted `from` that decrypts to the empty string
 * replaces nothing, a clear empty `from` inserts `to` before every character and at the end like Rust and
 * Python do.  Plan op names "replace:<from_cap>:<out_cap>" (b = from || to) and
 * "replace_clear:<from_len>:<out_cap>" (clear = from || to). */
int fhe_str_replace_general(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint64_t *from,
                            uint32_t from_cap, const uint64_t *to, uint32_t to_cap, uint32_t out_cap,
                            uint64_t *out);
int fhe_str_replace_clear_general(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint8_t *from,
                                  uint32_t from_len, const uint8_t *to, uint32_t to_len, uint32_t out_cap,
                                  uint64_t *out);
int fhe_str_replace(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint64_t *from_to,
                    uint32_t pat_cap, uint64_t *out);
int fhe_str_replace_clear(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint8_t *from,
                          const uint8_t *to, uint32_t pat_len, uint64_t *out);
/* Many strings against ONE second operand in one pass (fhe_plan_run_batch on the operation's cached plan): `op` is an
 * operation name of fhe_str_plan_create ("eq", "ne", "contains", "find", "starts_with", "to_lower", ...; "<op>_clear" with
 * a clear pattern), rows = [count][a_cap * blocks] ciphertexts, b = the shared encrypted operand ([b_cap * blocks]
 * ciphertexts, b_cap = 0 and NULL for unary operations and clear patterns), out = [count][n_outputs] ciphertexts,
 * *n_outputs (optional) = outputs per row; with out == NULL the call only builds the plan and returns that number.  One pattern against 32 rows: FheString::eq 12.8 ms alone, under 5 ms per row. */
int fhe_str_op_many(fhe_engine *eng, const char *op, const uint64_t *rows, uint32_t a_cap, uint32_t count, const uint64_t *b,
                    uint32_t b_cap, const uint8_t *clear, uint32_t clear_len, uint64_t *out, uint32_t *n_outputs);
/* len: ceil(log_msg_mod(cap+1)) little-endian digits; is_empty: one 0/1 block */
int fhe_str_len(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
int fhe_str_is_empty(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
/* strip_prefix / strip_suffix with a clear pattern: out = 1 + cap*blocks LWEs: first a 0/1 block
 * ("the pattern was there and has been removed"), then the resulting string */
int fhe_str_strip_prefix_clear(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint8_t *pat,
                               uint32_t pat_len, uint64_t *out);
int fhe_str_strip_suffix_clear(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint8_t *pat,
                               uint32_t pat_len, uint64_t *out);
/* the same with an encrypted, zero padded pattern (hidden length) */
int fhe_str_strip_prefix(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint64_t *pat,
                         uint32_t pat_cap, uint64_t *out);
int fhe_str_strip_suffix(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, const uint64_t *pat,
                         uint32_t pat_cap, uint64_t *out);
int fhe_str_to_upper(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);
int fhe_str_to_lower(fhe_engine *eng, const uint64_t *a, uint32_t a_cap, uint64_t *out);

/* ---- client side (CPU): keys, encryption, decryption ---------------------------------------- */
/* ClientKey::new / encrypt / decrypt_message_and_carry / ServerKey::new of the reference
 * (shortint/engine/client_side.rs:13-128, shortint/client_key/mod.rs:281-337,
 * shortint/engine/server_side.rs:54-160).
 * Randomness: all of it -- secret keys, masks, noise -- is ChaCha20 keystream under the 256-bit `seed`
 * (one stream per purpose and key row; stands in for the reference's AES-128-CTR concrete-csprng with
 * forked generators).  Published masks are PRF output and reveal neither the seed nor the noise.  Take
 * the seed from fhe_random_seed (the OS CSPRNG) in production; a fixed seed reproduces every key and
 * ciphertext bit for bit and is for tests only.  The seed is as secret as the secret keys. */
typedef struct fhe_client_key fhe_client_key;
int fhe_random_seed(uint8_t seed[32]);
/* the block function itself (known-answer tests): out = ChaCha20(key; words 12,13 = counter, 14,15 = stream) */
int fhe_chacha20_block(const uint8_t key[32], uint64_t counter, uint64_t stream, uint32_t out[16]);
size_t fhe_params_ksk_len(const fhe_params_t *p);
size_t fhe_params_bsk_len(const fhe_params_t *p);
int fhe_client_key_create(const fhe_params_t *params, const uint8_t seed[32], fhe_client_key **out);
int fhe_client_key_destroy(fhe_client_key *ck);
/* msgs[i] in [0, msg_mod*carry_mod); cts: count x (kN+1) u64 (big-key encryption, glwe noise). */
int fhe_client_encrypt(fhe_client_key *ck, const uint64_t *msgs, uint32_t count, uint64_t *cts);
/* message-and-carry decode of every ciphertext. */
int fhe_client_decrypt(fhe_client_key *ck, const uint64_t *cts, uint32_t count, uint64_t *msgs);
/* Standard-domain BSK + KSK for fhe_engine_load_keys (sizes: fhe_params_{bsk,ksk}_len). */
int fhe_client_gen_server_keys(fhe_client_key *ck, uint64_t *bsk_std, uint64_t *ksk, int threads);
/* Copy out the secret keys (either pointer may be NULL): glwe_sk k*N u64, small_sk n u64. */
int fhe_client_secret_keys(fhe_client_key *ck, uint64_t *glwe_sk, uint64_t *small_sk);

/* ---- tfhe-rs wire format (serde + bincode 1.x, fixed-width little endian) ---------------------- */
/* Byte forms of core_crypto's LweCiphertext<Vec<u64>>, LweKeyswitchKey<Vec<u64>>, standard-domain
 * LweBootstrapKey<Vec<u64>> and shortint::Ciphertext as tfhe-rs 0.5 writes them with bincode::serialize /
 * safe_serialize (entities/lwe_ciphertext.rs:500-507, lwe_keyswitch_key.rs:76-86,
 * lwe_bootstrap_key.rs:98-106 + ggsw_ciphertext_list.rs:9-20, commons/ciphertext_modulus.rs:41-64,
 * shortint/ciphertext/mod.rs:261-270, safe_deserialization.rs:16-99).  Native modulus 2^64 only.
 * Writers: `out` may be NULL to query the size; *written receives the byte count either way.
 * Readers validate every dimension against the parameter set (the reference's ParameterSetConformant)
 * and never read past in_len.  The reference ships no serialized fixture: parity unpinned. */
typedef struct fhe_shortint_meta {
    uint64_t degree, noise_level, message_modulus, carry_modulus;
    uint32_t pbs_order;   /* 0 = KeyswitchBootstrap, 1 = BootstrapKeyswitch (commons/parameters.rs:233-245) */
} fhe_shortint_meta;
int fhe_wire_write_lwe_ciphertext(const uint64_t *ct, size_t lwe_size, uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_lwe_ciphertext(const uint8_t *in, size_t in_len, uint64_t *ct, size_t ct_cap, size_t *lwe_size,
                                 size_t *consumed);
int fhe_wire_write_keyswitch_key(const fhe_params_t *p, const uint64_t *ksk, uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_keyswitch_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint64_t *ksk, size_t *consumed);
int fhe_wire_write_bootstrap_key(const fhe_params_t *p, const uint64_t *bsk_std, uint8_t *out, size_t out_cap,
                                 size_t *written);
int fhe_wire_read_bootstrap_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint64_t *bsk_std,
                                size_t *consumed);
/* safe_framing != 0: the version / type-name header of safe_serialize; size_limit (0 = none) bounds the
 * object's bytes like safe_deserialize's serialized_size_limit */
int fhe_wire_write_shortint_ciphertext(const uint64_t *ct, size_t lwe_size, const fhe_shortint_meta *meta,
                                       int safe_framing, uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_shortint_ciphertext(const uint8_t *in, size_t in_len, int safe_framing, uint64_t size_limit,
                                      uint64_t *ct, size_t ct_cap, size_t *lwe_size, fhe_shortint_meta *meta,
                                      size_t *consumed);

/* ---- seeded ("compressed") server keys -------------------------------------------------------
 * What a tfhe-rs client sends: shortint CompressedServerKey = SeededLweKeyswitchKey + SeededLweBootstrapKey
 * (or SeededLweMultiBitBootstrapKey), shortint/server_key/compressed.rs.  A seeded key holds the bodies only
 * (keyswitch key: k*N*ks_level words; bootstrap key: n_ggsw*pbs_level*(k+1) polynomials of N words) and a
 * 128-bit compression seed; the masks are the seed's AES-128-CTR stream (concrete-csprng) in storage order
 * (seeded_lwe_keyswitch_key_decompression.rs, seeded_lwe_bootstrap_key_decompression.rs,
 * seeded_lwe_multi_bit_bootstrap_key_decompression.rs; details in csrc/seeded_keys.cpp).
 * fhe_seeded_decompress_* rebuild the standard-domain keys fhe_engine_load_keys takes; fhe_seeded_split_* are
 * the inverse (bodies of a standard key); fhe_seeded_mask_words exposes the mask stream (tests, and clients that
 * want to ENCRYPT seeded keys).  The AES block function carries the FIPS-197 vector the reference tests with;
 * stream position, integer packing and draw order follow the reference's sources: parity with a real client
 * is unpinned (no seeded fixture in the reference). */
/* The device-side route: upload the bodies only and expand both mask streams on the GPU (csrc/seeded_kernels.hip.h),
 * then install like fhe_engine_load_keys.  bsk_std_out / ksk_out (may be NULL) receive the standard-domain keys --
 * bit-identical to fhe_seeded_decompress_*. */
int fhe_engine_load_seeded_keys(fhe_engine *eng, const uint8_t ksk_seed[16], const uint64_t *ksk_bodies,
                                const uint8_t bsk_seed[16], const uint64_t *bsk_bodies, uint64_t *bsk_std_out,
                                uint64_t *ksk_out);
/* Compressed ciphertexts (shortint CompressedCiphertext: a body and a compression seed EACH, 92 bytes on the wire
 * instead of 8 (kN+1); shortint/ciphertext/mod.rs:471-478, seeded_lwe_ciphertext_decompression.rs): expanded on the
 * GPU into d_out (device, count x (kN+1) words; may be NULL) and / or host_out.  fhe_seeded_decompress_lwe_batch is
 * the host-side twin (any LWE dimension). */
int fhe_engine_expand_seeded_lwe(fhe_engine *eng, const uint8_t *seeds /* [count][16] */, const uint64_t *bodies,
                                 uint32_t count, uint64_t *d_out, uint64_t *host_out);
int fhe_seeded_decompress_lwe_batch(uint32_t lwe_dim, const uint8_t *seeds, const uint64_t *bodies, uint32_t count,
                                    uint64_t *out);
int fhe_wire_write_compressed_ciphertext(uint64_t body, size_t lwe_size, const uint8_t seed[16], const fhe_shortint_meta *meta,
                                         uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_compressed_ciphertext(const uint8_t *in, size_t in_len, uint64_t *body, size_t *lwe_size, uint8_t seed[16],
                                        fhe_shortint_meta *meta, size_t *consumed);
/* integer RadixCiphertext / CompressedRadixCiphertext: Vec of blocks, least significant first
 * (integer/ciphertext/mod.rs:18-21,30,45) -- an FheUint8 character is four blocks under PARAM_MESSAGE_2_CARRY_2. */
int fhe_wire_write_radix_ciphertext(const uint64_t *cts, size_t lwe_size, const fhe_shortint_meta *metas, size_t n_blocks,
                                    uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_radix_ciphertext(const uint8_t *in, size_t in_len, uint64_t *cts, size_t lwe_size, size_t max_blocks,
                                   fhe_shortint_meta *metas, size_t *n_blocks, size_t *consumed);
int fhe_wire_write_compressed_radix_ciphertext(const uint64_t *bodies, const uint8_t *seeds /* [n][16] */, size_t lwe_size,
                                               const fhe_shortint_meta *metas, size_t n_blocks, uint8_t *out, size_t out_cap,
                                               size_t *written);
int fhe_wire_read_compressed_radix_ciphertext(const uint8_t *in, size_t in_len, uint64_t *bodies, uint8_t *seeds,
                                              size_t *lwe_size, size_t max_blocks, fhe_shortint_meta *metas,
                                              size_t *n_blocks, size_t *consumed);
int fhe_aes128_encrypt_block(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]);
int fhe_seeded_mask_words(const uint8_t seed[16], uint64_t *out, size_t count);
int fhe_seeded_decompress_keyswitch_key(const fhe_params_t *p, const uint8_t seed[16], const uint64_t *bodies, uint64_t *ksk);
int fhe_seeded_decompress_bootstrap_key(const fhe_params_t *p, const uint8_t seed[16], const uint64_t *bodies,
                                        uint64_t *bsk_std);
int fhe_seeded_split_keyswitch_key(const fhe_params_t *p, const uint64_t *ksk, uint64_t *bodies);
int fhe_seeded_split_bootstrap_key(const fhe_params_t *p, const uint64_t *bsk_std, uint64_t *bodies);
/* bincode forms (entities/seeded_lwe_keyswitch_key.rs:11-21, seeded_ggsw_ciphertext_list.rs:12-23,
 * seeded_lwe_multi_bit_bootstrap_key.rs:16-25, lwe_multi_bit_bootstrap_key.rs:11-20); the bootstrap-key
 * functions read / write the multi-bit container when the parameter set has a grouping factor */
int fhe_wire_write_seeded_keyswitch_key(const fhe_params_t *p, const uint8_t seed[16], const uint64_t *bodies, uint8_t *out,
                                        size_t out_cap, size_t *written);
int fhe_wire_read_seeded_keyswitch_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint8_t seed[16],
                                       uint64_t *bodies, size_t *consumed);
int fhe_wire_write_seeded_bootstrap_key(const fhe_params_t *p, const uint8_t seed[16], const uint64_t *bodies, uint8_t *out,
                                        size_t out_cap, size_t *written);
int fhe_wire_read_seeded_bootstrap_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint8_t seed[16],
                                       uint64_t *bodies, size_t *consumed);
/* shortint CompressedServerKey, the object a client serializes for the server (shortint/server_key/compressed.rs:11-17,
 * 44-55): seeded keyswitch key, Classic / MultiBit seeded bootstrap key (by the parameter set's grouping factor),
 * message and carry modulus (checked against the parameter set), max_degree, ciphertext modulus, pbs_order. */
int fhe_wire_write_compressed_server_key(const fhe_params_t *p, const uint8_t ksk_seed[16], const uint64_t *ksk_bodies,
                                         const uint8_t bsk_seed[16], const uint64_t *bsk_bodies, uint64_t max_degree,
                                         uint32_t pbs_order, uint8_t *out, size_t out_cap, size_t *written);
int fhe_wire_read_compressed_server_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint8_t ksk_seed[16],
                                        uint64_t *ksk_bodies, uint8_t bsk_seed[16], uint64_t *bsk_bodies,
                                        uint64_t *max_degree, uint32_t *pbs_order, size_t *consumed);
int fhe_wire_write_multi_bit_bootstrap_key(const fhe_params_t *p, const uint64_t *bsk_std, uint8_t *out, size_t out_cap,
                                           size_t *written);
int fhe_wire_read_multi_bit_bootstrap_key(const fhe_params_t *p, const uint8_t *in, size_t in_len, uint64_t *bsk_std,
                                          size_t *consumed);

#ifdef __cplusplus
}
#endif
#endif
