// circuit.h -- levelised batches of shortint operations ("plan").
//
// The reference evaluates integer / string operations one shortint block at a time, each
// `apply_lookup_table` being a full KS+PBS on one ciphertext, with rayon over the blocks of one
// integer (integer/server_key/radix_parallel/comparison.rs:22-28, scalar_comparison.rs:167-184).
// Here an operation is first recorded as a DAG of two node kinds
//   LIN  -- linear combination of ciphertexts plus a clear constant (unchecked_add_assign,
//           unchecked_scalar_mul_assign, unchecked_scalar_add_assign, bivariate packing:
//           shortint/server_key/add.rs:520-524, scalar_mul.rs:206-208, scalar_add.rs:211-218,
//           bivariate_pbs.rs:167-182)
//   PBS  -- apply_lookup_table on a node (shortint/server_key/mod.rs:457-476)
// then levelised: every PBS whose inputs are ready forms one batch = one lincomb launch + one
// keyswitch launch + one blind-rotate launch over all ciphertexts of that level.  LIN nodes are
// never materialised on their own; they are folded into the gather of the PBS that consumes them.
//
// Metadata follows the reference's bookkeeping (shortint/ciphertext/mod.rs:28-55,263-270,
// server_key/mod.rs:855-856, add.rs:520-524):
//   value range [vmin, vmax] -- `degree` made two-sided: a PBS input must lie in [0, msg*carry) (or in
//       (-msg*carry, msg*carry) when the caller declares that it uses the padding bit), otherwise
//       the build fails instead of silently evaluating the negacyclic extension of the table;
//   noise -- variance in units of one nominal ciphertext (fresh PBS output = NoiseLevel::NOMINAL):
//       a linear combination carries sum coeff^2 * noise(source) over its *materialised* sources
//       (the same source reached twice is merged first), checked before every PBS against the
//       parameter set's budget (noise_model.h; the reference's MaxNoiseLevel::validate,
//       shortint/ciphertext/mod.rs:28-55).
//
// Multi-GPU: every PBS node has an owner rank; a rank runs its own jobs of a level and
// only the nodes another rank (or the output gather) consumes are exchanged -- one all-gather of
// `e_max` ciphertexts per rank for a level that exports anything, no collective otherwise.  Builders
// steer ownership with owner hints (e.g. one slice of the characters per rank, reduced locally to a
// single block before anything is communicated).
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "engine.h"

namespace fhe {

struct Term {
    uint32_t node;
    int32_t coeff;
};

struct Node {
    enum Kind : uint8_t { INPUT, LIN, PBS } kind;
    bool exported = false;     // PBS: consumed by another rank or by the outputs -> part of the all-gather
    bool half = false;         // PBS (pbs_full_box): the ciphertext holds value - 1/2; every consumer adds coeff * delta / 2
    int16_t owner = -1;        // PBS: rank that runs it (-1 while building = decided at finalize)
    uint32_t level = 0;        // 0 for inputs; PBS: 1 + max level of deps; LIN: max level of deps
    int64_t vmin = 0, vmax = 0;   // clear value range (message+carry space; vmin < 0 = reaches the padding bit)
    double noise = 0.0;        // variance, in units of one nominal ciphertext
    // LIN (already flattened onto materialised nodes: INPUT / PBS)
    std::vector<Term> terms;
    int64_t cst = 0;           // clear constant (in message units, scaled by delta at execution)
    // PBS
    uint32_t src = 0;          // node id of the LIN/any node fed to the table
    uint32_t lut = 0;          // plan-local LUT id
    // materialised nodes
    uint32_t slot = 0;         // pool slot consumers read
    uint32_t job = 0;          // index inside its level
    uint64_t degree() const { return vmax > 0 ? (uint64_t)vmax : 0; }
};

class Circuit {
public:
    // eng may be null: the plan can then be built, finalised and exported, but not run
    Circuit(const fhe_params_t& params, Engine* eng);

    uint32_t input(uint64_t degree);                             // next input ciphertext
    // degree_override >= 0: the caller asserts the value lies in [0, degree_override] (it knows more
    // than interval arithmetic, e.g. "at most one term is non-zero")
    uint32_t lin(const std::vector<Term>& terms, int64_t cst = 0, int64_t degree_override = -1);
    uint32_t add(uint32_t a, uint32_t b) { return lin({{a, 1}, {b, 1}}); }
    uint32_t trivial(int64_t value) { return lin({}, value); }   // create_trivial (mod.rs:684-721)
    // apply_lookup_table.  signed_input: the value may be negative, i.e. the torus value uses the
    // padding bit and the table is read through its negacyclic extension f(x - T) = -f(x).
    uint32_t pbs(uint32_t node, uint32_t lut, bool signed_input = false);
    // Reduction of exactly msg*carry = T bits in ONE lookup (the reference's are_all_comparisons_block_true /
    // is_at_least_one_comparisons_block_true take T - 1 per lookup, scalar_comparison.rs:147-233): the sum s lies in
    // [0, T], and s = T is the padding bit, read as -f(0) by the negacyclic table.  With f = g - 1/2 that is consistent:
    //   any (s != 0):  f(0) = -1/2, f(1..T-1) = +1/2  =>  f(T) = +1/2;      all (s == T):  f = -1/2 on [0, T)  =>  f(T) = +1/2
    // The table's entries are -/+ delta/2 (not multiples of delta), the missing +1/2 is added by whoever reads the result
    // (Node::half: folded into the constant of every linear combination at finalize) -- no extra operation, the same
    // output noise, the same decision margin of delta/2 at the input.  `node` must be a sum with value range [0, T].
    uint32_t pbs_full_box(uint32_t node, bool all);
    // The general form: any 0/1 table g on [0, T) read on an input in [0, T], where the value at T is forced to 1 - g(0)
    // by the table's negacyclic extension (so g(0) = 0 for a threshold [x >= theta] whose accept set reaches T).
    uint32_t pbs_box(uint32_t node, const std::vector<uint8_t>& g);
    // generate_lookup_table with a cache keyed on the table contents (mod.rs:383-399)
    uint32_t lut(const std::vector<uint64_t>& table);
    template <class F>
    uint32_t lut_fn(F f) {
        std::vector<uint64_t> t(total_modulus());
        for (uint32_t i = 0; i < t.size(); i++) t[i] = (uint64_t)f((uint64_t)i);
        return lut(t);
    }
    void output(uint32_t node) { outputs_.push_back(node); }

    // ---- multi-GPU building hints ----
    // world the plan is being built for (builders that shard by hand read it); finalize() must be
    // given the same value.  PBS nodes created while a hint is set belong to that rank.
    void set_build_world(uint32_t world) { build_world_ = world ? world : 1; }
    uint32_t build_world() const { return build_world_; }
    void set_owner_hint(int rank) { owner_hint_ = rank; }
    int owner_hint() const { return owner_hint_; }
    // owner known at build time (-1: none yet): hinted PBS nodes, and LIN / PBS nodes all of whose
    // materialised PBS sources share one owner
    int owner_of(uint32_t node) const;

    // ---- noise budget ----
    // largest noise (nominal-variance units) a PBS input may carry; <= 0 disables the check
    void set_noise_budget(double b) { noise_budget_ = b; }
    double noise_budget() const { return noise_budget_; }
    double max_pbs_input_noise() const { return max_pbs_input_noise_; }

    uint32_t total_modulus() const { return p_.msg_mod * p_.carry_mod; }
    uint32_t msg_modulus() const { return p_.msg_mod; }
    const fhe_params_t& params() const { return p_; }
    uint32_t n_luts() const { return (uint32_t)lut_accs_.size(); }
    const std::vector<uint64_t>& lut_accumulator(uint32_t id) const { return lut_accs_[id]; }
    const Node& node(uint32_t id) const { return nodes_[id]; }
    bool failed() const { return !error_.empty(); }
    const std::string& error() const { return error_; }
    std::string take_error() { std::string e; e.swap(error_); return e; }   // report once, then keep building

    // ---- finalise + query ----
    int finalize(uint32_t world);                 // owners, exports, pool layout, gather descriptions
    uint32_t n_inputs() const { return n_inputs_; }
    uint32_t n_outputs() const { return (uint32_t)outputs_.size(); }
    uint32_t n_levels() const { return (uint32_t)levels_.size(); }
    uint32_t n_pbs() const { return n_pbs_; }
    uint32_t pool_slots() const { return pool_slots_; }
    uint32_t world() const { return world_; }
    // ciphertexts every rank receives over all levels' all-gathers (world * e_max summed)
    uint64_t gathered_lwes() const;
    struct Level {
        std::vector<uint32_t> jobs;       // PBS node ids, ordered by (owner, exported first, creation order)
        std::vector<uint32_t> rank_off;   // [world + 1]: rank r runs jobs [rank_off[r], rank_off[r+1])
        std::vector<uint32_t> n_export;   // [world]: the first n_export[r] jobs of rank r are exported
        uint32_t e_max = 0;               // all-gather count per rank; 0 = this level needs no collective
        uint32_t local_base = 0;          // job i of a rank's range writes pool slot local_base + i ...
        uint32_t local_size = 0;          // ... (same slots on every rank, each holding its own data)
        uint32_t recv_base = 0;           // all-gather destination: [world][e_max] slots
        // CSR gather description of every job, in `jobs` order (host copy; uploaded at finalize)
        std::vector<uint32_t> off, src, lut;
        std::vector<int32_t> coeff;
        std::vector<uint64_t> cst;      // already multiplied by delta
        size_t meta_off = 0, meta_src = 0, meta_coeff = 0, meta_cst = 0, meta_lut = 0;  // byte offsets in d_meta
    };
    const Level& level(uint32_t l) const { return levels_[l]; }
    // output gather (LIN over pool) in CSR form
    const Level& out_level() const { return out_; }

    // ---- execution on the engine's GPU; d_pool holds pool_slots() big LWEs ----
    int upload_meta();
    int run_level_rank(uint64_t* d_pool, uint32_t level, uint32_t rank);   // that rank's jobs -> its local region
    int gather_outputs(const uint64_t* d_pool, uint64_t* d_out);
    int run_host(const uint64_t* inputs, uint64_t* outputs);
    int run_host_parts(const uint64_t* const* parts, const uint32_t* counts, uint32_t n_parts, uint64_t* outputs);     // single GPU, host buffers
    // `instances` independent copies of the plan (finalised for world 1) in one pass: level l of all instances is ONE
    // gather + keyswitch + blind-rotation batch of instances x jobs(l) ciphertexts -- the reference's throughput shape
    // (many independent inputs per call, benches/core_crypto/pbs_bench.rs:430-549; rayon over the blocks of an integer,
    // integer/server_key/radix_parallel/comparison.rs:22-28) for whole operations.  inputs [instance][n_inputs],
    // outputs [instance][n_outputs] ciphertexts; device pointers, ordered on the engine's stream.
    int run_batch_dev(const uint64_t* d_inputs, uint64_t* d_outputs, uint32_t instances);
    // host arrays: rows [instances][row_count] + `shared` (the remaining inputs, read by every instance; may be null when
    // row_count == n_inputs)
    int run_batch_host(const uint64_t* rows, uint32_t row_count, const uint64_t* shared, uint64_t* outputs, uint32_t instances);
    ~Circuit();
    Engine* engine() const { return eng_; }   // nullptr: offline plan

private:
    void set_error(std::string e) { if (error_.empty()) error_ = std::move(e); }   // the first error is the cause
    void flatten(uint32_t node, int64_t mult, std::map<uint32_t, int64_t>& acc, int64_t& cst) const;
    void build_csr(Level& lv, const std::vector<uint32_t>& lin_nodes);
    int batch_prepare(uint32_t instances);
    int batch_load(const uint64_t* d_src, uint32_t first, uint32_t count, uint32_t instances, uint32_t in_slot, uint32_t in_inst);
    int batch_execute(uint64_t* d_outputs, uint32_t instances);

    fhe_params_t p_;
    Engine* eng_;
    std::vector<std::vector<uint64_t>> lut_accs_;     // accumulator of every plan-local LUT id
    std::vector<std::vector<uint64_t>> lut_tables_;   // its clear table
    std::vector<Node> nodes_;
    std::vector<uint32_t> outputs_;
    std::map<std::vector<uint64_t>, uint32_t> lut_cache_;
    std::vector<Level> levels_;
    Level out_;
    uint32_t n_inputs_ = 0, n_pbs_ = 0, pool_slots_ = 0, world_ = 1, build_world_ = 1;
    int owner_hint_ = -1;
    std::map<std::vector<uint8_t>, uint32_t> box_lut_cache_;   // plan-local ids of the -/+ delta/2 tables, by their 0/1 table
    double noise_budget_ = 0.0, max_pbs_input_noise_ = 0.0;
    std::string error_;
    void* d_meta_ = nullptr;
    uint64_t* d_stage_ = nullptr;   // lincomb output / keyswitch input of one level slice
    size_t stage_cap_ = 0;
    uint64_t* d_own_pool_ = nullptr;
    uint64_t* d_own_out_ = nullptr;
    // batch execution (grown on demand): slot-major pool [slot][instance], one level's gathered rows, their table ids,
    // staging of host inputs / outputs
    uint64_t *d_bpool_ = nullptr, *d_bstage_ = nullptr, *d_bio_ = nullptr;
    uint32_t* d_blut_ = nullptr;
    size_t bpool_cap_ = 0, bstage_cap_ = 0, bio_cap_ = 0, blut_cap_ = 0;
};

}  // namespace fhe
