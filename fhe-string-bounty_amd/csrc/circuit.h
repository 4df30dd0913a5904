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
// Metadata follows the reference's bookkeeping (shortint/ciphertext/mod.rs:263-270,
// server_key/mod.rs:855-856): `degree` = largest clear value a node can hold, checked against
// msg_mod*carry_mod-1 before every PBS so a packing overflow is a build-time error.
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
    uint32_t level = 0;        // 0 for inputs; PBS: 1 + max level of deps; LIN: max level of deps
    uint64_t degree = 0;       // largest clear value (message+carry space)
    // LIN (already flattened onto materialised nodes: INPUT / PBS)
    std::vector<Term> terms;
    int64_t cst = 0;           // clear constant (in message units, scaled by delta at execution)
    // PBS
    uint32_t src = 0;          // node id of the LIN/any node fed to the table
    uint32_t lut = 0;          // engine LUT id
    // materialised nodes
    uint32_t slot = 0;         // pool slot
    uint32_t job = 0;          // index inside its level
};

class Circuit {
public:
    // eng may be null: the plan can then be built, finalised and exported, but not run
    Circuit(const fhe_params_t& params, Engine* eng) : p_(params), eng_(eng) {}

    uint32_t input(uint64_t degree);                             // next input ciphertext
    // degree_override >= 0 replaces the conservative degree bound (caller knows better)
    uint32_t lin(const std::vector<Term>& terms, int64_t cst = 0, int64_t degree_override = -1);
    uint32_t add(uint32_t a, uint32_t b) { return lin({{a, 1}, {b, 1}}); }
    uint32_t trivial(int64_t value) { return lin({}, value); }   // create_trivial (mod.rs:684-721)
    uint32_t pbs(uint32_t node, uint32_t lut);
    // generate_lookup_table with a cache keyed on the table contents (mod.rs:383-399)
    uint32_t lut(const std::vector<uint64_t>& table);
    template <class F>
    uint32_t lut_fn(F f) {
        std::vector<uint64_t> t(total_modulus());
        for (uint32_t i = 0; i < t.size(); i++) t[i] = (uint64_t)f((uint64_t)i);
        return lut(t);
    }
    void output(uint32_t node) { outputs_.push_back(node); }

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
    int finalize(uint32_t world);                 // assigns levels' pool regions (padded per rank)
    uint32_t n_inputs() const { return n_inputs_; }
    uint32_t n_outputs() const { return (uint32_t)outputs_.size(); }
    uint32_t n_levels() const { return (uint32_t)levels_.size(); }
    uint32_t n_pbs() const { return n_pbs_; }
    uint32_t pool_slots() const { return pool_slots_; }
    uint32_t world() const { return world_; }
    struct Level {
        std::vector<uint32_t> jobs;     // PBS node ids
        uint32_t base = 0;              // first pool slot of the level's region
        uint32_t per_rank = 0;          // padded jobs per rank
        // CSR gather description of every job (host copy; uploaded at finalize)
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
    int run_level_slice(uint64_t* d_pool, uint32_t level, uint32_t lo, uint32_t hi);
    int gather_outputs(const uint64_t* d_pool, uint64_t* d_out);
    int run_host(const uint64_t* inputs, uint64_t* outputs);
    int run_host_parts(const uint64_t* const* parts, const uint32_t* counts, uint32_t n_parts, uint64_t* outputs);     // single GPU, host buffers
    ~Circuit();

private:
    void flatten(uint32_t node, int64_t mult, std::map<uint32_t, int64_t>& acc, int64_t& cst) const;
    void build_csr(Level& lv, const std::vector<uint32_t>& lin_nodes);

    fhe_params_t p_;
    Engine* eng_;
    std::vector<std::vector<uint64_t>> lut_accs_;     // accumulator of every plan-local LUT id
    std::vector<std::vector<uint64_t>> lut_tables_;   // its clear table
    std::vector<Node> nodes_;
    std::vector<uint32_t> outputs_;
    std::map<std::vector<uint64_t>, uint32_t> lut_cache_;
    std::vector<Level> levels_;
    Level out_;
    uint32_t n_inputs_ = 0, n_pbs_ = 0, pool_slots_ = 0, world_ = 1;
    std::string error_;
    void* d_meta_ = nullptr;
    uint64_t* d_stage_ = nullptr;   // lincomb output / keyswitch input of one level slice
    size_t stage_cap_ = 0;
    uint64_t* d_own_pool_ = nullptr;
    uint64_t* d_own_out_ = nullptr;
};

}  // namespace fhe
