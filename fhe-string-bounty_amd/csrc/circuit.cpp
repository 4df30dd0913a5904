// circuit.cpp -- builder, leveliser and GPU executor of shortint circuits (see circuit.h).
#include "circuit.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "noise_model.h"

namespace fhe {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

Circuit::Circuit(const fhe_params_t& params, Engine* eng) : p_(params), eng_(eng) {
    noise_budget_ = default_noise_budget(params);
}

uint32_t Circuit::input(uint64_t degree) {
    Node n;
    n.kind = Node::INPUT;
    n.level = 0;
    n.vmin = 0;
    n.vmax = (int64_t)degree;
    n.noise = 1.0;             // a fresh ciphertext counts as nominal (shortint/ciphertext/mod.rs:13-26)
    n.slot = n_inputs_++;
    nodes_.push_back(n);
    return (uint32_t)nodes_.size() - 1;
}

void Circuit::flatten(uint32_t id, int64_t mult, std::map<uint32_t, int64_t>& acc, int64_t& cst) const {
    const Node& n = nodes_[id];
    if (n.kind == Node::LIN) {
        for (const Term& t : n.terms) acc[t.node] += mult * t.coeff;   // already flat
        cst += mult * n.cst;
    } else {
        acc[id] += mult;
    }
}

uint32_t Circuit::lin(const std::vector<Term>& terms, int64_t cst, int64_t degree_override) {
    std::map<uint32_t, int64_t> acc;
    int64_t c = cst;
    // value range from the operands as given (an operand LIN may carry a tighter, caller-asserted
    // range than its flattened leaves would suggest)
    int64_t lo = cst, hi = cst;
    for (const Term& t : terms) {
        if (t.node >= nodes_.size()) { set_error("lin: bad node id"); return 0; }
        const Node& s = nodes_[t.node];
        if (t.coeff >= 0) { lo += (int64_t)t.coeff * s.vmin; hi += (int64_t)t.coeff * s.vmax; }
        else              { lo += (int64_t)t.coeff * s.vmax; hi += (int64_t)t.coeff * s.vmin; }
        flatten(t.node, t.coeff, acc, c);
    }
    Node n;
    n.kind = Node::LIN;
    n.cst = c;
    uint32_t lvl = 0;
    double noise = 0.0;
    for (auto& kv : acc) {
        if (kv.second == 0) continue;
        if (kv.second > INT32_MAX || kv.second < INT32_MIN) { set_error("lin: coefficient overflow"); return 0; }
        n.terms.push_back({kv.first, (int32_t)kv.second});
        lvl = std::max(lvl, nodes_[kv.first].level);
        noise += (double)kv.second * (double)kv.second * nodes_[kv.first].noise;   // add.rs:523, on variances
    }
    n.level = lvl;
    n.noise = noise;
    if (degree_override >= 0) { n.vmin = 0; n.vmax = degree_override; }
    else { n.vmin = lo; n.vmax = hi; }
    nodes_.push_back(n);
    return (uint32_t)nodes_.size() - 1;
}

int Circuit::owner_of(uint32_t id) const {
    const Node& n = nodes_[id];
    if (n.kind == Node::PBS) return n.owner;
    if (n.kind == Node::INPUT) return -1;
    int owner = -1;
    for (const Term& t : n.terms) {
        const Node& s = nodes_[t.node];
        if (s.kind != Node::PBS) continue;            // inputs are replicated on every rank
        if (s.owner < 0) return -1;
        if (owner >= 0 && owner != s.owner) return -1;
        owner = s.owner;
    }
    return owner;
}

uint32_t Circuit::lut(const std::vector<uint64_t>& table) {
    auto it = lut_cache_.find(table);
    if (it != lut_cache_.end()) return it->second;
    if (table.size() != total_modulus()) { set_error("lut: table size must be msg_mod*carry_mod"); return 0; }
    std::vector<uint64_t> acc;
    fill_accumulator(p_, table.data(), acc);
    const uint32_t id = (uint32_t)lut_accs_.size();
    lut_accs_.push_back(acc);
    lut_tables_.push_back(table);
    lut_cache_[table] = id;
    return id;
}

uint32_t Circuit::pbs(uint32_t id, uint32_t lut_id, bool signed_input) {
    if (id >= nodes_.size()) { set_error("pbs: bad node id"); return 0; }
    if (lut_id >= lut_tables_.size()) { set_error("pbs: LUT was not created through this plan"); return 0; }
    const std::vector<uint64_t>* table = &lut_tables_[lut_id];
    uint32_t src = id;
    if (nodes_[id].kind != Node::LIN) src = lin({{id, 1}});
    const Node& s = nodes_[src];
    const int64_t T = (int64_t)total_modulus();
    if (s.vmax >= T) {
        set_error("pbs: input degree " + std::to_string(s.vmax) + " overflows the message+carry space");
        return 0;
    }
    if (s.vmin < 0 && !signed_input) {
        set_error("pbs: input may be negative (down to " + std::to_string(s.vmin) + "): it would wrap into the padding bit "
                 "and be read through the table's negacyclic extension; add a constant, or declare a signed input");
        return 0;
    }
    if (s.vmin <= -T) {
        set_error("pbs: signed input " + std::to_string(s.vmin) + " leaves the padding bit's range");
        return 0;
    }
    if (s.terms.empty()) {
        // trivial ciphertext: clear table lookup (shortint/server_key/mod.rs:763-781)
        const int64_t v = s.cst;
        if (v < 0 || v >= T) { set_error("pbs: trivial value out of range"); return 0; }
        return trivial((int64_t)(*table)[(size_t)v]);
    }
    // MaxNoiseLevel::validate (shortint/ciphertext/mod.rs:28-55), on variances
    max_pbs_input_noise_ = std::max(max_pbs_input_noise_, s.noise);
    if (noise_budget_ > 0.0 && s.noise > noise_budget_) {
        char buf[200];
        snprintf(buf, sizeof buf, "pbs: input noise %.1f nominal variances exceeds this parameter set's budget of %.1f",
                 s.noise, noise_budget_);
        std::string detail = buf;
        if (getenv("FHESTR_DEBUG_NOISE")) {
            detail += " [terms:";
            for (const Term& t : s.terms) detail += " " + std::to_string(t.coeff) + "*n" + std::to_string(nodes_[t.node].noise);
            detail += "]";
        }
        set_error(detail);
        return 0;
    }
    Node n;
    n.kind = Node::PBS;
    n.src = src;
    n.lut = lut_id;
    n.level = s.level + 1;
    n.vmin = 0;
    n.vmax = (int64_t)*std::max_element(table->begin(), table->end());   // mod.rs:855
    n.noise = 1.0;                                                        // NoiseLevel::NOMINAL, mod.rs:856
    n.owner = (int16_t)(owner_hint_ >= 0 ? owner_hint_ : owner_of(src));
    nodes_.push_back(n);
    n_pbs_++;
    return (uint32_t)nodes_.size() - 1;
}

uint32_t Circuit::pbs_full_box(uint32_t id, bool all) {
    std::vector<uint8_t> g(total_modulus());
    for (size_t i = 0; i < g.size(); i++) g[i] = (!all && i != 0) ? 1 : 0;     // any: [s != 0]; all: 0 below T (and 1 at T)
    return pbs_box(id, g);
}

uint32_t Circuit::pbs_box(uint32_t id, const std::vector<uint8_t>& g) {
    if (id >= nodes_.size()) { set_error("pbs_box: bad node id"); return 0; }
    const int64_t T = (int64_t)total_modulus();
    if ((int64_t)g.size() != T) { set_error("pbs_box: the table has msg*carry entries"); return 0; }
    for (uint8_t v : g) if (v > 1) { set_error("pbs_box: a 0/1 table"); return 0; }
    uint32_t src = id;
    if (nodes_[id].kind != Node::LIN) src = lin({{id, 1}});
    const Node& s = nodes_[src];
    if (s.vmin < 0 || s.vmax > T) {
        set_error("pbs_box: the input must lie in [0, msg*carry], got [" + std::to_string(s.vmin) + ", " + std::to_string(s.vmax) + "]");
        return 0;
    }
    if (s.terms.empty()) return trivial(s.cst == T ? (int64_t)(1 - g[0]) : (int64_t)g[(size_t)s.cst]);
    max_pbs_input_noise_ = std::max(max_pbs_input_noise_, s.noise);
    if (noise_budget_ > 0.0 && s.noise > noise_budget_) {
        char buf[200];
        snprintf(buf, sizeof buf, "pbs: input noise %.1f nominal variances exceeds this parameter set's budget of %.1f",
                 s.noise, noise_budget_);
        set_error(buf);
        return 0;
    }
    auto it = box_lut_cache_.find(g);
    uint32_t lut_id;
    if (it != box_lut_cache_.end()) {
        lut_id = it->second;
    } else {
        const uint64_t half_delta = ((1ull << 63) / (uint64_t)T) / 2;
        std::vector<uint64_t> values((size_t)T), clear((size_t)T);
        for (int64_t i = 0; i < T; i++) {
            values[(size_t)i] = g[(size_t)i] ? half_delta : 0 - half_delta;
            clear[(size_t)i] = g[(size_t)i];
        }
        std::vector<uint64_t> acc;
        fill_accumulator_torus(p_, values.data(), acc);
        lut_id = (uint32_t)lut_accs_.size();
        lut_accs_.push_back(acc);
        lut_tables_.push_back(clear);                   // not in lut_cache_: an ordinary table with these values is another accumulator
        box_lut_cache_[g] = lut_id;
    }
    Node n;
    n.kind = Node::PBS;
    n.half = true;
    n.src = src;
    n.lut = lut_id;
    n.level = s.level + 1;
    n.vmin = 0;
    n.vmax = 1;
    n.noise = 1.0;
    n.owner = (int16_t)(owner_hint_ >= 0 ? owner_hint_ : owner_of(src));
    nodes_.push_back(n);
    n_pbs_++;
    return (uint32_t)nodes_.size() - 1;
}

void Circuit::build_csr(Level& lv, const std::vector<uint32_t>& lin_nodes) {
    const uint64_t delta = (1ull << 63) / total_modulus();
    lv.off.assign(1, 0);
    lv.src.clear(); lv.coeff.clear(); lv.cst.clear();
    for (uint32_t id : lin_nodes) {
        const Node& s = nodes_[id];
        // sources produced by pbs_full_box hold value - 1/2: their share of the constant is coeff * delta / 2
        if (s.kind == Node::LIN) {
            int64_t halves = 0;
            for (const Term& t : s.terms) {
                lv.src.push_back(nodes_[t.node].slot);
                lv.coeff.push_back(t.coeff);
                if (nodes_[t.node].half) halves += t.coeff;
            }
            lv.cst.push_back((uint64_t)s.cst * delta + (uint64_t)halves * (delta / 2));
        } else {
            lv.src.push_back(s.slot);
            lv.coeff.push_back(1);
            lv.cst.push_back(s.half ? delta / 2 : 0);
        }
        lv.off.push_back((uint32_t)lv.src.size());
    }
}

uint64_t Circuit::gathered_lwes() const {
    uint64_t g = 0;
    for (const auto& lv : levels_) g += (uint64_t)lv.e_max * world_;
    return g;
}

int Circuit::finalize(uint32_t world) {
    if (failed()) return fail("circuit build error: " + error_);
    if (world == 0) return fail("world must be >= 1");
    if (build_world_ != 1 && build_world_ != world)
        return fail("plan was built with owner hints for world " + std::to_string(build_world_));
    world_ = world;
    uint32_t max_level = 0;
    for (const Node& n : nodes_)
        if (n.kind == Node::PBS) max_level = std::max(max_level, n.level);
    levels_.assign(max_level, Level());
    for (uint32_t id = 0; id < nodes_.size(); id++)
        if (nodes_[id].kind == Node::PBS) levels_[nodes_[id].level - 1].jobs.push_back(id);

    // ---- owners: hinted / inherited ones stand; the rest of a level inherit from their sources now
    //      that those are decided (majority, lowest rank on ties), and nodes fed by inputs only are
    //      dealt out in contiguous slices ----
    for (auto& lv : levels_) {
        std::vector<uint32_t> free_nodes;
        for (uint32_t id : lv.jobs) {
            Node& n = nodes_[id];
            if (n.owner >= (int)world) return fail("owner hint beyond the world size");
            if (world == 1) { n.owner = 0; continue; }
            if (n.owner >= 0) continue;
            std::vector<uint32_t> votes(world, 0);
            uint32_t voters = 0;
            for (const Term& t : nodes_[n.src].terms)
                if (nodes_[t.node].kind == Node::PBS) { votes[nodes_[t.node].owner]++; voters++; }
            if (!voters) { free_nodes.push_back(id); continue; }
            n.owner = (int16_t)(std::max_element(votes.begin(), votes.end()) - votes.begin());
        }
        const uint32_t chunk = ((uint32_t)free_nodes.size() + world - 1) / world;
        for (uint32_t i = 0; i < free_nodes.size(); i++) nodes_[free_nodes[i]].owner = (int16_t)(i / chunk);
    }
    // ---- exports: what another rank, or the output gather, reads ----
    for (Node& n : nodes_) n.exported = false;
    if (world > 1) {
        for (const Node& n : nodes_) {
            if (n.kind != Node::PBS) continue;
            for (const Term& t : nodes_[n.src].terms) {
                Node& s = nodes_[t.node];
                if (s.kind == Node::PBS && s.owner != n.owner) s.exported = true;
            }
        }
        for (uint32_t o : outputs_) {
            if (nodes_[o].kind == Node::PBS) nodes_[o].exported = true;
            if (nodes_[o].kind == Node::LIN)
                for (const Term& t : nodes_[o].terms)
                    if (nodes_[t.node].kind == Node::PBS) nodes_[t.node].exported = true;
        }
    }
    // ---- pool layout ----
    uint32_t base = n_inputs_;
    for (auto& lv : levels_) {
        std::stable_sort(lv.jobs.begin(), lv.jobs.end(), [&](uint32_t a, uint32_t b) {
            const Node &x = nodes_[a], &y = nodes_[b];
            if (x.owner != y.owner) return x.owner < y.owner;
            return x.exported && !y.exported;
        });
        lv.rank_off.assign(world + 1, 0);
        lv.n_export.assign(world, 0);
        for (uint32_t id : lv.jobs) {
            lv.rank_off[nodes_[id].owner + 1]++;
            if (nodes_[id].exported) lv.n_export[nodes_[id].owner]++;
        }
        lv.local_size = 0;
        for (uint32_t r = 0; r < world; r++) {
            lv.local_size = std::max(lv.local_size, lv.rank_off[r + 1]);
            lv.rank_off[r + 1] += lv.rank_off[r];
        }
        lv.e_max = *std::max_element(lv.n_export.begin(), lv.n_export.end());
        lv.local_base = base;
        base += lv.local_size;
        lv.recv_base = base;
        base += lv.e_max * world;
        for (uint32_t j = 0; j < lv.jobs.size(); j++) {
            Node& n = nodes_[lv.jobs[j]];
            const uint32_t i = j - lv.rank_off[n.owner];      // index inside its rank's range
            n.job = j;
            n.slot = n.exported ? lv.recv_base + (uint32_t)n.owner * lv.e_max + i : lv.local_base + i;
        }
    }
    pool_slots_ = base;
    for (auto& lv : levels_) {
        std::vector<uint32_t> srcs;
        lv.lut.clear();
        for (uint32_t id : lv.jobs) {
            srcs.push_back(nodes_[id].src);
            lv.lut.push_back(nodes_[id].lut);
        }
        build_csr(lv, srcs);
    }
    build_csr(out_, outputs_);
    return eng_ ? upload_meta() : 0;
}

static size_t align8(size_t x) { return (x + 7) / 8 * 8; }

int Circuit::upload_meta() {
    if (eng_->use()) return 1;
    // plan-local LUT ids -> engine LUT ids (identical accumulators share one resident copy)
    std::vector<uint32_t> engine_id(lut_accs_.size());
    for (size_t i = 0; i < lut_accs_.size(); i++)
        if (eng_->lut_upload_dedup(lut_accs_[i], &engine_id[i])) return 1;
    size_t total = 0;
    auto place = [&](Level& lv) {
        lv.meta_off = total;   total += align8(lv.off.size() * 4);
        lv.meta_src = total;   total += align8((lv.src.size() + 1) * 4);
        lv.meta_coeff = total; total += align8((lv.coeff.size() + 1) * 4);
        lv.meta_cst = total;   total += align8((lv.cst.size() + 1) * 8);
        lv.meta_lut = total;   total += align8((lv.lut.size() + 1) * 4);
    };
    for (auto& lv : levels_) place(lv);
    place(out_);
    std::vector<unsigned char> host(total, 0);
    auto fill = [&](const Level& lv) {
        std::memcpy(host.data() + lv.meta_off, lv.off.data(), lv.off.size() * 4);
        std::memcpy(host.data() + lv.meta_src, lv.src.data(), lv.src.size() * 4);
        std::memcpy(host.data() + lv.meta_coeff, lv.coeff.data(), lv.coeff.size() * 4);
        std::memcpy(host.data() + lv.meta_cst, lv.cst.data(), lv.cst.size() * 8);
        std::vector<uint32_t> ids(lv.lut.size());
        for (size_t i = 0; i < ids.size(); i++) ids[i] = engine_id[lv.lut[i]];
        std::memcpy(host.data() + lv.meta_lut, ids.data(), ids.size() * 4);
    };
    for (auto& lv : levels_) fill(lv);
    fill(out_);
    if (d_meta_) { HIP_TRY(hipFree(d_meta_)); d_meta_ = nullptr; }
    HIP_TRY(hipMalloc(&d_meta_, total ? total : 8));
    HIP_TRY(hipMemcpyAsync(d_meta_, host.data(), total, hipMemcpyHostToDevice, eng_->stream));
    HIP_TRY(hipStreamSynchronize(eng_->stream));
    size_t max_jobs = 1;
    for (auto& lv : levels_) max_jobs = std::max(max_jobs, (size_t)lv.local_size);
    const size_t big = (size_t)p_.k * p_.N + 1;
    if (stage_cap_ < max_jobs * big * 8) {
        if (d_stage_) HIP_TRY(hipFree(d_stage_));
        d_stage_ = nullptr;
        HIP_TRY(hipMalloc((void**)&d_stage_, max_jobs * big * 8));
        stage_cap_ = max_jobs * big * 8;
    }
    return 0;
}

int Circuit::run_level_rank(uint64_t* d_pool, uint32_t l, uint32_t rank) {
    if (!eng_) return fail("offline plan: no engine bound (there is no CPU execution path)");
    if (l >= levels_.size()) return fail("bad level");
    if (rank >= world_) return fail("bad rank");
    const Level& lv = levels_[l];
    const uint32_t lo = lv.rank_off[rank], hi = lv.rank_off[rank + 1];
    if (lo >= hi) return 0;
    if (eng_->use()) return 1;
    const size_t big = (size_t)p_.k * p_.N + 1;
    if ((size_t)(hi - lo) * big * 8 > stage_cap_) return fail("internal: stage buffer smaller than a level slice");
    const unsigned char* m = reinterpret_cast<const unsigned char*>(d_meta_);
    const uint32_t* d_off = reinterpret_cast<const uint32_t*>(m + lv.meta_off) + lo;
    const uint32_t* d_src = reinterpret_cast<const uint32_t*>(m + lv.meta_src);
    const int32_t* d_coeff = reinterpret_cast<const int32_t*>(m + lv.meta_coeff);
    const uint64_t* d_cst = reinterpret_cast<const uint64_t*>(m + lv.meta_cst) + lo;
    const uint32_t* d_lut = reinterpret_cast<const uint32_t*>(m + lv.meta_lut) + lo;
    if (eng_->lincomb_dev(d_pool, d_off, d_src, d_coeff, d_cst, d_stage_, hi - lo)) return 1;
    return eng_->ks_pbs_dev(d_stage_, d_lut, d_pool + (size_t)lv.local_base * big, hi - lo);
}

int Circuit::gather_outputs(const uint64_t* d_pool, uint64_t* d_out) {
    if (!eng_) return fail("offline plan: no engine bound (there is no CPU execution path)");
    if (eng_->use()) return 1;
    const unsigned char* m = reinterpret_cast<const unsigned char*>(d_meta_);
    return eng_->lincomb_dev(d_pool, reinterpret_cast<const uint32_t*>(m + out_.meta_off),
                             reinterpret_cast<const uint32_t*>(m + out_.meta_src),
                             reinterpret_cast<const int32_t*>(m + out_.meta_coeff),
                             reinterpret_cast<const uint64_t*>(m + out_.meta_cst), d_out, n_outputs());
}

int Circuit::run_host(const uint64_t* inputs, uint64_t* outputs) {
    const uint64_t* parts[1] = {inputs};
    const uint32_t counts[1] = {n_inputs_};
    return run_host_parts(parts, counts, 1, outputs);
}

// Inputs given as several host arrays (e.g. the two operands of a string comparison) that together
// hold n_inputs LWEs: each goes to the device pool directly, no staging copy on the host.
int Circuit::run_host_parts(const uint64_t* const* parts, const uint32_t* counts, uint32_t n_parts, uint64_t* outputs) {
    if (!eng_) return fail("offline plan: no engine bound (there is no CPU execution path)");
    if (eng_->use()) return 1;
    if (world_ != 1) return fail("run_host needs a plan finalised for world = 1");
    const size_t big = (size_t)p_.k * p_.N + 1;
    if (!d_own_pool_) HIP_TRY(hipMalloc((void**)&d_own_pool_, (size_t)std::max<uint32_t>(pool_slots_, 1) * big * 8));
    if (!d_own_out_) HIP_TRY(hipMalloc((void**)&d_own_out_, (size_t)std::max<uint32_t>(n_outputs(), 1) * big * 8));
    uint32_t placed = 0;
    for (uint32_t i = 0; i < n_parts; i++) {
        if (!counts[i]) continue;
        if (placed + counts[i] > n_inputs_) return fail("run_host: more input LWEs than the plan has inputs");
        HIP_TRY(hipMemcpyAsync(d_own_pool_ + (size_t)placed * big, parts[i], (size_t)counts[i] * big * 8,
                               hipMemcpyHostToDevice, eng_->stream));
        placed += counts[i];
    }
    if (placed != n_inputs_) return fail("run_host: fewer input LWEs than the plan has inputs");
    for (uint32_t l = 0; l < levels_.size(); l++)
        if (run_level_rank(d_own_pool_, l, 0)) return 1;
    if (gather_outputs(d_own_pool_, d_own_out_)) return 1;
    HIP_TRY(hipMemcpyAsync(outputs, d_own_out_, (size_t)n_outputs() * big * 8, hipMemcpyDeviceToHost, eng_->stream));
    HIP_TRY(hipStreamSynchronize(eng_->stream));
    return eng_->cluster_check();
}

static int grow(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(ptr, bytes));
    *cap = bytes;
    return 0;
}

// ---- many instances of a plan in one pass ----
int Circuit::batch_prepare(uint32_t M) {
    if (!eng_) return fail("offline plan: no engine bound (there is no CPU execution path)");
    if (world_ != 1) return fail("run_batch needs a plan finalised for world = 1 (instances shard over ranks, not levels)");
    if (eng_->use()) return 1;
    const size_t big = (size_t)p_.k * p_.N + 1;
    size_t max_jobs = 1;
    for (auto& lv : levels_) max_jobs = std::max(max_jobs, lv.jobs.size());
    if ((uint64_t)pool_slots_ * M > 0x7FFFFFFFull || (uint64_t)max_jobs * M > 0x7FFFFFFFull) return fail("run_batch: too many instances");
    // the streams of an earlier pass may still read these buffers while they are being replaced
    if (bpool_cap_ < (size_t)pool_slots_ * M * big * 8 || bstage_cap_ < max_jobs * M * big * 8 || blut_cap_ < max_jobs * M * 4)
        if (eng_->sync_all_streams()) return 1;
    if (grow((void**)&d_bpool_, &bpool_cap_, (size_t)std::max<uint32_t>(pool_slots_, 1) * M * big * 8)) return 1;
    if (grow((void**)&d_bstage_, &bstage_cap_, max_jobs * M * big * 8)) return 1;
    if (grow((void**)&d_blut_, &blut_cap_, max_jobs * M * 4)) return 1;
    return 0;
}

// input slots [first, first + count) of every instance <- d_src row (s * in_slot + i * in_inst); in_inst = 0 replicates
int Circuit::batch_load(const uint64_t* d_src, uint32_t first, uint32_t count, uint32_t M, uint32_t in_slot, uint32_t in_inst) {
    if (first + count > n_inputs_) return fail("run_batch: more input LWEs than the plan has inputs");
    const size_t big = (size_t)p_.k * p_.N + 1;
    return eng_->restride_dev(d_src, d_bpool_ + (size_t)first * M * big, count, M, in_slot, in_inst, M, 1);
}

int Circuit::batch_execute(uint64_t* d_outputs, uint32_t M) {
    const size_t big = (size_t)p_.k * p_.N + 1;
    const unsigned char* m = reinterpret_cast<const unsigned char*>(d_meta_);
    for (const Level& lv : levels_) {
        const uint32_t J = (uint32_t)lv.jobs.size();
        if (!J) continue;
        if (eng_->lincomb_batch_dev(d_bpool_, reinterpret_cast<const uint32_t*>(m + lv.meta_off), reinterpret_cast<const uint32_t*>(m + lv.meta_src),
                                    reinterpret_cast<const int32_t*>(m + lv.meta_coeff), reinterpret_cast<const uint64_t*>(m + lv.meta_cst),
                                    d_bstage_, J, M, M, 1, M, 1, reinterpret_cast<const uint32_t*>(m + lv.meta_lut), d_blut_))
            return 1;
        // rows (job j, instance i) -> pool rows (local_base + j, i): one contiguous batch (world 1: every job is local)
        if (eng_->ks_pbs_dev(d_bstage_, d_blut_, d_bpool_ + (size_t)lv.local_base * M * big, J * M)) return 1;
    }
    return eng_->lincomb_batch_dev(d_bpool_, reinterpret_cast<const uint32_t*>(m + out_.meta_off), reinterpret_cast<const uint32_t*>(m + out_.meta_src),
                                   reinterpret_cast<const int32_t*>(m + out_.meta_coeff), reinterpret_cast<const uint64_t*>(m + out_.meta_cst),
                                   d_outputs, n_outputs(), M, M, 1, 1, n_outputs(), nullptr, nullptr);
}

int Circuit::run_batch_dev(const uint64_t* d_inputs, uint64_t* d_outputs, uint32_t instances) {
    if (instances == 0) return 0;
    if (batch_prepare(instances)) return 1;
    if (batch_load(d_inputs, 0, n_inputs_, instances, 1, n_inputs_)) return 1;     // [instance][n_inputs] -> [slot][instance]
    return batch_execute(d_outputs, instances);
}

// Host arrays: `rows` = [instances][row_count] ciphertexts (the first row_count inputs of every instance), `shared` =
// [n_inputs - row_count] ciphertexts every instance reads (one encrypted pattern against many strings), or null.
int Circuit::run_batch_host(const uint64_t* rows, uint32_t row_count, const uint64_t* shared, uint64_t* outputs, uint32_t instances) {
    if (instances == 0) return 0;
    if (row_count > n_inputs_ || (row_count < n_inputs_ && !shared)) return fail("run_batch: fewer input LWEs than the plan has inputs");
    if (batch_prepare(instances)) return 1;
    const size_t big = (size_t)p_.k * p_.N + 1;
    const uint32_t n_shared = n_inputs_ - row_count;
    const size_t rows_words = (size_t)row_count * instances * big, shared_words = (size_t)n_shared * big,
                 out_words = (size_t)n_outputs() * instances * big;
    if (bio_cap_ < (rows_words + shared_words + out_words) * 8 && eng_->sync_all_streams()) return 1;
    if (grow((void**)&d_bio_, &bio_cap_, (rows_words + shared_words + out_words) * 8)) return 1;
    uint64_t *d_rows = d_bio_, *d_shared = d_bio_ + rows_words, *d_out = d_shared + shared_words;
    if (rows_words) HIP_TRY(hipMemcpyAsync(d_rows, rows, rows_words * 8, hipMemcpyHostToDevice, eng_->stream));
    if (shared_words) HIP_TRY(hipMemcpyAsync(d_shared, shared, shared_words * 8, hipMemcpyHostToDevice, eng_->stream));
    if (batch_load(d_rows, 0, row_count, instances, 1, row_count)) return 1;
    if (n_shared && batch_load(d_shared, row_count, n_shared, instances, 1, 0)) return 1;
    if (batch_execute(d_out, instances)) return 1;
    HIP_TRY(hipMemcpyAsync(outputs, d_out, out_words * 8, hipMemcpyDeviceToHost, eng_->stream));
    HIP_TRY(hipStreamSynchronize(eng_->stream));
    return eng_->cluster_check();
}

Circuit::~Circuit() {
    if (!eng_) return;
    (void)hipSetDevice(eng_->device);
    if (d_bpool_) (void)hipFree(d_bpool_);
    if (d_bstage_) (void)hipFree(d_bstage_);
    if (d_bio_) (void)hipFree(d_bio_);
    if (d_blut_) (void)hipFree(d_blut_);
    if (d_meta_) (void)hipFree(d_meta_);
    if (d_stage_) (void)hipFree(d_stage_);
    if (d_own_pool_) (void)hipFree(d_own_pool_);
    if (d_own_out_) (void)hipFree(d_own_out_);
}

}  // namespace fhe
