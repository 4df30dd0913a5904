// fhe_string.cpp -- FheString operations as batched shortint circuits.
//
// The reference snapshot has no FheString type; its building blocks are the integer
// layer's block-wise comparison loops, restated here on whole strings:
//   unchecked_eq / unchecked_ne            integer/server_key/radix_parallel/comparison.rs:10-83
//   unchecked_scalar_eq / _ne (+ packing)  .../scalar_comparison.rs:104-138,366-558
//   are_all_comparisons_block_true         .../scalar_comparison.rs:147-191
//   is_at_least_one_comparisons_block_true .../scalar_comparison.rs:200-233
//   compare_blocks_with_zero               .../scalar_comparison.rs:254-296
// and the char-wise patterns of the docs tutorial / regex engine
// (tfhe/docs/tutorials/ascii_fhe_string.md:84-131, tfhe/examples/regex_engine/execution.rs:63-86).
//
// An FheString is `cap` characters, zero padded; one 8-bit character = 8/log2(msg_mod) shortint
// blocks, little endian (integer/block_decomposition.rs:119-144).  Semantics of every operation =
// the corresponding clear-text function on the unpadded ASCII string.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <set>
#include <string>

#include "circuit.h"

namespace fhe {

struct Str {
    uint32_t cap = 0;
    std::vector<std::vector<uint32_t>> ch;   // [char][block] node ids
    std::vector<uint32_t> occ;               // optional, [char]: 0/1 node "this slot holds a character"
};

class StrOps {
public:
    explicit StrOps(Circuit& c) : c(c) {
        M = c.msg_modulus();
        T = c.total_modulus();
        bits_per_block = 0;
        while ((1u << bits_per_block) < M) bits_per_block++;
        bpc = 8 / bits_per_block;
        ok = (1u << bits_per_block) == M && 8 % bits_per_block == 0 && T / M >= M;
        world = c.build_world();
    }
    Circuit& c;
    uint32_t M, T, bits_per_block, bpc, world;
    bool ok;

    // ---- multi-GPU: which rank owns the work on element idx of `total` (contiguous slices) ----
    int owner_for(uint32_t idx, uint32_t total) const { return world > 1 && total ? (int)((uint64_t)idx * world / total) : -1; }
    struct Scope {   // PBS nodes created while alive belong to `rank` (no-op for rank < 0)
        Circuit& c; int prev;
        Scope(Circuit& c, int rank) : c(c), prev(c.owner_hint()) { if (rank >= 0) c.set_owner_hint(rank); }
        ~Scope() { c.set_owner_hint(prev); }
    };
    // ---- noise: split a sum into pieces one lookup may take (value bound max_terms, noise budget) ----
    double budget() const { return c.noise_budget() > 0 ? c.noise_budget() : 1e300; }
    bool p_large() const { return c.params().N >= 16384; }      // a lookup level costs what its lookups cost, not one PBS latency
    // A whole character as ONE lookup input, hi * M + lo: needs two blocks per char whose 8 bits fit the message +
    // carry space (PARAM_MESSAGE_4_CARRY_4: 4-bit blocks, 256 plaintexts) and 1 + M^2 nominal variances within the
    // noise budget -- "one PBS per encrypted character".  Per-character predicates and case conversion then cost 1
    // PBS instead of 3.
    bool whole_char_fits(const std::vector<uint32_t>& b) const {
        return bpc == 2 && (uint64_t)M * M <= (uint64_t)T &&
               c.node(b[0]).noise + (double)M * M * c.node(b[1]).noise <= budget();
    }
    uint32_t whole_char(const std::vector<uint32_t>& b) { return c.lin({{b[1], (int32_t)M}, {b[0], 1}}); }
    std::vector<std::vector<Term>> term_groups(const std::vector<Term>& terms, size_t max_terms) const {
        std::vector<std::vector<Term>> out;
        double nu = 0;
        for (const Term& t : terms) {
            const double add = (double)t.coeff * t.coeff * c.node(t.node).noise;
            if (out.empty() || out.back().size() >= max_terms || (nu + add > budget() && !out.back().empty())) {
                out.emplace_back();
                nu = 0;
            }
            out.back().push_back(t);
            nu += add;
        }
        return out;
    }
    bool is_trivial(uint32_t node, int64_t* value = nullptr) const {
        const Node& n = c.node(node);
        if (n.kind != Node::LIN || !n.terms.empty()) return false;
        if (value) *value = n.cst;
        return true;
    }
    // Encrypted-vs-encrypted comparisons: true = compare two blocks per PBS (see packed_pair_eq),
    // false = the reference's one-bivariate-PBS-per-block shape (comparison.rs:10-33).
    bool packed_compare = true;
    bool full_box_reduce = true;       // reductions take T bits per lookup (false: T - 1, the reference's chunks)

    Str input_string(uint32_t cap) {
        Str s;
        s.cap = cap;
        s.ch.resize(cap);
        for (auto& blocks : s.ch)
            for (uint32_t b = 0; b < bpc; b++) blocks.push_back(c.input(M - 1));
        return s;
    }
    // clear byte -> block digits (integer/block_decomposition.rs:119-144)
    uint32_t clear_block(uint8_t v, uint32_t b) const { return (v >> (b * bits_per_block)) & (M - 1); }

    // ---- reductions of 0/1 blocks ----
    // are_all_comparisons_block_true (scalar_comparison.rs:147-191) / is_at_least_one_comparisons_block_true
    // (scalar_comparison.rs:200-233) on one rank: chunks of up to msg*carry - 1 bits (fewer if the
    // noise budget says so) are summed and sent through `x == chunk_len` / `x != 0`, repeated to one bit
    // Default (not the reference-shaped plans): chunks of T bits where that saves a lookup level -- the sum T is the
    // padding bit, answered consistently by a table of -/+ delta/2 (Circuit::pbs_full_box).  AND over a 16-char pattern
    // and OR over up to 256 offsets then take one level less each (PARAM_MESSAGE_2_CARRY_2: contains 16-in-256 7 -> 5 levels).
    uint32_t reduce_local(std::vector<uint32_t> bits, bool all) {
        const uint32_t nz = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
        while (bits.size() > 1) {
            std::vector<Term> terms;
            for (uint32_t b : bits) terms.push_back({b, 1});
            std::vector<uint32_t> next;
            for (const auto& g : term_groups(terms, full_box_reduce ? T : T - 1)) {
                const size_t len = g.size();
                if (len == T) { next.push_back(c.pbs_full_box(c.lin(g), all)); continue; }
                const uint32_t l = all ? c.lut_fn([len](uint64_t x) { return (uint64_t)(x == len); }) : nz;
                next.push_back(c.pbs(c.lin(g), l));
            }
            bits.swap(next);
        }
        return bits[0];
    }
    // The same over several ranks: every rank first reduces the bits it owns to ONE block,
    // only those blocks travel (one all-gather), and the last PBS combines them.
    uint32_t reduce_bits(const std::vector<uint32_t>& bits, bool all) {
        if (bits.empty()) return c.trivial(all ? 1 : 0);
        if (world <= 1 || c.owner_hint() >= 0) return reduce_local(bits, all);
        std::map<int, std::vector<uint32_t>> by_owner;
        for (uint32_t b : bits) by_owner[c.owner_of(b)].push_back(b);
        if (by_owner.size() == 1 && by_owner.begin()->first < 0) return reduce_local(bits, all);
        std::vector<uint32_t> partial;
        for (auto& kv : by_owner) {
            if (kv.first < 0) { partial.insert(partial.end(), kv.second.begin(), kv.second.end()); continue; }
            Scope sc(c, kv.first);
            partial.push_back(reduce_local(kv.second, all));
        }
        return reduce_local(partial, all);
    }
    uint32_t all_true(const std::vector<uint32_t>& bits) { return reduce_bits(bits, true); }
    uint32_t any_true(const std::vector<uint32_t>& bits) { return reduce_bits(bits, false); }

    // message_extract (shortint/server_key/mod.rs: x -> x % msg_mod) of a block that carries more than
    // nominal noise (a sum of ciphertexts): back to NoiseLevel::NOMINAL for one PBS
    std::map<uint32_t, uint32_t> fresh_memo;
    uint32_t fresh(uint32_t block) {
        if (c.node(block).noise <= 1.0) return block;
        auto it = fresh_memo.find(block);
        if (it != fresh_memo.end()) return it->second;
        const uint32_t mm = M;
        return fresh_memo[block] = c.pbs(block, c.lut_fn([mm](uint64_t x) { return x % mm; }));
    }
    // ---- block-level comparisons ----
    // bivariate LUT on lhs*M + rhs (bivariate_pbs.rs:71-96,167-182)
    uint32_t block_eq(uint32_t a, uint32_t b, bool want_equal) {
        const uint32_t m = M;
        if ((double)M * M * c.node(a).noise + c.node(b).noise > budget()) a = fresh(a);     // noisy operands
        if ((double)M * M * c.node(a).noise + c.node(b).noise > budget()) b = fresh(b);
        const uint32_t l = c.lut_fn([m, want_equal](uint64_t x) {
            const uint64_t lhs = (x / m) % m, rhs = (x % m) % m;
            return (uint64_t)((lhs == rhs) == want_equal);
        });
        return c.pbs(c.lin({{a, (int32_t)M}, {b, 1}}), l);
    }
    // Two blocks per PBS: (lo_a + M*hi_a) - (lo_b + M*hi_b) lies in (-T, T); as a torus value it uses
    // the padding bit for the sign, and a negacyclic table with f(0) = 1, f(1..T-1) = 0 returns
    // -f(x - T) = 0 on the negative side as well, so one lookup answers "both blocks equal".
    // (Only equality can be read this way: a table that is 1 on both signs cannot be negacyclic.)
    uint32_t packed_pair_eq(uint32_t lo_a, uint32_t hi_a, uint32_t lo_b, uint32_t hi_b) {
        const uint32_t l = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 0); });
        return c.pbs(c.lin({{lo_a, 1}, {hi_a, (int32_t)M}, {lo_b, -1}, {hi_b, -(int32_t)M}}), l, /*signed_input=*/true);
    }
    // equality bits of two encrypted chars: bpc bits (reference shape) or bpc/2 bits (packed)
    bool packed_pair_fits(uint32_t lo_a, uint32_t hi_a, uint32_t lo_b, uint32_t hi_b) const {
        const double m2 = (double)M * M;
        return c.node(lo_a).noise + c.node(lo_b).noise + m2 * (c.node(hi_a).noise + c.node(hi_b).noise) <= budget();
    }
    void char_eq_bits(const std::vector<uint32_t>& x, const std::vector<uint32_t>& y, std::vector<uint32_t>& out) {
        // packed form only where its 2(1 + M^2) nominal variances fit the parameter set's noise budget
        // (noise_model.h); otherwise the reference's bivariate shape, M^2 + 1
        bool packed = packed_compare && bpc % 2 == 0;
        for (uint32_t k = 0; packed && k + 1 < bpc; k += 2) packed = packed_pair_fits(x[k], x[k + 1], y[k], y[k + 1]);
        if (packed) {
            for (uint32_t k = 0; k + 1 < bpc; k += 2) out.push_back(packed_pair_eq(x[k], x[k + 1], y[k], y[k + 1]));
        } else {
            for (uint32_t k = 0; k < bpc; k++) out.push_back(block_eq(x[k], y[k], true));
        }
    }
    // pack_block_chunk + scalar LUT (scalar_comparison.rs:104-138,312-336,431-452): two blocks of a
    // char packed as hi*M + lo, compared with the clear value packed the same way.
    uint32_t packed_scalar_cmp(uint32_t lo, uint32_t hi, uint32_t clear, bool want_equal) {
        const uint32_t l = c.lut_fn([clear, want_equal](uint64_t x) { return (uint64_t)((x == clear) == want_equal); });
        return c.pbs(c.lin({{hi, (int32_t)M}, {lo, 1}}), l);
    }
    // 0/1 blocks saying char == clear byte (want_equal) / != per packed pair
    void char_scalar_bits(const std::vector<uint32_t>& blocks, uint8_t v, bool want_equal, std::vector<uint32_t>& out) {
        for (uint32_t b = 0; b + 1 < bpc; b += 2) {
            const uint32_t clear = clear_block(v, b + 1) * M + clear_block(v, b);
            out.push_back(packed_scalar_cmp(blocks[b], blocks[b + 1], clear, want_equal));
        }
    }
    // one 0/1 block: char == clear byte
    uint32_t char_scalar_eq(const std::vector<uint32_t>& blocks, uint8_t v) {
        std::vector<uint32_t> bits;
        char_scalar_bits(blocks, v, true, bits);
        return all_true(bits);
    }
    // compare_blocks_with_zero (scalar_comparison.rs:254-296): one 0/1 block, char == 0
    uint32_t char_is_zero(const std::vector<uint32_t>& blocks) {
        const uint32_t per = (T - 1) / (M - 1);
        const uint32_t tt = T;
        const uint32_t z = c.lut_fn([tt](uint64_t x) { return (uint64_t)((x % tt) == 0); });
        std::vector<uint32_t> bits;
        for (size_t i = 0; i < blocks.size(); i += per) {
            std::vector<Term> terms;
            for (size_t j = i; j < std::min(blocks.size(), i + per); j++) terms.push_back({blocks[j], 1});
            bits.push_back(c.pbs(c.lin(terms), z));
        }
        return all_true(bits);
    }
    const std::vector<uint32_t>* ch_or_null(const Str& s, uint32_t i) const { return i < s.cap ? &s.ch[i] : nullptr; }

    // ---- whole-string equality ----
    uint32_t eq(const Str& a, const Str& b, bool want_equal) {
        if (packed_compare && !want_equal) return not_bit(eq(a, b, true));   // ne = 1 - eq (linear)
        const uint32_t n = std::max(a.cap, b.cap);
        std::vector<uint32_t> bits;
        for (uint32_t i = 0; i < n; i++) {
            Scope sc(c, owner_for(i, n));   // a contiguous slice of the characters per rank
            const auto* x = ch_or_null(a, i);
            const auto* y = ch_or_null(b, i);
            if (x && y) {
                if (packed_compare) char_eq_bits(*x, *y, bits);
                else for (uint32_t k = 0; k < bpc; k++) bits.push_back(block_eq((*x)[k], (*y)[k], want_equal));
            } else {
                // the shorter string is implicitly zero padded: compare the other one's char with 0
                std::vector<uint32_t> one;
                char_scalar_bits(x ? *x : *y, 0, want_equal, one);
                bits.insert(bits.end(), one.begin(), one.end());
            }
        }
        return want_equal ? all_true(bits) : any_true(bits);
    }
    uint32_t eq_clear(const Str& a, const uint8_t* clear, uint32_t len, bool want_equal) {
        for (uint32_t i = a.cap; i < len; i++)
            if (clear[i] != 0) return c.trivial(want_equal ? 0 : 1);   // longer than the capacity
        std::vector<uint32_t> bits;
        for (uint32_t i = 0; i < a.cap; i++) {
            Scope sc(c, owner_for(i, a.cap));
            char_scalar_bits(a.ch[i], i < len ? clear[i] : 0, want_equal, bits);
        }
        return want_equal ? all_true(bits) : any_true(bits);
    }

    // ---- pattern matching with an encrypted, zero padded pattern ----
    // z[i] = [pat[i] == 0]
    std::vector<uint32_t> pattern_zero_bits(const Str& pat) {
        std::vector<uint32_t> z;
        for (uint32_t i = 0; i < pat.cap; i++) z.push_back(char_is_zero(pat.ch[i]));
        return z;
    }
    // sum of the block-equality bits of (s[ci], pat[pi]) and how many bits that is (n: all equal)
    std::pair<uint32_t, uint32_t> char_eq_sum(const Str& s, uint32_t ci, const Str& pat, uint32_t pi,
                                              std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>>& memo) {
        auto key = std::make_pair(ci, pi);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        std::vector<uint32_t> bits;
        char_eq_bits(s.ch[ci], pat.ch[pi], bits);
        std::vector<Term> terms;
        for (uint32_t bit : bits) terms.push_back({bit, 1});
        return memo[key] = std::make_pair(c.lin(terms), (uint32_t)bits.size());
    }
    // t = [pat[pi] == 0] OR [s[ci] == pat[pi]]  (prefix-style match: pattern padding matches anything)
    // q = [s[ci] == pat[pi]]                    (exact match incl. padding)
    uint32_t char_match(const Str& s, uint32_t ci, const Str& pat, uint32_t pi, const std::vector<uint32_t>& z,
                        bool padding_wildcard, std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>>& sums,
                        std::map<std::pair<uint32_t, uint32_t>, uint32_t>& memo) {
        if (ci >= s.cap) return z[pi];   // null char: equal iff pat[pi] == 0 (both readings)
        auto key = std::make_pair(ci, pi);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        const auto sn = char_eq_sum(s, ci, pat, pi, sums);
        const uint32_t sum = sn.first, n = sn.second;
        uint32_t r;
        if (padding_wildcard) {
            // value = eq_sum + n*z in [0, 2n]  ->  match iff value >= n (z = 1, or all n bits set)
            const uint32_t l = c.lut_fn([n](uint64_t x) { return (uint64_t)(x >= n); });
            r = c.pbs(c.lin({{sum, 1}, {z[pi], (int32_t)n}}), l);
        } else {
            const uint32_t l = c.lut_fn([n](uint64_t x) { return (uint64_t)(x == n); });
            r = c.pbs(sum, l);
        }
        return memo[key] = r;
    }
    // Several pattern characters per lookup (padding_wildcard only; relies on the padding being at the END of the pattern,
    // z[i] <= z[i+1]).  With a_j in [0, n] the number of equal block (pair)s of character j and z_j its padding bit, the
    // characters of a group match iff  L = sum_j (c_j a_j + d_j z_j) >= theta = n sum_j c_j  when the weights grow from the
    // last character backwards as  d_j = n c_j  (a padding character counts as a full match whatever its a_j) and
    // c_j = 1 + sum_{j' > j} d_j'  (one missing bit of a live character cannot be made up by everything behind it).
    // L_max = 2 n sum c_j must fit [0, T]: PARAM_MESSAGE_2_CARRY_2 (n = 2 packed pairs per character): two characters,
    // c = (3, 1), d = (6, 2), L in [0, 16] = the whole box incl. the padding bit (Circuit::pbs_box), 60 nominal variances.
    // Halves the per-(offset, character) level of contains / find / starts_with.
    struct GroupWeights { std::vector<int32_t> c, d; int64_t theta = 0, lmax = 0; double noise = 0; };
    GroupWeights group_weights(uint32_t k, uint32_t n) const {
        GroupWeights w;
        w.c.assign(k, 0); w.d.assign(k, 0);
        int64_t behind = 0, csum = 0;
        for (int j = (int)k - 1; j >= 0; j--) {
            w.c[j] = (int32_t)(1 + behind);
            w.d[j] = (int32_t)(n * w.c[j]);
            behind += w.d[j];
            csum += w.c[j];
            w.noise += (double)w.c[j] * w.c[j] * n + (double)w.d[j] * w.d[j];
        }
        w.theta = (int64_t)n * csum;
        w.lmax = 2 * (int64_t)n * csum;
        return w;
    }
    uint32_t group_size(uint32_t n, uint32_t remaining) const {
        uint32_t k = 1;
        while (k < remaining && k < 8) {
            const GroupWeights w = group_weights(k + 1, n);
            if (w.lmax > (int64_t)T || w.noise > budget()) break;
            k++;
        }
        return k;
    }
    uint32_t group_match(const Str& s, uint32_t ci, const Str& pat, uint32_t pi, uint32_t k, const std::vector<uint32_t>& z,
                         std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>>& sums,
                         std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t>& memo) {
        if (ci >= s.cap) return z[pi];          // the whole group lies behind the string: all of it must be padding = z[pi]
        auto key = std::make_tuple(ci, pi, k);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        uint32_t n = 0;
        std::vector<uint32_t> a(k, UINT32_MAX);
        for (uint32_t j = 0; j < k; j++) {
            if (ci + j >= s.cap) continue;      // behind the string: no equal blocks, only its padding bit can save it
            const auto sn = char_eq_sum(s, ci + j, pat, pi + j, sums);
            a[j] = sn.first;
            n = sn.second;
        }
        const GroupWeights w = group_weights(k, n);
        std::vector<Term> terms;
        for (uint32_t j = 0; j < k; j++) {
            if (a[j] != UINT32_MAX) terms.push_back({a[j], w.c[j]});
            terms.push_back({z[pi + j], w.d[j]});
        }
        const uint32_t L = c.lin(terms);
        const int64_t theta = w.theta;
        uint32_t r;
        if (c.node(L).vmax >= (int64_t)T) {
            std::vector<uint8_t> g(T);
            for (uint32_t x = 0; x < T; x++) g[x] = (int64_t)x >= theta;
            r = c.pbs_box(L, g);
        } else {
            r = c.pbs(L, c.lut_fn([theta](uint64_t x) { return (uint64_t)((int64_t)x >= theta); }));
        }
        return memo[key] = r;
    }
    // match[o] for o in [0, n_off): AND_i char_match(s[o+i], pat[i])
    // need_end: additionally require the string to end right after the window (s[o+pat.cap] null)
    std::vector<uint32_t> window_matches(const Str& s, const Str& pat, uint32_t n_off, bool padding_wildcard,
                                         bool need_end = false) {
        std::vector<uint32_t> z = pattern_zero_bits(pat);
        std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>> sums;
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> memo;
        std::map<uint32_t, uint32_t> s_zero;
        std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> group_memo;
        // characters per lookup: more than one only with wildcard padding, in the default (not reference-shaped) plans
        uint32_t per_lookup = 1;
        if (padding_wildcard && full_box_reduce && pat.cap > 1 && s.cap > 0) {
            // how many equality bits a character has ((s[0], pat[0]) is needed by offset 0 anyway; memoised in `sums`)
            per_lookup = group_size(char_eq_sum(s, 0, pat, 0, sums).second, pat.cap);
        }
        std::vector<uint32_t> match;
        for (uint32_t o = 0; o < n_off; o++) {
            Scope sc(c, owner_for(o, n_off));   // a contiguous slice of the offsets per rank
            std::vector<uint32_t> bits;
            if (per_lookup > 1) {
                for (uint32_t i = 0; i < pat.cap; i += per_lookup)
                    bits.push_back(group_match(s, o + i, pat, i, std::min(per_lookup, pat.cap - i), z, sums, group_memo));
            } else if (!padding_wildcard && full_box_reduce) {
                // exact match incl. padding: no per-character lookup at all -- the equal-block counts of up to T / n
                // characters are summed and tested against their maximum in one lookup (T itself through pbs_full_box)
                std::vector<Term> run;
                int64_t run_max = 0;
                auto flush = [&]() {
                    if (run.empty()) return;
                    const uint32_t sum = c.lin(run);
                    const int64_t want = run_max;
                    bits.push_back(want == (int64_t)T ? c.pbs_full_box(sum, true)
                                                      : c.pbs(sum, c.lut_fn([want](uint64_t x) { return (uint64_t)((int64_t)x == want); })));
                    run.clear();
                    run_max = 0;
                };
                for (uint32_t i = 0; i < pat.cap; i++) {
                    if (o + i >= s.cap) { bits.push_back(z[i]); continue; }       // behind the string: the pattern must be padding there
                    const auto sn = char_eq_sum(s, o + i, pat, i, sums);
                    if (run_max + (int64_t)sn.second > (int64_t)T || c.node(sn.first).noise * (double)(run.size() + 1) > budget()) flush();
                    run.push_back({sn.first, 1});
                    run_max += sn.second;
                }
                flush();
            } else
            for (uint32_t i = 0; i < pat.cap; i++)
                bits.push_back(char_match(s, o + i, pat, i, z, padding_wildcard, sums, memo));
            if (need_end && o + pat.cap < s.cap) {
                auto it = s_zero.find(o + pat.cap);
                if (it == s_zero.end()) it = s_zero.emplace(o + pat.cap, char_is_zero(s.ch[o + pat.cap])).first;
                bits.push_back(it->second);
            }
            match.push_back(all_true(bits));
        }
        return match;
    }
    // Clear pattern, whole characters as lookup inputs (4-bit blocks, 8-bit plaintext space): classify instead of
    // compare.  Level 1 maps every character to the id of the distinct pattern character it equals (0 = none), one
    // PBS per position whatever the pattern; level L = 2, 4, ... maps the pair (class of the window of length L/2 at t,
    // class of the one at t + L/2), packed as a + B b, to the id of the pattern substring of length L it spells
    // (0 = none).  A window equals at most one distinct substring, so ids are well defined even with repeated
    // characters.  match[o] is the class of the whole pattern at o (two overlapping top windows and one more PBS
    // when len is not a power of two).  Cost: about (1 + log2 len) PBS per position instead of (distinct characters
    // + AND tree): 3,068 instead of 5,105 for a 4-char pattern in 1024 chars, 5 n instead of ~18 n for 16 chars.
    // Returns false when a packing does not fit (space or noise budget): the caller falls back.
    bool window_matches_clear_classes(const Str& s, const uint8_t* pat, uint32_t len, uint32_t n_off, std::vector<uint32_t>& match) {
        if (len < 2 || n_off < 2) return false;
        const uint32_t span = n_off + len - 1;                 // characters any window touches
        for (uint32_t t = 0; t < span; t++) if (!whole_char_fits(s.ch[t])) return false;
        uint32_t top = 1;
        while (top * 2 <= len) top *= 2;
        // offsets (inside the pattern) of the windows needed at every length
        std::map<uint32_t, std::set<uint32_t>> need;
        need[top].insert(0);
        if (top != len) need[top].insert(len - top);
        for (uint32_t L = top; L >= 2; L /= 2)
            for (uint32_t o : need[L]) { need[L / 2].insert(o); need[L / 2].insert(o + L / 2); }
        // ids of the distinct needed substrings per length
        std::map<uint32_t, std::map<std::string, uint32_t>> ids;
        for (auto& [L, offs] : need)
            for (uint32_t o : offs) {
                const std::string sub(reinterpret_cast<const char*>(pat) + o, L);
                if (!ids[L].count(sub)) { const uint32_t id = (uint32_t)ids[L].size() + 1; ids[L][sub] = id; }
            }
        for (uint32_t L = 1; L < top; L *= 2) {
            const uint64_t B = ids[L].size() + 1;
            if (B * B > T || 1.0 + (double)(B * B) > budget()) return false;      // pbs outputs are nominal: 1 + B^2 variances
        }
        if (top != len) { const uint64_t B = ids[top].size() + 1; if (B * B > T || 1.0 + (double)(B * B) > budget()) return false; }
        // level 1
        std::vector<uint32_t> cur(span);
        {
            std::vector<uint64_t> table(256, 0);
            for (auto& [sub, id] : ids[1]) table[(uint8_t)sub[0]] = id;
            const uint32_t l1 = c.lut_fn([&](uint64_t x) { return x < 256 ? table[x] : 0; });
            for (uint32_t t = 0; t < span; t++) {
                Scope sc(c, owner_for(t, span));
                cur[t] = c.pbs(whole_char(s.ch[t]), l1);
            }
        }
        // doubling
        for (uint32_t L = 2; L <= top; L *= 2) {
            const uint32_t B = (uint32_t)ids[L / 2].size() + 1, count = span - L + 1;
            std::map<uint64_t, uint64_t> pair_id;           // a + B b -> id
            for (auto& [sub, id] : ids[L])
                pair_id[ids[L / 2].at(sub.substr(0, L / 2)) + (uint64_t)B * ids[L / 2].at(sub.substr(L / 2))] = id;
            const uint32_t lut = c.lut_fn([&](uint64_t x) { auto it = pair_id.find(x); return it == pair_id.end() ? (uint64_t)0 : it->second; });
            std::vector<uint32_t> next(count);
            for (uint32_t t = 0; t < count; t++) {
                Scope sc(c, owner_for(t, count));
                next[t] = c.pbs(c.lin({{cur[t], 1}, {cur[t + L / 2], (int32_t)B}}), lut);
            }
            cur.swap(next);
        }
        match.clear();
        if (top == len) {            // one class at the top: its id 1 is the match bit
            match.assign(cur.begin(), cur.begin() + n_off);
            return true;
        }
        const uint32_t B = (uint32_t)ids[top].size() + 1;
        const uint64_t want = ids[top].at(std::string(reinterpret_cast<const char*>(pat), top)) +
                              (uint64_t)B * ids[top].at(std::string(reinterpret_cast<const char*>(pat) + len - top, top));
        const uint32_t fin = c.lut_fn([want](uint64_t x) { return (uint64_t)(x == want); });
        for (uint32_t o = 0; o < n_off; o++) {
            Scope sc(c, owner_for(o, n_off));
            match.push_back(c.pbs(c.lin({{cur[o], 1}, {cur[o + len - top], (int32_t)B}}), fin));
        }
        return true;
    }
    // clear pattern: match[o] = AND_{i<len} [s[o+i] == pat[i]], o + len <= cap
    std::vector<uint32_t> window_matches_clear(const Str& s, const uint8_t* pat, uint32_t len, uint32_t n_off) {
        {
            std::vector<uint32_t> classed;
            if (window_matches_clear_classes(s, pat, len, n_off, classed)) return classed;
        }
        std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> memo;   // (char, byte) -> bits
        std::vector<uint32_t> match;
        for (uint32_t o = 0; o < n_off; o++) {
            Scope sc(c, owner_for(o, n_off));
            std::vector<uint32_t> bits;
            for (uint32_t i = 0; i < len; i++) {
                auto key = std::make_pair(o + i, (uint32_t)pat[i]);
                auto it = memo.find(key);
                if (it == memo.end()) {
                    std::vector<uint32_t> b;
                    char_scalar_bits(s.ch[o + i], pat[i], true, b);
                    it = memo.emplace(key, b).first;
                }
                bits.insert(bits.end(), it->second.begin(), it->second.end());
            }
            match.push_back(all_true(bits));
        }
        return match;
    }

    uint32_t starts_with(const Str& s, const Str& pat) { return window_matches(s, pat, 1, true)[0]; }
    uint32_t starts_with_clear(const Str& s, const uint8_t* pat, uint32_t len) {
        if (len > s.cap) return c.trivial(0);
        if (len == 0) return c.trivial(1);
        return window_matches_clear(s, pat, len, 1)[0];
    }
    uint32_t contains(const Str& s, const Str& pat) { return any_true(window_matches(s, pat, s.cap, true)); }
    uint32_t contains_clear(const Str& s, const uint8_t* pat, uint32_t len) {
        if (len > s.cap) return c.trivial(0);
        if (len == 0) return c.trivial(1);
        return any_true(window_matches_clear(s, pat, len, s.cap - len + 1));
    }
    // ends_with: some suffix of s (incl. the empty one at offset cap) equals the whole padded pattern
    uint32_t ends_with(const Str& s, const Str& pat) { return any_true(window_matches(s, pat, s.cap + 1, false, true)); }
    uint32_t ends_with_clear(const Str& s, const uint8_t* pat, uint32_t len) {
        if (len > s.cap) return c.trivial(0);
        if (len == 0) return c.trivial(1);
        // s[o..o+len) == pat and (o+len == cap or s[o+len] == 0)
        std::vector<uint32_t> m = window_matches_clear(s, pat, len, s.cap - len + 1);
        std::vector<uint32_t> cand;
        for (uint32_t o = 0; o + len <= s.cap; o++) {
            if (o + len == s.cap) { cand.push_back(m[o]); continue; }
            const uint32_t zend = char_is_zero(s.ch[o + len]);
            const uint32_t l = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 2); });
            cand.push_back(c.pbs(c.lin({{m[o], 1}, {zend, 1}}), l));
        }
        return any_true(cand);
    }

    // ---- find: (found, index digits) of the first match ----
    // prefix_any[o] = OR_{o' <= o} bits[o'] through a blocked scan (fan-in T-1 per PBS)
    std::vector<uint32_t> prefix_or(const std::vector<uint32_t>& bits) {
        const uint32_t F = full_box_reduce ? T : T - 1;        // a run of T bits: Circuit::pbs_full_box
        const uint32_t nz = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
        const size_t n = bits.size();
        if (n <= 1) return bits;
        std::vector<uint32_t> within(n), block_tot;
        for (size_t b = 0; b < n; b += F) {
            std::vector<Term> run;
            for (size_t j = b; j < std::min(n, b + F); j++) {
                run.push_back({bits[j], 1});
                within[j] = run.size() == 1 ? bits[j] : run.size() == T ? c.pbs_full_box(c.lin(run), false) : c.pbs(c.lin(run), nz);
            }
            block_tot.push_back(within[std::min(n, b + F) - 1]);
        }
        if (block_tot.size() == 1) return within;
        std::vector<uint32_t> block_pre = prefix_or(block_tot);   // inclusive prefix over blocks
        std::vector<uint32_t> out(n);
        for (size_t j = 0; j < n; j++) {
            const size_t b = j / F;
            out[j] = b == 0 ? within[j] : c.pbs(c.lin({{within[j], 1}, {block_pre[b - 1], 1}}), nz);
        }
        return out;
    }
    // outputs: found bit, then n_digits base-M digits (little endian) of the first matching offset
    // from_end: select the LAST matching offset instead of the first (rfind)
    void find_from_matches(const std::vector<uint32_t>& match_in, uint32_t n_digits, std::vector<uint32_t>& out,
                           bool from_end = false) {
        std::vector<uint32_t> match(match_in);
        if (from_end) std::reverse(match.begin(), match.end());
        // first[o] = match[o] AND no match before o.  "Before o" = earlier in o's block of the scan (prefix inside the
        // block) or in an earlier block (prefix over the block totals): the two are fed to the lookup side by side,
        // (match[o] + 2 * before_in_block + 4 * before_in_earlier_blocks) == 1, so the full prefix is never materialised
        // (one lookup level and n lookups less than prefix_or + a test against it).
        const uint32_t first_lut = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 1); });
        const uint32_t nz = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
        const uint32_t Fb = full_box_reduce ? T : T - 1;
        const size_t n = match.size();
        if (n == 0) {                                   // no candidate offset at all: not found, index 0
            out.push_back(c.trivial(0));
            for (uint32_t d = 0; d < n_digits; d++) out.push_back(c.trivial(0));
            return;
        }
        std::vector<uint32_t> within(n), block_tot;
        for (size_t b = 0; b < n; b += Fb) {
            std::vector<Term> run;
            for (size_t j = b; j < std::min(n, b + Fb); j++) {
                run.push_back({match[j], 1});
                within[j] = run.size() == 1 ? match[j] : run.size() == T ? c.pbs_full_box(c.lin(run), false) : c.pbs(c.lin(run), nz);
            }
            block_tot.push_back(within[std::min(n, b + Fb) - 1]);
        }
        const std::vector<uint32_t> block_pre = prefix_or(block_tot);      // inclusive, over the blocks
        const uint32_t found = block_pre.back();
        std::vector<uint32_t> first(n);
        for (size_t o = 0; o < n; o++) {
            const size_t b = o / Fb, r = o % Fb;
            std::vector<Term> t{{match[o], 1}};
            if (T >= 8) {
                if (r > 0) t.push_back({within[o - 1], 2});
                if (b > 0) t.push_back({block_pre[b - 1], 4});
            } else if (r > 0 && b > 0) {
                // boxes of fewer than 8 values (PARAM_MESSAGE_1_CARRY_1: T = 4) cannot hold the three bits side by side
                // (ADVICE r3): the two "before" bits are OR-ed by one more lookup, then match + 2 * before <= 3
                t.push_back({c.pbs(c.lin({{within[o - 1], 1}, {block_pre[b - 1], 1}}), nz), 2});
            } else if (r > 0) {
                t.push_back({within[o - 1], 2});
            } else if (b > 0) {
                t.push_back({block_pre[b - 1], 2});
            }
            first[o] = t.size() == 1 ? match[o] : c.pbs(c.lin(t), first_lut);
        }
        if (from_end) std::reverse(first.begin(), first.end());   // one-hot vector back on the original offsets
        out.push_back(found);
        const uint32_t F = T - 1;
        const uint32_t mm = M;
        const uint32_t idm = c.lut_fn([mm](uint64_t x) { return x % mm; });
        for (uint32_t d = 0; d < n_digits; d++) {
            // digit d of the index = sum_o first[o] * digit_d(o): at most one term is non-zero, so the
            // value stays <= M-1 whatever the fan-in; what limits a group is the noise, dig^2 per term
            std::vector<Term> terms;
            for (size_t o = 0; o < first.size(); o++) {
                const uint32_t dig = (uint32_t)((o >> (d * bits_per_block)) & (M - 1));
                if (dig) terms.push_back({first[o], (int32_t)dig});
            }
            std::vector<uint32_t> cleaned;
            for (const auto& g : term_groups(terms, F)) cleaned.push_back(c.pbs(c.lin(g, 0, (int64_t)M - 1), idm));
            while (cleaned.size() > 1) {   // again at most one part is non-zero
                std::vector<Term> parts;
                for (uint32_t p : cleaned) parts.push_back({p, 1});
                std::vector<uint32_t> next;
                for (const auto& g : term_groups(parts, F)) next.push_back(c.pbs(c.lin(g, 0, (int64_t)M - 1), idm));
                cleaned.swap(next);
            }
            out.push_back(cleaned.empty() ? c.trivial(0) : cleaned[0]);
        }
    }

    // ---- case conversion (docs/tutorials/ascii_fhe_string.md:88-131 semantics) ----
    // to_lower: c in ['A','Z'] -> c + 32 ; to_upper: c in ['a','z'] -> c - 32.  32 = 2 * 16 only
    // touches bit 5, i.e. block (5 / bits_per_block); letters never carry out of that block.
    void change_case(const Str& s, bool to_lower, std::vector<uint32_t>& out) {
        const uint32_t lo_first = to_lower ? 'A' : 'a', lo_last = to_lower ? 'Z' : 'z';
        const uint32_t hiA = lo_first >> 4, hiB = lo_last >> 4;          // 4 / 5  or  6 / 7
        const uint32_t loA = lo_first & 15, loB = lo_last & 15;         // 1 and 10
        const uint32_t blk = 5 / bits_per_block, bit_in_blk = 5 % bits_per_block;
        const uint32_t addv = 1u << bit_in_blk;
        // level 1: classify high nibble (0 / 1 = first row / 2 = second row) and low nibble
        // (bit0 = lo >= loA, bit1 = lo <= loB); level 2: combine into addv * is_letter
        const uint32_t hi_lut = c.lut_fn([hiA, hiB](uint64_t x) { return (uint64_t)(x == hiA ? 1 : (x == hiB ? 2 : 0)); });
        const uint32_t lo_lut = c.lut_fn([loA, loB](uint64_t x) { return (uint64_t)((x >= loA ? 1 : 0) | (x <= loB ? 2 : 0)); });
        const uint32_t comb = c.lut_fn([addv](uint64_t x) {
            const uint64_t a = x / 4, b = x % 4;
            const bool is = (a == 1 && (b & 1)) || (a == 2 && (b & 2));
            return (uint64_t)(is ? addv : 0);
        });
        const uint32_t half = bpc / 2;   // blocks per nibble (2 for 2-bit blocks)
        // whole-char form: the block that holds bit 5, already converted
        const uint32_t Mv = M, shift = blk * bits_per_block;
        const uint32_t whole_lut = c.lut_fn([=](uint64_t x) {
            const uint64_t block = (x >> shift) & (Mv - 1);
            const bool is = x >= lo_first && x <= lo_last;
            return is ? (to_lower ? block + addv : block - addv) : block;
        });
        for (uint32_t i = 0; i < s.cap; i++) {
            Scope sc(c, owner_for(i, s.cap));
            const auto& b = s.ch[i];
            if (whole_char_fits(b)) {
                const uint32_t converted = c.pbs(whole_char(b), whole_lut);
                for (uint32_t k = 0; k < bpc; k++) out.push_back(k == blk ? converted : b[k]);
                continue;
            }
            std::vector<Term> lo_terms, hi_terms;
            for (uint32_t k = 0; k < half; k++) {
                lo_terms.push_back({b[k], (int32_t)(1u << (k * bits_per_block))});
                hi_terms.push_back({b[half + k], (int32_t)(1u << (k * bits_per_block))});
            }
            const uint32_t hc = c.pbs(c.lin(hi_terms), hi_lut);
            const uint32_t lc = c.pbs(c.lin(lo_terms), lo_lut);
            const uint32_t delta = c.pbs(c.lin({{hc, 4}, {lc, 1}}), comb);
            for (uint32_t k = 0; k < bpc; k++) {
                if (k == blk) out.push_back(c.lin({{b[k], 1}, {delta, to_lower ? 1 : -1}}, 0, (int64_t)M - 1));   // letters never carry
                else out.push_back(b[k]);
            }
        }
    }

    // ---- whitespace trimming ----
    // wz[i] = [s[i] is ASCII whitespace (9..13, 32)] (+ [s[i] == 0] when null_too): 3 PBS per char
    std::vector<uint32_t> whitespace_bits(const Str& s, bool null_too) {
        const uint32_t hi_lut = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 0 ? 1 : (x == 2 ? 2 : 0)); });
        const uint32_t lo_lut = c.lut_fn([null_too](uint64_t x) {
            const bool row0 = (x >= 9 && x <= 13) || (null_too && x == 0);
            return (uint64_t)((row0 ? 1 : 0) | (x == 0 ? 2 : 0));
        });
        const uint32_t comb = c.lut_fn([](uint64_t x) {
            const uint64_t a = x / 4, b = x % 4;
            return (uint64_t)(((a == 1 && (b & 1)) || (a == 2 && (b & 2))) ? 1 : 0);
        });
        const uint32_t half = bpc / 2;
        const uint32_t whole_lut = c.lut_fn([null_too](uint64_t x) {
            return (uint64_t)(((x >= 9 && x <= 13) || x == 32 || (null_too && x == 0)) ? 1 : 0);
        });
        std::vector<uint32_t> out;
        for (uint32_t i = 0; i < s.cap; i++) {
            Scope sc(c, owner_for(i, s.cap));
            if (whole_char_fits(s.ch[i])) {
                out.push_back(c.pbs(whole_char(s.ch[i]), whole_lut));
                continue;
            }
            std::vector<Term> lo_terms, hi_terms;
            for (uint32_t k = 0; k < half; k++) {
                lo_terms.push_back({s.ch[i][k], (int32_t)(1u << (k * bits_per_block))});
                hi_terms.push_back({s.ch[i][half + k], (int32_t)(1u << (k * bits_per_block))});
            }
            const uint32_t hc = c.pbs(c.lin(hi_terms), hi_lut), lc = c.pbs(c.lin(lo_terms), lo_lut);
            out.push_back(c.pbs(c.lin({{hc, 4}, {lc, 1}}), comb));
        }
        return out;
    }
    // block * bit (bit in {0,1}) : LUT on bit + 2*block  (noise: 1 + 2*level(block))
    uint32_t gate_block(uint32_t block, uint32_t bit, bool keep_if_set) {
        int64_t v = 0;
        if (is_trivial(block, &v) && v == 0) return block;                       // 0 * anything
        if (is_trivial(bit, &v)) return ((v != 0) == keep_if_set) ? block : c.trivial(0);
        const uint32_t m2 = 2 * M;
        const uint32_t l = c.lut_fn([keep_if_set, m2](uint64_t x) {
            return (uint64_t)((x < m2 && ((x & 1) != 0) == keep_if_set) ? x >> 1 : 0);
        });
        return c.pbs(c.lin({{bit, 1}, {block, 2}}), l);
    }
    uint32_t not_bit(uint32_t bit) { return c.lin({{bit, -1}}, 1, 1); }
    // sel ? b : a for two blocks in [0, M-1] in ONE lookup (default plans): a + sel * (b - a), the product read off
    // (b - a) + (M-1) + (2M-1) sel in [0, 4M-3].  The result keeps a's noise plus one nominal variance (the two-gate form
    // returns two fresh lookups): callers that chain it clean the blocks at the end (fresh()).  Falls back to the two
    // gates where the packing does not fit the box or the budget (PARAM_MESSAGE_4_CARRY_4: (2M-1)^2 = 961 variances).
    uint32_t select_block(uint32_t a, uint32_t b, uint32_t sel, uint32_t hi) {
        int64_t v = 0;
        if (is_trivial(sel, &v)) return v ? b : a;
        const uint32_t D = 2 * hi + 1;                           // distinct values of b - a, both in [0, hi]
        const uint32_t x = full_box_reduce && 2 * D <= T ? c.lin({{b, 1}, {a, -1}, {sel, (int32_t)D}}, (int64_t)hi) : UINT32_MAX;
        if (x == UINT32_MAX || c.node(x).noise > budget())       // (noise of the combination as built: shared sources add up)
            return c.lin({{gate_block(a, sel, false), 1}, {gate_block(b, sel, true), 1}}, 0, hi);
        const uint32_t l = c.lut_fn([D, hi](uint64_t y) { return (uint64_t)(y >= 2 * D ? hi : (y / D ? y % D : hi)); });
        return c.lin({{a, 1}, {c.pbs(x, l), 1}}, -(int64_t)hi, (int64_t)hi);
    }
    // trim_end: zero every char from the last non-whitespace one onwards
    Str trim_end(const Str& s) {
        std::vector<uint32_t> wz = whitespace_bits(s, true), nw(s.cap);
        for (uint32_t i = 0; i < s.cap; i++) nw[s.cap - 1 - i] = not_bit(wz[i]);   // reversed, 1 = real char
        std::vector<uint32_t> keep_rev = prefix_or(nw);        // keep[i] = OR_{j >= i} nonws[j]
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++)
            for (uint32_t k = 0; k < bpc; k++) out.ch[i].push_back(gate_block(s.ch[i][k], keep_rev[s.cap - 1 - i], true));
        return out;
    }
    // trim_start: shift left by the (encrypted) number of leading whitespace chars with a barrel
    // shifter; stage t shifts by 2^t iff the first 2^t chars are all whitespace
    Str trim_start(const Str& s) {
        std::vector<uint32_t> ws = whitespace_bits(s, false), nws(s.cap);
        for (uint32_t i = 0; i < s.cap; i++) nws[i] = not_bit(ws[i]);
        std::vector<uint32_t> seen = prefix_or(nws);           // seen[i] = some non-ws char in [0, i]
        std::vector<uint32_t> lead(s.cap);                     // lead[i] = chars 0..i all whitespace
        for (uint32_t i = 0; i < s.cap; i++) lead[i] = not_bit(seen[i]);
        return shift_left_by_leading(s, lead);
    }
    // Shift `s` left by the number of leading ones of the monotone indicator lead (lead[i] = 1 for
    // i < z, 0 after): barrel shifter, stage t shifts by 2^t iff lead[2^t - 1] (i.e. z >= 2^t), the
    // indicator shifts along with the string.  (cmux per block: integer/server_key/radix_parallel/cmux.rs)
    Str shift_left_by_leading(const Str& s, std::vector<uint32_t> lead) {
        Str cur = s;
        int top = 0;
        while ((1u << (top + 1)) <= s.cap) top++;
        for (int t = top; t >= 0; t--) {
            const uint32_t sh = 1u << t;
            if (sh > s.cap) continue;
            const uint32_t sel = lead[sh - 1];
            int64_t sv = 0;
            if (is_trivial(sel, &sv) && sv == 0) continue;       // the indicator is known to be shorter
            Str nxt;
            nxt.cap = s.cap;
            nxt.ch.resize(s.cap);
            std::vector<uint32_t> nlead(s.cap);
            for (uint32_t i = 0; i < s.cap; i++) {
                for (size_t k = 0; k < cur.ch[i].size(); k++) {
                    if (i + sh < s.cap) nxt.ch[i].push_back(select_block(cur.ch[i][k], cur.ch[i + sh][k], sel, M - 1));
                    else nxt.ch[i].push_back(gate_block(cur.ch[i][k], sel, false));
                }
                // the monotone indicator shifts with the string
                // (a fresh bit every stage -- it is the next stages' selector, whose noise is weighted (2M-1)^2 above:
                //  one lookup on lead[i] + 2 lead[i + sh] + 4 sel)
                // (the noise of the combination as built: operands that share sources, e.g. lead[i] == sel, add up)
                const uint32_t packed3 = i + sh < s.cap && full_box_reduce && T >= 8
                                             ? c.lin({{lead[i], 1}, {lead[i + sh], 2}, {sel, 4}}) : UINT32_MAX;
                if (packed3 != UINT32_MAX && c.node(packed3).noise <= budget()) {
                    int64_t va = 0, vb = 0;
                    const bool ta = is_trivial(lead[i], &va), tb = is_trivial(lead[i + sh], &vb);
                    if (ta && tb && va == vb) nlead[i] = lead[i];
                    else nlead[i] = c.pbs(packed3, c.lut_fn([](uint64_t x) { return (uint64_t)((x & 4) ? (x >> 1) & 1 : x & 1); }));
                } else
                nlead[i] = i + sh < s.cap ? select_block(lead[i], lead[i + sh], sel, 1) : c.lin({{gate_block(lead[i], sel, false), 1}}, 0, 1);
            }
            cur = nxt;
            lead = nlead;
        }
        // blocks that went through select_block stages carry one nominal variance per stage: back to a fresh lookup's
        for (auto& chr : cur.ch)
            for (uint32_t& b : chr)
                if (c.node(b).noise > 2.0) b = fresh(b);
        return cur;
    }
    // Concatenation of two left-justified strings whose per-slot occupancy bits are known (occ): the
    // right operand is parked behind the left one's capacity and shifted left by the left one's free
    // slots; the occupancy travels as one more block.  No char_is_zero needed.
    Str concat_occ(const Str& a, const Str& b) {
        const uint32_t zero = c.trivial(0);
        Str u;
        u.cap = a.cap + b.cap;
        u.ch.resize(u.cap);
        const size_t nb = a.cap ? a.ch[0].size() : b.ch[0].size();
        for (uint32_t i = 0; i < a.cap; i++) u.ch[i].assign(nb + 1, zero);
        for (uint32_t j = 0; j < b.cap; j++) { u.ch[a.cap + j] = b.ch[j]; u.ch[a.cap + j].push_back(b.occ[j]); }
        std::vector<uint32_t> lead(u.cap, zero);
        for (uint32_t t = 0; t < a.cap; t++) lead[t] = not_bit(a.occ[a.cap - 1 - t]);   // 1 while in a's free tail
        Str v = shift_left_by_leading(u, lead);
        Str out;
        out.cap = u.cap;
        out.ch.resize(u.cap);
        out.occ.resize(u.cap);
        for (uint32_t i = 0; i < u.cap; i++) {
            for (size_t k = 0; k < nb; k++)
                out.ch[i].push_back(i < a.cap ? c.lin({{a.ch[i][k], 1}, {v.ch[i][k], 1}}, 0, M - 1) : v.ch[i][k]);
            out.occ[i] = i < a.cap ? c.lin({{a.occ[i], 1}, {v.ch[i][nb], 1}}, 0, 1) : v.ch[i][nb];
        }
        return out;
    }
    // concat: a ++ b without a's padding.  b is parked behind a's full capacity and shifted left by the
    // number of null characters at the end of a (monotone indicator read from the end of a); the result
    // (capacity a.cap + b.cap) is a + shifted(b): wherever b lands, a is null.
    Str concat(const Str& a, const Str& b) {
        Str u;
        u.cap = a.cap + b.cap;
        u.ch.resize(u.cap);
        const uint32_t zero = c.trivial(0);
        for (uint32_t i = 0; i < a.cap; i++) u.ch[i].assign(bpc, zero);
        for (uint32_t j = 0; j < b.cap; j++) u.ch[a.cap + j] = b.ch[j];
        std::vector<uint32_t> lead(u.cap, zero);
        for (uint32_t t = 0; t < a.cap; t++) lead[t] = char_is_zero(a.ch[a.cap - 1 - t]);   // 1 while still in a's padding
        Str v = shift_left_by_leading(u, lead);
        Str out;
        out.cap = u.cap;
        out.ch.resize(u.cap);
        for (uint32_t i = 0; i < u.cap; i++)
            for (uint32_t k = 0; k < bpc; k++)
                out.ch[i].push_back(i < a.cap ? c.lin({{a.ch[i][k], 1}, {v.ch[i][k], 1}}, 0, M - 1) : v.ch[i][k]);
        return out;
    }
    Str clear_string(const uint8_t* bytes, uint32_t len) {
        Str t;
        t.cap = len;
        t.ch.resize(len);
        for (uint32_t i = 0; i < len; i++)
            for (uint32_t k = 0; k < bpc; k++) t.ch[i].push_back(c.trivial(clear_block(bytes[i], k)));
        return t;
    }

    // ---- replace ----
    uint32_t and_bits(uint32_t a, uint32_t b) {
        int64_t v = 0;
        if (is_trivial(a, &v)) return v ? b : a;
        if (is_trivial(b, &v)) return v ? a : b;
        return c.pbs(c.lin({{a, 1}, {b, 1}}), c.lut_fn([](uint64_t x) { return (uint64_t)(x == 2); }));
    }
    // Occurrence bookkeeping of a replace: which offsets are selected (leftmost, non-overlapping) and
    // which characters they cover.  An occurrence selected at o - j still runs at o iff j < |from|:
    // always for a clear / unpadded pattern of length m, and iff nzf[j] = [from[j] != 0] for a padded
    // encrypted one (nzf == nullptr: unpadded).
    struct Occurrences {
        std::vector<uint32_t> sel;     // [offset]
        std::vector<uint32_t> cover;   // [char] 0/1: inside a selected occurrence
    };
    // The leftmost non-overlapping occurrences as a BLOCKED SCAN (round 4).  The recurrence below is a chain over the offsets,
    // one lookup level per two of them (n / 2 levels: 0.33 s for 256 characters on PARAM_MESSAGE_2_CARRY_2, each level a single-
    // PBS latency).  Its state is small: r[o] = how many more characters the occurrence selected before o still covers, in
    // [0, m).  One step is ONE lookup,   r[o+1] = (r[o] == 0) ? g[o] : r[o] - 1   on   x = r[o] + m g[o],
    // with g[o] = match[o] * (L - 1) (L = the pattern's length: m, or the hidden number of non-null characters of a padded
    // encrypted pattern -- then g costs one lookup per offset, off the chain).  Blocks of B offsets run their chains for every
    // possible incoming state at once (m hypotheses, constants at the block's start), the true state then hops from block to
    // block through the blocks' state maps (m lookups per hop: select F_b[R_b] by the encrypted R_b), and every block re-runs
    // its chain from its true incoming state.  Depth 2 B + n / B instead of n / 2 -- B + n / B with two offsets per chain step
    // (1024 characters, 4-character pattern: 66 levels instead of 511) --, about (m + 1) n chain lookups instead of n.  sel[o] = [r[o] == 0 and match[o]] and cover[o] = [r[o] > 0 or
    // match[o]] are read off x' = r[o] + m match[o] afterwards.  Needs m^2 <= T (hidden length) or 2 m <= T; the recurrence
    // stays for longer patterns, short strings and where the noise does not fit.
    bool blocked_scan = true;
    bool occurrences_scan(const std::vector<uint32_t>& match, uint32_t m, uint32_t n_chars, const std::vector<uint32_t>* nzf, Occurrences& oc) {
        const uint32_t n = (uint32_t)match.size();
        if (!blocked_scan || m < 2 || n < 48) return false;
        const bool hidden = nzf != nullptr;
        if (hidden ? (uint64_t)m * m > T : 2ull * m > T) return false;
        for (uint32_t o = 0; o < n; o++) if (is_trivial(match[o])) return false;      // (plans with clear operands: keep the recurrence)
        const uint32_t mm = m;
        std::vector<uint32_t> g(n);
        if (hidden) {
            std::vector<Term> t;
            double nu = 0;
            for (uint32_t j = 1; j < m; j++) { t.push_back({(*nzf)[j], 1}); nu += c.node((*nzf)[j]).noise; }
            const uint32_t lm1 = c.lin(t, 0, m - 1);          // L - 1 (padding sits at the end; an empty pattern matches nowhere)
            const uint32_t gl = c.lut_fn([mm](uint64_t x) { return (x & 1) ? std::min<uint64_t>(x >> 1, mm - 1) : (uint64_t)0; });
            for (uint32_t o = 0; o < n; o++) {
                if (c.node(match[o]).noise + 4 * nu > budget()) return false;
                Scope sc(c, owner_for(o, n));
                g[o] = c.pbs(c.lin({{match[o], 1}, {lm1, 2}}), gl);
            }
            if (1.0 + (double)m * m * 1.0 + (double)m > budget()) return false;    // chain input: r (up to m outputs summed) + m g
        } else {
            g = match;
            for (uint32_t o = 0; o < n; o++)
                if ((double)m + (double)m * m * c.node(match[o]).noise > budget()) return false;
        }
        const uint32_t step = hidden ? c.lut_fn([mm](uint64_t x) { return x % mm == 0 ? x / mm : x % mm - 1; })
                                     : c.lut_fn([mm](uint64_t x) { const uint64_t r = x % mm; return r == 0 ? (x / mm ? (uint64_t)mm - 1 : (uint64_t)0) : r - 1; });
        const uint32_t n_ext = std::min(n_chars, n + m - 1);      // past the last offset an occurrence may still run
        const uint32_t zero = c.trivial(0);
        auto g_at = [&](uint32_t o) { return o < n ? g[o] : zero; };
        // Two offsets per chain step where the box and the noise allow it: x = r + m g[o] + W g[o+1], W = m^2 (hidden length:
        // g in [0, m)) or 2 m (known length: g is the match bit) -- known length m = 4 fills the 16 values of PARAM_MESSAGE_2_CARRY_2
        // exactly.  The state between the two (needed for sel / cover on the true chains only) is one more lookup off the chain.
        const uint32_t W = hidden ? m * m : 2 * m;
        const bool pairs = (hidden ? (uint64_t)m * m * m : 4ull * m) <= T && (double)m + (double)m * m + (double)W * W <= budget();
        auto one = [mm, hidden](uint64_t r, uint64_t gv) -> uint64_t { return r == 0 ? (hidden ? gv : (gv ? (uint64_t)mm - 1 : (uint64_t)0)) : r - 1; };
        const uint32_t step2 = !pairs ? 0u : c.lut_fn([mm, W, one](uint64_t x) { return one(one(x % mm, (x % W) / mm), x / W); });
        // state entering lo, lo + 1, ..., hi (hi - lo + 1 values; only the last one unless `all`) from the state `start` entering lo
        auto chain = [&](uint32_t start, uint32_t lo, uint32_t hi, bool all) {
            std::vector<uint32_t> r{start};
            uint32_t cur = start;
            for (uint32_t o = lo; o < hi;) {
                if (pairs && o + 1 < hi) {
                    if (all) r.push_back(c.pbs(c.lin({{cur, 1}, {g_at(o), (int32_t)mm}}), step));
                    cur = c.pbs(c.lin({{cur, 1}, {g_at(o), (int32_t)mm}, {g_at(o + 1), (int32_t)W}}), step2);
                    o += 2;
                } else {
                    cur = c.pbs(c.lin({{cur, 1}, {g_at(o), (int32_t)mm}}), step);
                    o += 1;
                }
                if (all) r.push_back(cur);
            }
            if (!all) r.assign(1, cur);
            return r;
        };
        uint32_t B = 1;
        while ((pairs ? 1ull : 2ull) * B * B < n_ext) B++;        // minimises (2 B or B) + n / B
        if (p_large()) B = std::max(B, std::min<uint32_t>(64, n_ext));      // N >= 16384: a level costs what its lookups cost, fewer hypotheses' worth of them
        const uint32_t nb = (n_ext + B - 1) / B;
        std::vector<uint32_t> r_true(n_ext + 1);
        {
            const std::vector<uint32_t> r0 = chain(zero, 0, std::min(B, n_ext), true);
            std::copy(r0.begin(), r0.end(), r_true.begin());
        }
        uint32_t R = r_true[std::min(B, n_ext)];                  // true state entering block 1
        for (uint32_t b = 1; b < nb; b++) {
            const uint32_t lo = b * B, hi = std::min(n_ext, lo + B);
            const std::vector<uint32_t> rt = chain(R, lo, hi, true);
            std::copy(rt.begin(), rt.end(), r_true.begin() + lo);
            if (b + 1 < nb) {                                      // the block's state map, then the hop to the next block
                std::vector<Term> parts;
                for (uint32_t sidx = 0; sidx < m; sidx++) {
                    const uint32_t f = chain(c.trivial(sidx), lo, hi, false).back();
                    const uint32_t pick = c.lut_fn([mm, sidx](uint64_t x) { return x % mm == sidx ? x / mm : (uint64_t)0; });
                    parts.push_back({c.pbs(c.lin({{R, 1}, {f, (int32_t)mm}}), pick), 1});
                }
                R = c.lin(parts, 0, m - 1);                        // exactly one hypothesis is the true one
            }
        }
        const uint32_t is_sel = c.lut_fn([mm](uint64_t x) { return (uint64_t)(x == mm); });
        const uint32_t nzl = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
        oc.sel.resize(n);
        oc.cover.assign(n_chars, zero);
        for (uint32_t i = 0; i < n_ext; i++) {
            Scope sc(c, owner_for(i, n_ext));
            if (i < n) {
                const uint32_t x = c.lin({{r_true[i], 1}, {match[i], (int32_t)mm}});
                oc.sel[i] = i == 0 ? match[0] : c.pbs(x, is_sel);
                oc.cover[i] = i == 0 ? match[0] : c.pbs(x, nzl);
            } else {
                oc.cover[i] = c.pbs(r_true[i], nzl);
            }
        }
        return true;
    }
    Occurrences occurrences(const std::vector<uint32_t>& match, uint32_t m, uint32_t n_chars, bool may_overlap,
                            const std::vector<uint32_t>* nzf) {
        Occurrences oc;
        if (may_overlap && occurrences_scan(match, m, n_chars, nzf, oc)) return oc;
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> running;   // (offset, j) -> sel[offset] AND nzf[j]
        auto run_bit = [&](uint32_t o, uint32_t j) {
            if (!nzf || j == 0) return oc.sel[o];
            auto key = std::make_pair(o, j);
            auto it = running.find(key);
            if (it == running.end()) it = running.emplace(key, and_bits(oc.sel[o], (*nzf)[j])).first;
            return it->second;
        };
        oc.sel.resize(match.size());
        const uint32_t l3 = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 3); });
        // Two offsets per lookup level (default plans).  sel[o+1] needs sel[o], but sel[o] = match[o] AND NOT B with
        // B = [an earlier occurrence still runs at o], so with p = match[o] AND [the pattern is longer than one character]
        // (off the chain) and B' = [an earlier occurrence still runs at o + 1] (B' <= B: the same occurrence, one character on):
        //   sel[o+1] = match[o+1] AND NOT (p AND NOT B) AND NOT B',   one lookup on  (B + B') + 3 p + 6 match[o+1]  in [0, 11]
        // at the same level as sel[o].  Halves the depth of the recurrence (one level per TWO offsets).
        const uint32_t l_second = c.lut_fn([](uint64_t x) {
            const uint64_t mt = x / 6, pp = (x % 6) / 3, t = x % 3;
            return (uint64_t)(mt == 1 && !(pp == 1 && t == 0) && t != 2);
        });
        for (uint32_t o = 0; o < match.size(); o++) {
            std::vector<uint32_t> blockers;
            if (may_overlap)
                for (uint32_t j = 1; j < m && j <= o; j++) blockers.push_back(run_bit(o - j, j));
            if (blockers.empty()) { oc.sel[o] = match[o]; continue; }
            // x = 2 match + 1 - (blockers; at most one is set) in {0..3}; selected iff x == 3
            double nu = 4 * c.node(match[o]).noise;
            for (uint32_t b : blockers) nu += c.node(b).noise;
            const bool reduced = nu > budget();
            if (reduced) blockers.assign(1, reduce_local(blockers, false));
            std::vector<Term> terms{{match[o], 2}};
            for (uint32_t b : blockers) terms.push_back({b, -1});
            oc.sel[o] = c.pbs(c.lin(terms, 1, 3), l3);
            if (!full_box_reduce || reduced || o + 1 >= match.size() || T < 12) continue;
            // the second offset of the pair, from what was known before sel[o]
            std::vector<Term> tt;
            double nu2 = 36 * c.node(match[o + 1]).noise;
            for (uint32_t b : blockers) { tt.push_back({b, 1}); nu2 += c.node(b).noise; }
            for (uint32_t j = 2; j < m && j <= o + 1; j++) {
                const uint32_t b = run_bit(o + 1 - j, j);
                tt.push_back({b, 1});
                nu2 += c.node(b).noise;
            }
            const uint32_t p_bit = nzf ? and_bits(match[o], (*nzf)[1]) : match[o];
            nu2 += 9 * c.node(p_bit).noise;
            if (nu2 > budget()) continue;
            tt.push_back({p_bit, 3});
            tt.push_back({match[o + 1], 6});
            oc.sel[o + 1] = c.pbs(c.lin(tt, 0, 11), l_second);
            o++;
        }
        oc.cover.resize(n_chars);
        for (uint32_t i = 0; i < n_chars; i++) {
            std::vector<Term> terms;
            for (uint32_t j = 0; j < m && j <= i; j++)
                if (i - j < oc.sel.size()) terms.push_back({run_bit(i - j, j), 1});
            oc.cover[i] = c.lin(terms, 0, 1);                                  // at most one occurrence covers i
            if (terms.size() > 8) oc.cover[i] = c.pbs(oc.cover[i], c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); }));
        }
        return oc;
    }
    // sel * (clear block v): one lookup on the selector (the linear form v * sel would cost v^2 in noise)
    uint32_t scale_bit(uint32_t bit, uint32_t v) {
        if (v == 0) return c.trivial(0);
        int64_t b = 0;
        if (is_trivial(bit, &b)) return c.trivial(b ? (int64_t)v : 0);
        return c.pbs(bit, c.lut_fn([v](uint64_t x) { return (uint64_t)(x == 1 ? v : 0); }));
    }
    // Equal lengths (|from| = |to| = m, `to` clear or unpadded encrypted): the string keeps its shape, every
    // covered character is rewritten in place.
    Str replace_in_place(const Str& s, const Occurrences& oc, uint32_t m, const uint8_t* to_clear, const Str* to_enc) {
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++) {
            Scope sc(c, owner_for(i, s.cap));
            if (is_trivial(oc.cover[i])) { out.ch[i] = s.ch[i]; continue; }
            for (uint32_t k = 0; k < bpc; k++) {
                std::vector<Term> terms{{gate_block(s.ch[i][k], oc.cover[i], false), 1}};
                std::vector<Term> lin_contrib;      // clear `to`: v_j * sel[i - j], cheap while its noise fits
                double nu = c.node(terms[0].node).noise;
                for (uint32_t j = 0; j < m && j <= i; j++) {
                    if (i - j >= oc.sel.size()) continue;
                    if (to_clear) {
                        const uint32_t v = clear_block(to_clear[j], k);
                        if (v) { lin_contrib.push_back({oc.sel[i - j], (int32_t)v}); nu += (double)v * v * c.node(oc.sel[i - j]).noise; }
                    } else {
                        terms.push_back({gate_block(to_enc->ch[j][k], oc.sel[i - j], true), 1});
                    }
                }
                if (nu <= budget()) terms.insert(terms.end(), lin_contrib.begin(), lin_contrib.end());
                else for (const Term& t : lin_contrib) terms.push_back({scale_bit(t.node, (uint32_t)t.coeff), 1});
                out.ch[i].push_back(c.lin(terms, 0, M - 1));   // at most one contribution is non-zero
            }
        }
        return out;
    }
    // Any lengths (clear or encrypted, padded or not): every position becomes a small left-justified
    // piece -- the replacement if an occurrence is selected there, the character itself if it is kept,
    // nothing if it is covered -- and the pieces are concatenated pairwise (log2 n rounds of the
    // occupancy-driven barrel shifter).  Result capacity = s.cap * max(1, |to| capacity), cut / padded
    // to out_cap.
    Str replace_general(const Str& s, const Occurrences& oc, const uint8_t* to_clear, uint32_t to_len, const Str* to_enc,
                        uint32_t out_cap) {
        const uint32_t w = std::max<uint32_t>(1, to_enc ? to_enc->cap : to_len);
        const uint32_t zero = c.trivial(0);
        std::vector<uint32_t> nzt;
        if (to_enc) for (uint32_t j = 0; j < to_enc->cap; j++) nzt.push_back(not_bit(char_is_zero(to_enc->ch[j])));
        const uint32_t l2 = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 2); });
        std::vector<Str> cur;
        for (uint32_t i = 0; i < s.cap; i++) {
            Scope sc(c, owner_for(i, s.cap));
            const bool has_sel = i < oc.sel.size();
            const uint32_t nz = not_bit(char_is_zero(s.ch[i]));
            // kept = this slot holds a character that survives: s[i] != 0 and not covered
            const uint32_t kept = is_trivial(oc.cover[i]) ? nz : c.pbs(c.lin({{nz, 1}, {oc.cover[i], -1}}, 1), l2);
            Str piece;
            piece.cap = w;
            piece.ch.resize(w);
            piece.occ.resize(w);
            for (uint32_t j = 0; j < w; j++) {
                const bool to_slot = has_sel && j < (to_enc ? to_enc->cap : to_len);
                for (uint32_t k = 0; k < bpc; k++) {
                    uint32_t contrib = zero;
                    if (to_slot) contrib = to_enc ? gate_block(to_enc->ch[j][k], oc.sel[i], true)
                                                  : scale_bit(oc.sel[i], clear_block(to_clear[j], k));
                    if (j == 0) piece.ch[j].push_back(c.lin({{gate_block(s.ch[i][k], oc.cover[i], false), 1}, {contrib, 1}}, 0, M - 1));
                    else piece.ch[j].push_back(contrib);
                }
                const uint32_t sel_occ = !to_slot ? zero : (to_enc ? and_bits(oc.sel[i], nzt[j]) : oc.sel[i]);
                piece.occ[j] = j == 0 ? c.lin({{kept, 1}, {sel_occ, 1}}, 0, 1) : sel_occ;
            }
            cur.push_back(piece);
        }
        return fit(concat_tree(cur), out_cap);
    }
    // Empty clear pattern (Rust's str::replace("", to) / bytes.replace(b"", to)): `to` before every
    // character and after the last one.
    Str replace_empty_pattern(const Str& s, const uint8_t* to_clear, uint32_t t, uint32_t out_cap) {
        if (t == 0) return fit(s, out_cap);
        const uint32_t one = c.trivial(1);
        std::vector<uint32_t> nz;
        for (uint32_t i = 0; i < s.cap; i++) nz.push_back(not_bit(char_is_zero(s.ch[i])));
        std::vector<Str> cur;
        for (uint32_t i = 0; i <= s.cap; i++) {
            Scope sc(c, owner_for(std::min(i, s.cap - 1), s.cap));
            const uint32_t g = i == 0 ? one : nz[i - 1];       // position i is still inside (or right after) the string
            Str piece;
            piece.cap = t + (i < s.cap ? 1 : 0);
            piece.ch.resize(piece.cap);
            piece.occ.resize(piece.cap);
            for (uint32_t j = 0; j < t; j++) {
                for (uint32_t k = 0; k < bpc; k++) piece.ch[j].push_back(scale_bit(g, clear_block(to_clear[j], k)));
                piece.occ[j] = g;
            }
            if (i < s.cap) { piece.ch[t] = s.ch[i]; piece.occ[t] = nz[i]; }
            cur.push_back(piece);
        }
        return fit(concat_tree(cur), out_cap);
    }
    Str concat_tree(std::vector<Str> cur) {
        while (cur.size() > 1) {
            std::vector<Str> next;
            for (size_t i = 0; i + 1 < cur.size(); i += 2) next.push_back(concat_occ(cur[i], cur[i + 1]));
            if (cur.size() & 1) next.push_back(cur.back());
            cur.swap(next);
        }
        return cur[0];
    }
    // cut or zero-pad to `cap` characters
    Str fit(const Str& r, uint32_t cap) {
        Str out;
        out.cap = cap;
        out.ch.resize(cap);
        const uint32_t zero = c.trivial(0);
        for (uint32_t i = 0; i < cap; i++) {
            if (i < r.cap) out.ch[i].assign(r.ch[i].begin(), r.ch[i].begin() + bpc);
            else out.ch[i].assign(bpc, zero);
        }
        return out;
    }
    static bool has_border(const uint8_t* p, uint32_t m) {   // proper prefix == suffix => matches may overlap
        for (uint32_t b = 1; b < m; b++)
            if (std::memcmp(p, p + m - b, b) == 0) return true;
        return false;
    }
    // ---- length, emptiness, case-insensitive equality, prefix / suffix stripping ----
    uint32_t is_empty(const Str& s) { return char_is_zero(s.ch[0]); }
    // len = offset of the first null char (cap when there is none): found bit is always 1
    void len(const Str& s, uint32_t n_digits, std::vector<uint32_t>& out) {
        std::vector<uint32_t> z;
        for (uint32_t i = 0; i < s.cap; i++) z.push_back(char_is_zero(s.ch[i]));
        z.push_back(c.trivial(1));
        std::vector<uint32_t> tmp;
        find_from_matches(z, n_digits, tmp);
        out.assign(tmp.begin() + 1, tmp.end());
    }
    Str case_folded(const Str& s) {
        std::vector<uint32_t> blocks;
        change_case(s, true, blocks);
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++) out.ch[i].assign(blocks.begin() + (size_t)i * bpc, blocks.begin() + (size_t)(i + 1) * bpc);
        return out;
    }
    // sel ? a : b on blocks (if_then_else without the final message_extract, integer/.../cmux.rs:194-248)
    uint32_t select_block(uint32_t sel, uint32_t a, uint32_t b) {
        return c.lin({{gate_block(a, sel, true), 1}, {gate_block(b, sel, false), 1}}, 0, M - 1);
    }
    // strip_prefix (clear pattern of length m): if s starts with it, drop it (shift left by m)
    Str strip_prefix_clear(const Str& s, const uint8_t* pat, uint32_t m, uint32_t* stripped_bit) {
        const uint32_t sel = starts_with_clear(s, pat, m);
        *stripped_bit = sel;
        if (m == 0 || m > s.cap) return s;
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++)
            for (uint32_t k = 0; k < bpc; k++)
                out.ch[i].push_back(i + m < s.cap ? select_block(sel, s.ch[i + m][k], s.ch[i][k]) : gate_block(s.ch[i][k], sel, false));
        return out;
    }
    // strip_suffix (clear pattern): zero every char from the matching offset on
    Str strip_suffix_clear(const Str& s, const uint8_t* pat, uint32_t m, uint32_t* stripped_bit) {
        if (m > s.cap) { *stripped_bit = c.trivial(0); return s; }
        if (m == 0) { *stripped_bit = c.trivial(1); return s; }
        std::vector<uint32_t> mt = window_matches_clear(s, pat, m, s.cap - m + 1), cand;
        for (uint32_t o = 0; o + m <= s.cap; o++) {
            if (o + m == s.cap) { cand.push_back(mt[o]); continue; }
            const uint32_t l = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 2); });
            cand.push_back(c.pbs(c.lin({{mt[o], 1}, {char_is_zero(s.ch[o + m]), 1}}), l));
        }
        std::vector<uint32_t> cut = prefix_or(cand);            // cut[i] = suffix starts at or before i
        *stripped_bit = cut.back();
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++)
            for (uint32_t k = 0; k < bpc; k++)
                out.ch[i].push_back(i < cut.size() ? gate_block(s.ch[i][k], cut[i], false)
                                                   : gate_block(s.ch[i][k], cut.back(), false));
        return out;
    }
    // strip_prefix with an encrypted (padded) pattern: shift left by the pattern's hidden length iff
    // the string starts with it -- the barrel shifter's monotone indicator is sel AND [pat[t] != 0]
    Str strip_prefix(const Str& s, const Str& pat, uint32_t* stripped_bit) {
        const uint32_t sel = starts_with(s, pat);
        *stripped_bit = sel;
        std::vector<uint32_t> lead(s.cap, c.trivial(0));
        for (uint32_t t = 0; t < std::min(s.cap, pat.cap); t++) lead[t] = and_bits(sel, not_bit(char_is_zero(pat.ch[t])));
        return shift_left_by_leading(s, lead);
    }
    // strip_suffix with an encrypted (padded) pattern: cand[o] = "s[o..] is exactly the pattern"
    // (window match incl. the end-of-string test); every character from the first such o on is zeroed
    Str strip_suffix(const Str& s, const Str& pat, uint32_t* stripped_bit) {
        std::vector<uint32_t> cand = window_matches(s, pat, s.cap + 1, false, true);
        std::vector<uint32_t> cut = prefix_or(std::vector<uint32_t>(cand.begin(), cand.begin() + s.cap));
        *stripped_bit = reduce_local({cut.back(), cand[s.cap]}, false);
        Str out;
        out.cap = s.cap;
        out.ch.resize(s.cap);
        for (uint32_t i = 0; i < s.cap; i++)
            for (uint32_t k = 0; k < bpc; k++) out.ch[i].push_back(gate_block(s.ch[i][k], cut[i], false));
        return out;
    }
    // ---- lexicographic order (the reference's comparator idea: per-block sign, then pairwise
    //      selection "most significant non-equal wins", integer/server_key/comparator.rs:226-280) ----
    // sign encoding: 0 = a < b, 1 = equal, 2 = a > b.  Strings compare like big-endian integers of
    // their (zero padded) bytes, which is exactly bytes.__lt__ on the unpadded strings.
    uint32_t compare_sign(const Str& a, const Str* b, const uint8_t* clear, uint32_t clear_len) {
        const uint32_t n = b ? std::max(a.cap, b->cap) : std::max<uint32_t>(a.cap, clear_len);
        std::vector<uint32_t> signs;   // most significant first
        const uint32_t mm = M;
        for (uint32_t i = 0; i < n; i++) {
            bool packed = b && packed_compare && bpc % 2 == 0 && i < a.cap && i < b->cap;
            for (uint32_t k = 0; packed && k + 1 < bpc; k += 2)
                packed = packed_pair_fits(a.ch[i][k], a.ch[i][k + 1], b->ch[i][k], b->ch[i][k + 1]);
            if (packed) {
                // sign of a packed block pair in one PBS: the sign function is odd, hence negacyclic for
                // free -- f(0) = 0, f(1..T-1) = 1 gives -1 on the negative (padding-bit) side; +1 maps
                // {-1, 0, 1} onto the {0, 1, 2} encoding
                const uint32_t l = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
                for (int k = (int)bpc - 2; k >= 0; k -= 2) {
                    const uint32_t d = c.lin({{a.ch[i][k], 1}, {a.ch[i][k + 1], (int32_t)M}, {b->ch[i][k], -1},
                                              {b->ch[i][k + 1], -(int32_t)M}});
                    signs.push_back(c.lin({{c.pbs(d, l, /*signed_input=*/true), 1}}, 1, 2));
                }
                continue;
            }
            for (int k = (int)bpc - 1; k >= 0; k--) {
                const bool has_a = i < a.cap;
                if (b) {
                    const bool has_b = i < b->cap;
                    if (has_a && has_b) {   // sign(a - b) from a - b + (M-1) in [0, 2M-2]
                        const uint32_t l = c.lut_fn([mm](uint64_t x) { return (uint64_t)(x < mm - 1 ? 0 : (x == mm - 1 ? 1 : 2)); });
                        signs.push_back(c.pbs(c.lin({{a.ch[i][k], 1}, {b->ch[i][k], -1}}, (int64_t)M - 1, 2 * ((int64_t)M - 1)), l));
                    } else {               // the missing side is a null char
                        const uint32_t l = c.lut_fn([has_a](uint64_t x) { return (uint64_t)(x == 0 ? 1 : (has_a ? 2 : 0)); });
                        signs.push_back(c.pbs(has_a ? a.ch[i][k] : b->ch[i][k], l));
                    }
                } else {
                    const uint32_t v = clear_block(i < clear_len ? clear[i] : 0, (uint32_t)k);
                    if (!has_a) { if (v) signs.push_back(c.trivial(0)); continue; }   // a is null there: a < clear iff clear != 0
                    const uint32_t l = c.lut_fn([v](uint64_t x) { return (uint64_t)(x < v ? 0 : (x == v ? 1 : 2)); });
                    signs.push_back(c.pbs(a.ch[i][k], l));
                }
            }
        }
        if (signs.empty()) return c.trivial(1);
        const uint32_t pick = c.lut_fn([](uint64_t x) { const uint64_t hi = x / 3, lo = x % 3; return hi != 1 ? (hi > 2 ? (uint64_t)1 : hi) : lo; });
        while (signs.size() > 1) {
            std::vector<uint32_t> next;
            for (size_t i = 0; i + 1 < signs.size(); i += 2)
                next.push_back(c.pbs(c.lin({{signs[i], 3}, {signs[i + 1], 1}}, 0, 8), pick));
            if (signs.size() & 1) next.push_back(signs.back());
            signs.swap(next);
        }
        return signs[0];
    }
    uint32_t order_bit(uint32_t sign, const std::string& op) {
        const bool lt = op == "lt", le = op == "le", gt = op == "gt";
        const uint32_t l = c.lut_fn([lt, le, gt](uint64_t s) {
            return (uint64_t)(lt ? s == 0 : (le ? s != 2 : (gt ? s == 2 : s != 0)));
        });
        return c.pbs(sign, l);
    }
    void emit(const Str& s) {
        for (auto& blocks : s.ch)
            for (uint32_t b : blocks) c.output(b);
    }
};

// ------------------------------------------------------------------------------------------------
// op dispatch used by the C ABI: builds the circuit for `op` and declares its outputs
int build_string_op(Circuit& c, const std::string& op, uint32_t a_cap, uint32_t b_cap,
                    const uint8_t* clear, uint32_t clear_len) {
    StrOps s(c);
    // "name:p1:p2": numeric parameters of the general replace (pattern capacity / length, output capacity)
    std::vector<uint32_t> op_params;
    std::string op_name = op;
    {
        size_t colon = op_name.find(':');
        if (colon != std::string::npos) {
            std::string rest = op_name.substr(colon + 1);
            op_name = op_name.substr(0, colon);
            size_t pos = 0;
            while (pos <= rest.size()) {
                size_t nxt = rest.find(':', pos);
                if (nxt == std::string::npos) nxt = rest.size();
                const std::string tok = rest.substr(pos, nxt - pos);
                if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos || tok.size() > 9)
                    return fail("bad numeric parameter in string op: " + op);
                op_params.push_back((uint32_t)std::stoul(tok));
                pos = nxt + 1;
            }
        }
    }
    const std::string ref_suffix = "_reference";
    for (const char* tail : {"_reference_clear", "_reference"}) {
        const std::string t(tail);
        if (op_name.size() > t.size() && op_name.compare(op_name.size() - t.size(), t.size(), t) == 0) {
            s.packed_compare = false;          // the reference's block-by-block circuit shape
            s.full_box_reduce = false;         // ... and its chunks of T - 1 comparison bits
            op_name = op_name.substr(0, op_name.size() - t.size()) + (t == "_reference_clear" ? "_clear" : "");
            break;
        }
    }
    if (!s.ok) return fail("string ops need msg_mod = 2^b with b | 8 and carry_mod >= msg_mod");
    if (a_cap == 0) return fail("string capacity must be > 0");
    const bool is_clear = op_name.size() > 6 && op_name.compare(op_name.size() - 6, 6, "_clear") == 0;
    const std::string base = is_clear ? op_name.substr(0, op_name.size() - 6) : op_name;
    if (is_clear && !clear && clear_len) return fail("null clear pattern");
    Str a = s.input_string(a_cap);
    Str b;
    const bool unary = base == "to_upper" || base == "to_lower" || base == "trim_start" || base == "trim_end" ||
                       base == "strip" || base == "trim" || base == "len" || base == "is_empty";
    const bool is_replace = base == "replace";
    if (!is_clear && !unary) {
        if (b_cap == 0) return fail("pattern capacity must be > 0");
        b = s.input_string(b_cap);
    }
    uint32_t n_digits = 0;
    while ((1ull << (n_digits * s.bits_per_block)) < (uint64_t)a_cap + 1) n_digits++;
    if (base == "eq" || base == "ne") {
        const bool want = base == "eq";
        c.output(is_clear ? s.eq_clear(a, clear, clear_len, want) : s.eq(a, b, want));
    } else if (base == "starts_with") {
        c.output(is_clear ? s.starts_with_clear(a, clear, clear_len) : s.starts_with(a, b));
    } else if (base == "ends_with") {
        c.output(is_clear ? s.ends_with_clear(a, clear, clear_len) : s.ends_with(a, b));
    } else if (base == "contains") {
        c.output(is_clear ? s.contains_clear(a, clear, clear_len) : s.contains(a, b));
    } else if (base == "lt" || base == "le" || base == "gt" || base == "ge") {
        c.output(s.order_bit(s.compare_sign(a, is_clear ? nullptr : &b, clear, clear_len), base));
    } else if (base == "eq_ignore_case") {
        if (is_clear) {
            std::vector<uint8_t> lower(clear, clear + clear_len);
            for (auto& ch : lower) if (ch >= 'A' && ch <= 'Z') ch += 32;
            c.output(s.eq_clear(s.case_folded(a), lower.data(), clear_len, true));
        } else {
            c.output(s.eq(s.case_folded(a), s.case_folded(b), true));
        }
    } else if (base == "is_empty") {
        c.output(s.is_empty(a));
    } else if (base == "len") {
        std::vector<uint32_t> outs;
        s.len(a, n_digits, outs);
        for (uint32_t o : outs) c.output(o);
    } else if (base == "strip_prefix" || base == "strip_suffix") {
        uint32_t bit = 0;
        Str r = is_clear ? (base == "strip_prefix" ? s.strip_prefix_clear(a, clear, clear_len, &bit)
                                                   : s.strip_suffix_clear(a, clear, clear_len, &bit))
                         : (base == "strip_prefix" ? s.strip_prefix(a, b, &bit) : s.strip_suffix(a, b, &bit));
        c.output(bit);          // first output: 1 iff the pattern was stripped
        s.emit(r);
    } else if (base == "find" || base == "rfind") {
        const bool from_end = base == "rfind";
        std::vector<uint32_t> match;
        if (is_clear) {
            if (clear_len > a_cap) match.assign(1, c.trivial(0));
            else if (clear_len == 0) match.assign(1, c.trivial(1));
            else match = s.window_matches_clear(a, clear, clear_len, a_cap - clear_len + 1);
        } else {
            match = s.window_matches(a, b, a_cap, true);
            if (from_end) {
                // an (all-padding) pattern also "matches" past the end of the string: only offsets
                // o <= len(s) count, i.e. o == 0 or s[o-1] != 0
                const uint32_t both = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 1); });
                for (uint32_t o = 1; o < match.size(); o++)
                    match[o] = c.pbs(c.lin({{match[o], 1}, {s.char_is_zero(a.ch[o - 1]), 2}}, 0, 3), both);
                // ... and an empty pattern matches at offset cap when the string fills its capacity
                // (bytes.rfind(b"") == len): pattern empty AND s[cap-1] != 0
                match.push_back(c.pbs(c.lin({{s.char_is_zero(b.ch[0]), 1}, {s.char_is_zero(a.ch[a_cap - 1]), 2}}, 0, 3), both));
            }
        }
        std::vector<uint32_t> outs;
        if (from_end && is_clear && clear_len == 0) {       // "".rfind in s == len(s)
            outs.push_back(c.trivial(1));
            std::vector<uint32_t> digits;
            s.len(a, n_digits, digits);
            outs.insert(outs.end(), digits.begin(), digits.end());
        } else {
            s.find_from_matches(match, n_digits, outs, from_end);
        }
        for (uint32_t o : outs) c.output(o);
    } else if (base == "concat") {
        // out capacity = a_cap + (b_cap | clear_len); clear right operand: "concat_clear"
        if (is_clear && clear_len == 0) s.emit(a);
        else s.emit(s.concat(a, is_clear ? s.clear_string(clear, clear_len) : b));
    } else if (base == "repeat") {
        // clear repetition count in clear[0] (>= 1): out capacity = count * a_cap
        if (!is_clear || clear_len != 1 || clear[0] == 0) return fail("repeat_clear takes one clear byte: the count (>= 1)");
        Str r = a;
        for (uint32_t i = 1; i < clear[0]; i++) r = s.concat(r, a);
        s.emit(r);
    } else if (base == "to_upper" || base == "to_lower") {
        std::vector<uint32_t> outs;
        s.change_case(a, base == "to_lower", outs);
        for (uint32_t o : outs) c.output(o);
    } else if (base == "trim_end") {
        s.emit(s.trim_end(a));
    } else if (base == "trim_start") {
        s.emit(s.trim_start(a));
    } else if (base == "strip" || base == "trim") {
        s.emit(s.trim_start(s.trim_end(a)));
    } else if (is_replace) {
        // replace_clear[:F:C]  clear = from (F bytes; default: half) || to,   output capacity C (default a_cap)
        // replace[:F:C]        b = encrypted from (capacity F; default: half) || to
        //   without parameters: equal lengths, encrypted operands fill their capacity, rewritten in place;
        //   with parameters: any lengths, encrypted operands may be zero padded (hidden lengths)
        if (!op_params.empty() && op_params.size() != 2) return fail("replace takes two parameters: pattern capacity and output capacity");
        const bool general = !op_params.empty();
        const uint32_t out_cap = general ? op_params[1] : a_cap;
        if (out_cap == 0) return fail("output capacity must be > 0");
        if (is_clear) {
            if (!general && clear_len % 2) return fail("replace_clear expects `from` and `to` of equal length, concatenated");
            const uint32_t m = general ? op_params[0] : clear_len / 2;
            if (m > clear_len) return fail("replace_clear: pattern length beyond the clear buffer");
            const uint32_t t = clear_len - m;
            const uint8_t* to = clear + m;
            if (m == 0) s.emit(general ? s.replace_empty_pattern(a, to, t, out_cap) : a);
            else if (m > a_cap) s.emit(s.fit(a, out_cap));
            else {
                std::vector<uint32_t> match = s.window_matches_clear(a, clear, m, a_cap - m + 1);
                StrOps::Occurrences oc = s.occurrences(match, m, a_cap, StrOps::has_border(clear, m), nullptr);
                s.emit(m == t && out_cap == a_cap ? s.replace_in_place(a, oc, m, to, nullptr)
                                                  : s.replace_general(a, oc, to, t, nullptr, out_cap));
            }
        } else {
            if (!general && b_cap % 2) return fail("replace expects `from` and `to` of equal capacity, concatenated");
            const uint32_t m = general ? op_params[0] : b_cap / 2;
            if (m == 0 || m > b_cap) return fail("replace: pattern capacity must be in 1..b_cap");
            Str from, to;
            from.cap = m;
            to.cap = b_cap - m;
            from.ch.assign(b.ch.begin(), b.ch.begin() + m);
            to.ch.assign(b.ch.begin() + m, b.ch.end());
            if (!general) {
                if (m > a_cap) { s.emit(a); }
                else {
                    std::vector<uint32_t> match = s.window_matches(a, from, a_cap - m + 1, false);
                    StrOps::Occurrences oc = s.occurrences(match, m, a_cap, true, nullptr);
                    s.emit(s.replace_in_place(a, oc, m, nullptr, &to));
                }
            } else {
                // padded pattern: its zero tail matches anything; an occurrence runs over from[j] != 0 only.
                // A pattern that decrypts to the empty string selects nothing (nzf[0] = 0).
                std::vector<uint32_t> nzf;
                for (uint32_t j = 0; j < m; j++) nzf.push_back(s.not_bit(s.char_is_zero(from.ch[j])));
                std::vector<uint32_t> match = s.window_matches(a, from, a_cap, true);
                for (uint32_t o = 0; o < match.size(); o++) {
                    StrOps::Scope sc(c, s.owner_for(o, (uint32_t)match.size()));
                    match[o] = s.and_bits(match[o], nzf[0]);
                }
                StrOps::Occurrences oc = s.occurrences(match, m, a_cap, true, &nzf);
                s.emit(s.replace_general(a, oc, nullptr, 0, to.cap ? &to : nullptr, out_cap));
            }
        }
    } else {
        return fail("unknown string op: " + op);
    }
    if (c.failed()) return fail("circuit build error: " + c.error());
    return 0;
}

}  // namespace fhe
