// fhe_integer.cpp -- radix-integer operations of the reference's integer layer as batched shortint
// circuits: the callers on the other side of the apply_lookup_table boundary that
// the string operations lean on, exposed through the plan ABI (fhe_int_plan_create).
//
// An unsigned radix integer = n_blocks shortint blocks, little endian, log2(msg_mod) bits each
// (integer/block_decomposition.rs:119-144).  Restated from the reference:
//   add / sub / scalar_add / scalar_sub   unchecked add, then the one-carry parallel propagation:
//        generate-or-propagate state per block, Hillis-Steele prefix scan of the states, add the incoming
//        carry, message_extract
//        (integer/server_key/radix_parallel/add.rs:13-42,487-542 propagate_single_carry_parallelized_low_latency,
//         :544-624 compute_prefix_sum_hillis_steele, :724-772 generate_init_carry_array;
//         scalar_add.rs:204-222; subtraction as a + (2^bits - 1 - b) + 1, the two's complement the
//         reference reaches through neg.rs's correcting terms)
//   message_extract / carry_extract       shortint/server_key/mod.rs (x % msg_mod, x / msg_mod)
//   cmux (if_then_else)                   radix_parallel/cmux.rs:194-248: zero out each side by the
//        condition (one lookup per block and side), add, message_extract
//   eq / ne / gt / ge / lt / le (+ scalar_*)   integer/server_key/comparator.rs:193-280: pack two blocks,
//        true LWE subtraction whose sign lands in the padding bit, sign lookup + 1 in {0, 1, 2},
//        pairwise reduction 4*msb + lsb, final sign -> bool lookup; equality through
//        radix_parallel/scalar_comparison.rs:147-233
// Every lookup of one level of one operation is one batched KS+PBS launch (circuit.h).
#include <algorithm>
#include <string>

#include "circuit.h"

namespace fhe {

namespace {

struct RadixOps {
    Circuit& c;
    uint32_t M, T, bits;
    explicit RadixOps(Circuit& c) : c(c), M(c.msg_modulus()), T(c.total_modulus()), bits(0) {
        while ((1u << bits) < M) bits++;
    }
    bool ok() const { return (1u << bits) == M && M >= 2 && T / M >= M; }

    std::vector<uint32_t> input(uint32_t n) {
        std::vector<uint32_t> v;
        for (uint32_t i = 0; i < n; i++) v.push_back(c.input(M - 1));
        return v;
    }
    std::vector<uint32_t> digits(uint64_t x, uint32_t n) const {
        std::vector<uint32_t> d;
        for (uint32_t i = 0; i < n; i++) { d.push_back((uint32_t)(x & (M - 1))); x >>= bits; }
        return d;
    }
    uint32_t message_extract(uint32_t b) { const uint32_t m = M; return c.pbs(b, c.lut_fn([m](uint64_t x) { return x % m; })); }
    uint32_t carry_extract(uint32_t b) { const uint32_t m = M; return c.pbs(b, c.lut_fn([m](uint64_t x) { return x / m; })); }

    // propagate_single_carry_parallelized_low_latency (add.rs:518-542): sums[i] in [0, 2M-1]
    std::vector<uint32_t> propagate(const std::vector<uint32_t>& sums) {
        const uint32_t n = (uint32_t)sums.size(), m = M;
        // generate_init_carry_array (add.rs:724-772): 0 = None, 1 = Generated, 2 = Propagated
        const uint32_t first = c.lut_fn([m](uint64_t x) { return (uint64_t)(x >= m ? 1 : 0); });
        const uint32_t other = c.lut_fn([m](uint64_t x) { return (uint64_t)(x >= m ? 1 : (x == m - 1 ? 2 : 0)); });
        std::vector<uint32_t> st(n);
        for (uint32_t i = 0; i < n; i++) st[i] = c.pbs(sums[i], i == 0 ? first : other);
        // compute_prefix_sum_hillis_steele (add.rs:572-624) with prefix_sum_carry_propagation (add.rs:36-42)
        const uint32_t comb = c.lut_fn([m](uint64_t x) { const uint64_t msb = (x / m) % m, lsb = x % m; return msb == 2 ? lsb : msb; });
        for (uint32_t space = 1; space < n; space *= 2) {
            std::vector<uint32_t> next(st);
            for (uint32_t i = space; i < n; i++) next[i] = c.pbs(c.lin({{st[i], (int32_t)M}, {st[i - space], 1}}), comb);
            st.swap(next);
        }
        // the output carry of block i-1 is the input carry of block i; add, then message_extract
        std::vector<uint32_t> out(n);
        for (uint32_t i = 0; i < n; i++)
            out[i] = message_extract(i == 0 ? sums[0] : c.lin({{sums[i], 1}, {st[i - 1], 1}}));
        return out;
    }
    std::vector<uint32_t> add(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        std::vector<uint32_t> s;
        for (size_t i = 0; i < a.size(); i++) s.push_back(c.lin({{a[i], 1}, {b[i], 1}}));   // unchecked_add (add.rs:88-100)
        return propagate(s);
    }
    std::vector<uint32_t> sub(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        std::vector<uint32_t> s;   // a + (2^bits - 1 - b) + 1
        for (size_t i = 0; i < a.size(); i++) s.push_back(c.lin({{a[i], 1}, {b[i], -1}}, (int64_t)M - 1 + (i == 0 ? 1 : 0)));
        return propagate(s);
    }
    std::vector<uint32_t> scalar_add(const std::vector<uint32_t>& a, uint64_t scalar) {
        const auto d = digits(scalar, (uint32_t)a.size());
        std::vector<uint32_t> s;
        for (size_t i = 0; i < a.size(); i++) s.push_back(c.lin({{a[i], 1}}, d[i]));       // unchecked_scalar_add (scalar_add.rs)
        return propagate(s);
    }
    std::vector<uint32_t> scalar_sub(const std::vector<uint32_t>& a, uint64_t scalar) {
        const uint32_t total_bits = bits * (uint32_t)a.size();
        const uint64_t mask = total_bits >= 64 ? ~0ull : ((1ull << total_bits) - 1);
        return scalar_add(a, (0 - scalar) & mask);
    }
    // zero_out_if + add + message_extract (cmux.rs:194-316)
    std::vector<uint32_t> cmux(uint32_t cond, const std::vector<uint32_t>& t, const std::vector<uint32_t>& f) {
        const uint32_t m2 = 2 * M;
        const uint32_t keep_set = c.lut_fn([m2](uint64_t x) { return (uint64_t)((x < m2 && (x & 1)) ? x >> 1 : 0); });
        const uint32_t keep_clear = c.lut_fn([m2](uint64_t x) { return (uint64_t)((x < m2 && !(x & 1)) ? x >> 1 : 0); });
        std::vector<uint32_t> out;
        for (size_t i = 0; i < t.size(); i++) {
            const uint32_t a = c.pbs(c.lin({{cond, 1}, {t[i], 2}}), keep_set);
            const uint32_t b = c.pbs(c.lin({{cond, 1}, {f[i], 2}}), keep_clear);
            out.push_back(message_extract(c.lin({{a, 1}, {b, 1}}, 0, M - 1)));
        }
        return out;
    }
    // are_all_comparisons_block_true / is_at_least_one (scalar_comparison.rs:147-233)
    uint32_t reduce(std::vector<uint32_t> bits_, bool all) {
        if (bits_.empty()) return c.trivial(all ? 1 : 0);
        const uint32_t nz = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });
        while (bits_.size() > 1) {
            std::vector<uint32_t> next;
            for (size_t i = 0; i < bits_.size(); i += T - 1) {
                const size_t len = std::min<size_t>(T - 1, bits_.size() - i);
                std::vector<Term> terms;
                for (size_t j = 0; j < len; j++) terms.push_back({bits_[i + j], 1});
                next.push_back(c.pbs(c.lin(terms), all ? c.lut_fn([len](uint64_t x) { return (uint64_t)(x == len); }) : nz));
            }
            bits_.swap(next);
        }
        return bits_[0];
    }
    // packed pairs hi*M + lo as LIN nodes (pack_block_chunk, scalar_comparison.rs:104-138); a trailing
    // single block stays alone.  Encrypted operand or clear digits.
    std::vector<uint32_t> packed(const std::vector<uint32_t>& a) {
        std::vector<uint32_t> p;
        for (size_t i = 0; i < a.size(); i += 2)
            p.push_back(i + 1 < a.size() ? c.lin({{a[i], 1}, {a[i + 1], (int32_t)M}}) : a[i]);
        return p;
    }
    std::vector<uint32_t> packed_clear(const std::vector<uint32_t>& d) const {
        std::vector<uint32_t> p;
        for (size_t i = 0; i < d.size(); i += 2) p.push_back(i + 1 < d.size() ? d[i] + d[i + 1] * M : d[i]);
        return p;
    }
    uint32_t eq(const std::vector<uint32_t>& a, const std::vector<uint32_t>* b, uint64_t scalar, bool want_equal) {
        const auto pa = packed(a);
        const uint32_t z = c.lut_fn([](uint64_t x) { return (uint64_t)(x == 0); });
        std::vector<uint32_t> bits_;
        if (b) {
            const auto pb = packed(*b);   // true subtraction: zero iff equal; the sign may reach the padding bit
            for (size_t i = 0; i < pa.size(); i++) bits_.push_back(c.pbs(c.lin({{pa[i], 1}, {pb[i], -1}}), z, true));
        } else {
            const auto pd = packed_clear(digits(scalar, (uint32_t)a.size()));
            for (size_t i = 0; i < pa.size(); i++) {
                const uint32_t v = pd[i];
                bits_.push_back(c.pbs(pa[i], c.lut_fn([v](uint64_t x) { return (uint64_t)(x == v); })));
            }
        }
        const uint32_t all = reduce(bits_, true);
        return want_equal ? all : c.lin({{all, -1}}, 1, 1);
    }
    // comparator.rs:193-280: sign of (a - b) per packed pair, most significant pair wins
    uint32_t sign(const std::vector<uint32_t>& a, const std::vector<uint32_t>* b, uint64_t scalar) {
        const auto pa = packed(a);
        const uint32_t sgn = c.lut_fn([](uint64_t x) { return (uint64_t)(x != 0); });   // odd function: -1 below zero for free
        std::vector<uint32_t> signs;   // least significant first, values {0: <, 1: ==, 2: >}
        if (b) {
            const auto pb = packed(*b);
            for (size_t i = 0; i < pa.size(); i++)
                signs.push_back(c.lin({{c.pbs(c.lin({{pa[i], 1}, {pb[i], -1}}), sgn, true), 1}}, 1, 2));
        } else {
            const auto pd = packed_clear(digits(scalar, (uint32_t)a.size()));
            for (size_t i = 0; i < pa.size(); i++)   // scalar_compare_block_assign (comparator.rs:240-255)
                signs.push_back(c.lin({{c.pbs(c.lin({{pa[i], 1}}, -(int64_t)pd[i]), sgn, true), 1}}, 1, 2));
        }
        // reduce_two_sign_blocks_assign (comparator.rs:257-268): 4 * msb + lsb
        const uint32_t pick = c.lut_fn([](uint64_t x) { const uint64_t msb = (x / 4) & 3, lsb = x & 3; return msb == 1 ? lsb : msb; });
        while (signs.size() > 1) {
            std::vector<uint32_t> next;
            for (size_t i = 0; i + 1 < signs.size(); i += 2) next.push_back(c.pbs(c.lin({{signs[i + 1], 4}, {signs[i], 1}}), pick));
            if (signs.size() & 1) next.push_back(signs.back());
            signs.swap(next);
        }
        return signs[0];
    }
};

}  // namespace

// op in {add, sub, scalar_add, scalar_sub, message_extract, carry_extract, cmux, eq, ne, gt, ge, lt, le,
// scalar_eq, scalar_ne, scalar_gt, scalar_ge, scalar_lt, scalar_le}.  Inputs, in order: [cond (cmux only)],
// a (n_blocks), [b (n_blocks) for the two-operand forms].  Outputs: n_blocks blocks, or one 0/1 block.
int build_integer_op(Circuit& c, const std::string& op, uint32_t n_blocks, uint64_t scalar) {
    RadixOps r(c);
    if (!r.ok()) return fail("integer ops need msg_mod = 2^b >= 2 and carry_mod >= msg_mod");
    if (n_blocks == 0) return fail("n_blocks must be > 0");
    if ((uint64_t)n_blocks * r.bits > 64 && op.compare(0, 7, "scalar_") == 0) return fail("scalar operands are limited to 64 bits");
    auto emit = [&](const std::vector<uint32_t>& v) { for (uint32_t b : v) c.output(b); };
    const bool is_scalar = op.compare(0, 7, "scalar_") == 0;
    const std::string base = is_scalar ? op.substr(7) : op;
    if (op == "cmux") {
        const uint32_t cond = c.input(1);
        const auto t = r.input(n_blocks), f = r.input(n_blocks);
        emit(r.cmux(cond, t, f));
    } else if (op == "message_extract" || op == "carry_extract") {
        std::vector<uint32_t> v;
        for (uint32_t i = 0; i < n_blocks; i++) v.push_back(c.input(r.T - 1));   // blocks with full carries
        for (uint32_t b : v) c.output(op == "message_extract" ? r.message_extract(b) : r.carry_extract(b));
    } else if (base == "add" || base == "sub") {
        const auto a = r.input(n_blocks);
        if (is_scalar) emit(base == "add" ? r.scalar_add(a, scalar) : r.scalar_sub(a, scalar));
        else { const auto b = r.input(n_blocks); emit(base == "add" ? r.add(a, b) : r.sub(a, b)); }
    } else if (base == "eq" || base == "ne") {
        const auto a = r.input(n_blocks);
        if (is_scalar) c.output(r.eq(a, nullptr, scalar, base == "eq"));
        else { const auto b = r.input(n_blocks); c.output(r.eq(a, &b, 0, base == "eq")); }
    } else if (base == "gt" || base == "ge" || base == "lt" || base == "le") {
        const auto a = r.input(n_blocks);
        uint32_t s;
        if (is_scalar) s = r.sign(a, nullptr, scalar);
        else { const auto b = r.input(n_blocks); s = r.sign(a, &b, 0); }
        const bool lt = base == "lt", le = base == "le", gt = base == "gt";
        c.output(c.pbs(s, c.lut_fn([lt, le, gt](uint64_t x) { return (uint64_t)(lt ? x == 0 : (le ? x != 2 : (gt ? x == 2 : x != 0))); })));
    } else {
        return fail("unknown integer op: " + op);
    }
    if (c.failed()) return fail("circuit build error: " + c.error());
    return 0;
}

}  // namespace fhe
