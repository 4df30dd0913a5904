"""FheString operations on the GPU vs clear-text Python string semantics (the reference's notion of
correctness: decrypt(op(enc(x))) == clear_op(x), integer/.../tests_cases_comparisons.rs:33-39,
docs/tutorials/ascii_fhe_string.md:141-152), plus plan-level parity with the CPU oracle."""
import numpy as np
import pytest

import oracle as O
from conftest import gpu_engine, keyset
from plan_oracle import run_with_oracle

pytestmark = pytest.mark.gpu


def _enc(ks, s: bytes, cap: int):
    import fhestr
    blocks = fhestr.string_to_blocks(gpu_engine(ks).params, s, cap)
    return ks.ck.encrypt_many(blocks)


def _dec(ks, cts):
    return ks.ck.decrypt_many(np.asarray(cts).reshape(-1, ks.params.big_size))


def _ops(ks):
    import fhestr
    return fhestr.FheStringOps(gpu_engine(ks))


CASES = [(b"hello", b"hello"), (b"hello", b"hellp"), (b"hello", b"hell"), (b"", b""), (b"", b"a"),
         (b"abcdefgh", b"abcdefgh"), (b"Zz", b"zZ")]


@pytest.mark.parametrize("a,b", CASES)
def test_eq_ne_encrypted_and_clear(toy_k1, a, b):
    ops = _ops(toy_k1)
    ea, eb = _enc(toy_k1, a, 8), _enc(toy_k1, b, 8)
    assert _dec(toy_k1, ops.eq(ea, eb))[0] == int(a == b)
    assert _dec(toy_k1, ops.ne(ea, eb))[0] == int(a != b)
    assert _dec(toy_k1, ops.eq(ea, b))[0] == int(a == b)
    assert _dec(toy_k1, ops.ne(ea, b))[0] == int(a != b)


def test_eq_different_capacities(toy_k1):
    ops = _ops(toy_k1)
    assert _dec(toy_k1, ops.eq(_enc(toy_k1, b"abc", 8), _enc(toy_k1, b"abc", 4)))[0] == 1
    assert _dec(toy_k1, ops.eq(_enc(toy_k1, b"abcde", 8), _enc(toy_k1, b"abc", 4)))[0] == 0
    assert _dec(toy_k1, ops.eq(_enc(toy_k1, b"abc", 4), b"abcdefgh"))[0] == 0  # clear longer than capacity


PATTERN_CASES = [(b"hello wd", b"hell"), (b"hello wd", b"o wd"), (b"hello wd", b"lo"), (b"hello wd", b"xyz"),
                 (b"hello", b"lo"), (b"hello", b""), (b"", b""), (b"", b"a"), (b"aaa", b"aa"), (b"abab", b"bab"),
                 (b"ab", b"abc"), (b"hello", b"hello"), (b"hellohel", b"hel")]


@pytest.mark.parametrize("s,pat", PATTERN_CASES)
def test_starts_ends_contains(toy_k1, s, pat):
    ops = _ops(toy_k1)
    es, ep = _enc(toy_k1, s, 8), _enc(toy_k1, pat, 4 if len(pat) <= 4 else 8)
    for name, want in (("starts_with", s.startswith(pat)), ("ends_with", s.endswith(pat)), ("contains", pat in s)):
        assert _dec(toy_k1, getattr(ops, name)(es, ep))[0] == int(want), (name, "encrypted pattern")
        assert _dec(toy_k1, getattr(ops, name)(es, pat))[0] == int(want), (name, "clear pattern")


@pytest.mark.parametrize("s,pat", PATTERN_CASES)
def test_find(toy_k1, s, pat):
    ops = _ops(toy_k1)
    es, ep = _enc(toy_k1, s, 8), _enc(toy_k1, pat, 4 if len(pat) <= 4 else 8)
    want = s.find(pat)
    for p in (ep, pat):
        out = _dec(toy_k1, ops.find(es, p))
        found, digits = out[0], out[1:]
        assert found == int(want >= 0)
        if want >= 0:
            assert sum(int(d) * 4**i for i, d in enumerate(digits)) == want


@pytest.mark.parametrize("s", [b"Hello Wd", b"az AZ@[`", b"", b"{}09\x7f\x01"])
def test_case_conversion(toy_k1, s):
    import fhestr
    ops = _ops(toy_k1)
    es = _enc(toy_k1, s, 8)
    P = gpu_engine(toy_k1).params
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.to_upper(es))) == s.upper().rstrip(b"\0")
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.to_lower(es))) == s.lower().rstrip(b"\0")


@pytest.mark.parametrize("s", [b"  hi  ", b"\t a b\n", b"abc", b"    ", b""])
def test_trim_ops(toy_k1, s):
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    es = _enc(toy_k1, s, 8)
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.trim_end(es))) == s.rstrip()
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.trim_start(es))) == s.lstrip()
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.strip(es))) == s.strip()


@pytest.mark.parametrize("s,frm,to", [(b"abcabc", b"bc", b"XY"), (b"aaaa", b"aa", b"bc"), (b"aaa", b"aa", b"xy"),
                                       (b"abababab", b"aba", b"xyz")])
def test_replace(toy_k1, s, frm, to):
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    es = _enc(toy_k1, s, 8)
    want = s.replace(frm, to)
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.replace(es, frm, to))) == want
    enc_out = ops.replace(es, _enc(toy_k1, frm, len(frm)), _enc(toy_k1, to, len(to)))
    assert fhestr.blocks_to_string(P, _dec(toy_k1, enc_out)) == want


def test_p22_replace_to_lower_strip_64_chars(p22):
    """The config-5 operations on PARAM_MESSAGE_2_CARRY_2 (4 blocks per char); config 5 itself --
    1024 chars on the reference's PARAM_MESSAGE_4_CARRY_4, N = 32768 -- is tests/test_gpu_config5.py."""
    import fhestr
    ops = _ops(p22)
    P = gpu_engine(p22).params
    s = b"  The Quick brown FOX jumps over the lazy dog; the end.   "
    es = _enc(p22, s, 64)
    assert fhestr.blocks_to_string(P, _dec(p22, ops.to_lower(es))) == s.lower()
    assert fhestr.blocks_to_string(P, _dec(p22, ops.replace(es, b"the ", b"THAT"))) == s.replace(b"the ", b"THAT")
    assert fhestr.blocks_to_string(P, _dec(p22, ops.strip(es))) == s.strip()


@pytest.mark.parametrize("s,frm,to", [
    ((b"ab" * 33)[:64], b"aba", b"XYZ"),                                        # self-overlapping, borders of length 1
    ((b"ab" * 32)[:63] + b"b", b"abab", b"WXYZ"),                               # borders of length 2, broken tail
    (b"a" * 64, b"aaa", b"123"),                                                # every offset matches
    (b"x" * 20 + b"the " + b"y" * 17 + b"the the " + b"z" * 15, b"the ", b"THAT"),   # isolated and adjacent occurrences
    (b"q" * 61 + b"abc", b"abc", b"XYZ"),                                       # only the last offset
])
def test_p22_replace_encrypted_pattern_64_chars_blocked_scan(p22, s, frm, to):
    """replace with an ENCRYPTED pattern and replacement on PARAM_MESSAGE_2_CARRY_2, 64 characters: the
    leftmost non-overlapping occurrences come from the blocked scan (fhe_string.cpp: occurrences_scan; two offsets per
    chain step for these unpadded 3- and 4-character patterns), against bytes.replace; then the padded (hidden-length) form."""
    import fhestr
    ops = _ops(p22)
    P = gpu_engine(p22).params
    es = _enc(p22, s, 64)
    want = s.replace(frm, to)
    out = ops.replace(es, _enc(p22, frm, len(frm)), _enc(p22, to, len(to)))
    assert fhestr.blocks_to_string(P, _dec(p22, out)) == want
    padded = ops.replace(es, _enc(p22, frm, 4), _enc(p22, to, 4), out_cap=64)
    assert fhestr.blocks_to_string(P, _dec(p22, padded)) == want


def test_replace_blocked_scan_fuzz(toy_k1):
    """Random strings over {a, b} (overlapping candidates everywhere), 50 .. 72 characters, encrypted patterns of 2 .. 4
    characters: unpadded in place (two offsets per scan step where 4 m <= 16) and zero padded with hidden lengths, against
    bytes.replace.  60 cases on the toy parameter set."""
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    rng = np.random.default_rng(0x5CA9)
    for case in range(60):
        cap = int(rng.integers(50, 73))
        n = int(rng.integers(cap - 8, cap + 1))
        s = bytes(rng.choice(np.frombuffer(b"ab", dtype=np.uint8), size=n))
        m = int(rng.integers(2, 5))
        frm = bytes(rng.choice(np.frombuffer(b"ab", dtype=np.uint8), size=m))
        to = bytes(rng.choice(np.frombuffer(b"XYZ", dtype=np.uint8), size=m))
        want = s.replace(frm, to)
        es = _enc(toy_k1, s, cap)
        if case % 2 == 0:
            out = ops.replace(es, _enc(toy_k1, frm, m), _enc(toy_k1, to, m))
        else:
            out = ops.replace(es, _enc(toy_k1, frm, 4), _enc(toy_k1, to, 4), out_cap=cap)
        assert fhestr.blocks_to_string(P, _dec(toy_k1, out)) == want, (case, s, frm, to)


def test_plan_matches_oracle_execution(toy_k1):
    """Same plan, same inputs: GPU executor vs the oracle stepping through the exported levels."""
    import fhestr
    eng = gpu_engine(toy_k1)
    plan = fhestr.Plan.string_op(eng, "contains", 8, 4)
    inputs = np.concatenate([_enc(toy_k1, b"abcabd", 8), _enc(toy_k1, b"abd", 4)])
    got = plan.run(inputs)
    want = run_with_oracle(plan, inputs, toy_k1.sk)
    assert np.array_equal(_dec(toy_k1, got), _dec(toy_k1, want))
    assert _dec(toy_k1, got)[0] == 1


def test_plan_pbs_counts_match_survey():
    """eq enc-enc on 256 chars = 1099 KS+PBS in 4 levels (reference shape); the default plan: 547."""
    import fhestr
    ks = keyset(O.TOY_K1)
    eng = gpu_engine(ks)
    info = fhestr.Plan.string_op(eng, "eq_reference", 256, 256).info()
    assert (info["n_pbs"], info["n_levels"]) == (1099, 4)
    info = fhestr.Plan.string_op(eng, "eq_clear", 256, 0, b"x" * 200).info()
    assert (info["n_pbs"], info["n_levels"]) == (547, 4)


def test_p22_eq_256_chars(p22):
    """BASELINE.json config 3: 256-char padded strings, equal / differing at one position."""
    ops = _ops(p22)
    rng = np.random.default_rng(0x5EED0003)
    a = bytes(rng.integers(0x20, 0x7F, size=200, dtype=np.uint8))
    pos = int(rng.integers(0, 200))
    b = a[:pos] + bytes([a[pos] ^ 1]) + a[pos + 1:]
    ea, eb = _enc(p22, a, 256), _enc(p22, b, 256)
    assert _dec(p22, ops.eq(ea, _enc(p22, a, 256)))[0] == 1
    assert _dec(p22, ops.eq(ea, eb))[0] == 0
    assert _dec(p22, ops.ne(ea, eb))[0] == 1
    assert _dec(p22, ops.eq(ea, a))[0] == 1
    assert _dec(p22, ops.eq(ea, b))[0] == 0


def test_p22_contains_find_16_in_256(p22):
    """BASELINE.json config 4 (single GPU leg): 16-char encrypted pattern in a 256-char haystack."""
    ops = _ops(p22)
    rng = np.random.default_rng(0x5EED0004)
    hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
    off = int(rng.integers(0, 240))
    pat = hay[off: off + 16]
    eh = _enc(p22, hay, 256)
    out = _dec(p22, ops.find(eh, _enc(p22, pat, 16)))
    assert out[0] == 1 and sum(int(d) * 4**i for i, d in enumerate(out[1:])) == hay.find(pat)
    absent = b"0123456789ABCDEF"
    assert _dec(p22, ops.contains(eh, _enc(p22, absent, 16)))[0] == 0
    assert _dec(p22, ops.contains(eh, _enc(p22, pat, 16)))[0] == 1


def test_sharded_runner_gpu_backend_single_rank(toy_k1):
    """The multi-GPU executor's product backend on one GPU (world 1) incl. a forced all-gather
    through a 1-rank RCCL group: same result as fhe_plan_run."""
    import os
    import torch
    import torch.distributed as dist
    import fhestr
    from fhestr.distributed import GpuBackend, ShardedPlanRunner
    eng = gpu_engine(toy_k1)
    plan = fhestr.Plan.string_op(eng, "find", 8, 4)
    inputs = np.concatenate([_enc(toy_k1, b"xxabdabd", 8), _enc(toy_k1, b"abd", 4)])
    want = _dec(toy_k1, plan.run(inputs))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")   # make sure the CUDA/HIP context exists before the RCCL group
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        backend = GpuBackend(plan, torch.device("cuda", 0))
        runner = ShardedPlanRunner(plan, 0, 1, backend)
        host_out = runner.run(inputs)
        got = _dec(toy_k1, host_out)
        assert np.array_equal(got, want) and got[0] == 1 and got[1] + 4 * got[2] == 2
        # inputs and outputs resident in HBM: the same words as the host round trip
        d_in = torch.from_numpy(inputs.view(np.int64)).cuda()
        d_out = runner.run(d_in, device_outputs=True)
        assert isinstance(d_out, torch.Tensor) and d_out.is_cuda
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), host_out)
        # exercise the collective itself on a level region
        pool = backend.alloc_pool(plan.info()["pool_slots"])
        lv = plan.level_info(0)
        backend.all_gather(pool, lv["local_base"], 1, lv["local_base"] + lv["local_size"], 1)
        torch.cuda.synchronize()
    finally:
        eng.set_stream(None)
        if created:
            dist.destroy_process_group()


class _LoopbackRanks:
    """All ranks of a sharded plan on ONE GPU: every simulated rank owns a pool in HBM and runs its share of each
    level through the product kernels (fhe_plan_run_level_rank_dev); the all-gather is replaced by the copies it
    stands for (rank q's exported region -> slot q of every rank's receive region)."""

    def __init__(self, plan, world):
        import torch
        from fhestr.distributed import GpuBackend
        self.torch, self.plan, self.world = torch, plan, world
        self.backend = GpuBackend(plan, torch.device("cuda", 0), reuse_pool=False)
        self.info = plan.info()
        self.levels = [plan.level_info(l) for l in range(self.info["n_levels"])]

    def run(self, inputs):
        b = self.backend
        pools = [b.alloc_pool(self.info["pool_slots"]) for _ in range(self.world)]
        for pool in pools:
            b.load_inputs(pool, inputs, self.info["n_inputs"])
        for l, lv in enumerate(self.levels):
            for r, pool in enumerate(pools):
                b.run_level(pool, l, r)
            e = lv["e_max"]
            if e:
                for pool in pools:
                    for q, src in enumerate(pools):
                        pool[lv["recv_base"] + q * e: lv["recv_base"] + (q + 1) * e].copy_(src[lv["local_base"]: lv["local_base"] + e])
        return [b.gather_outputs(pool, self.info["n_outputs"]) for pool in pools]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_plan_all_ranks_on_one_gpu(p22, world):
    """SURVEY 8(e) on the product kernels: eq (256 chars) and contains (16 in 256) built for `world` ranks, every
    rank's level shares run on this GPU with rank-local pools; each rank ends with the right answer and only the
    exported ciphertexts crossed."""
    import fhestr
    eng = _ops(p22).engine
    rng = np.random.default_rng(0x5EED0004)
    hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
    off = int(rng.integers(0, 240))
    try:
        for op, b_cap, second, want in (("eq", 256, hay, 1), ("eq", 256, hay[:100] + b"#" + hay[101:], 0),
                                        ("contains", 16, hay[off: off + 16], 1), ("contains", 16, b"0123456789ABCDEF", 0)):
            plan = fhestr.Plan.string_op(eng, op, 256, b_cap, world=world)
            inputs = np.concatenate([_enc(p22, hay, 256), _enc(p22, second, b_cap)])
            ranks = _LoopbackRanks(plan, world)
            outs = ranks.run(inputs)
            assert [int(_dec(p22, o)[0]) for o in outs] == [want] * world, (op, world)
            crossing = sum(lv["e_max"] * world for lv in ranks.levels)
            assert crossing < plan.info()["n_pbs"] // 8      # only reduced ciphertexts cross, not the level outputs
            plan.close()
    finally:
        eng.set_stream(None)


def test_string_ops_with_4_bit_blocks_large_n():
    """BASELINE.json config 5 geometry: msg_mod = carry_mod = 16 (one char = 2 blocks) on the
    N = 32768 polynomial size of PARAM_MESSAGE_4_CARRY_4 (toy LWE dimension so keys are quick)."""
    import fhestr
    ks = keyset(O.TOY_N32768)
    eng = gpu_engine(ks)
    ops = fhestr.FheStringOps(eng)
    assert ops.bpc == 2
    s = b" Hello, World "
    es = ks.ck.encrypt_many(fhestr.string_to_blocks(eng.params, s, 16))
    dec = lambda ct: fhestr.blocks_to_string(eng.params, ks.ck.decrypt_many(np.asarray(ct).reshape(-1, ks.params.big_size)))
    assert dec(ops.to_lower(es)) == s.lower()
    assert dec(ops.to_upper(es)) == s.upper()
    assert dec(ops.replace(es, b"l", b"L")) == s.replace(b"l", b"L")
    assert dec(ops.strip(es)) == s.strip()
    assert ks.ck.decrypt_many(ops.contains(es, b"World")[None, :])[0] == 1
    assert ks.ck.decrypt_many(ops.eq(es, es)[None, :])[0] == 1


def test_clear_patterns_by_classes_on_4_bit_blocks():
    """Clear patterns on 4-bit blocks go through the class-doubling matcher (fhe_string.cpp,
    window_matches_clear_classes): lengths that are and are not powers of two, repeated characters and
    self-overlapping patterns, present / absent / at both ends; contains, find, replace and strip against Python."""
    import fhestr
    ks = keyset(O.TOY_N32768)
    eng = gpu_engine(ks)
    ops = fhestr.FheStringOps(eng)
    P = eng.params
    dec = lambda ct: ks.ck.decrypt_many(np.asarray(ct).reshape(-1, ks.params.big_size))
    dec_str = lambda ct: fhestr.blocks_to_string(P, dec(ct))
    hay = b"the cat on the mat ate the abcabcab"
    es = ks.ck.encrypt_many(fhestr.string_to_blocks(P, hay, 40))
    for pat in (b"th", b"the ", b"abcab", b"aa", b"cab", b"the cat on the m", b"zz", b"tab", b"b", b"e the ab"):
        found = dec(ops.find(es, pat))
        want = hay.find(pat)
        assert int(found[0]) == (want >= 0), pat
        if want >= 0:
            assert sum(int(d) * P.msg_mod ** i for i, d in enumerate(found[1:])) == want, pat
        assert int(dec(ops.contains(es, pat))[0]) == (pat in hay), pat
    # the plan really is the classed one: (1 + log2 len) PBS per position, not one per distinct character
    n_pbs = fhestr.Plan.string_op(eng, "contains_clear", 40, 0, clear=b"the ").info()["n_pbs"]
    assert n_pbs < 3 * 40 + 8
    for frm, to in ((b"the ", b"THAT"), (b"ab", b"xy"), (b"abc", b"ABC"), (b"at", b"AT")):
        assert dec_str(ops.replace(es, frm, to)) == hay.replace(frm, to), (frm, to)
    assert dec_str(ops.replace(es, b"the ", b"a ", out_cap=40)) == hay.replace(b"the ", b"a ")
    bit, rest = ops.strip_prefix(es, b"the cat")
    assert int(dec(bit)[0]) == 1 and dec_str(rest) == hay[len(b"the cat"):]
    bit, rest = ops.strip_suffix(es, b"abcab")
    assert int(dec(bit)[0]) == 1 and dec_str(rest) == hay[:-5]


def test_len_rfind_ignore_case_strip_affix(toy_k1):
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    s = b"abcabc"
    es = _enc(toy_k1, s, 8)
    val = lambda digits: sum(int(d) * 4**i for i, d in enumerate(digits))
    assert val(_dec(toy_k1, ops.len(es))) == 6
    assert _dec(toy_k1, ops.is_empty(es))[0] == 0 and _dec(toy_k1, ops.is_empty(_enc(toy_k1, b"", 8)))[0] == 1
    out = _dec(toy_k1, ops.rfind(es, _enc(toy_k1, b"bc", 4)))
    assert out[0] == 1 and val(out[1:]) == s.rfind(b"bc")
    out = _dec(toy_k1, ops.rfind(es, b"zz"))
    assert out[0] == 0
    assert _dec(toy_k1, ops.eq_ignore_case(es, _enc(toy_k1, b"ABCabc", 8)))[0] == 1
    assert _dec(toy_k1, ops.eq_ignore_case(es, b"ABCABD"))[0] == 0
    bit, rest = ops.strip_prefix(es, b"abc")
    assert _dec(toy_k1, bit)[0] == 1 and fhestr.blocks_to_string(P, _dec(toy_k1, rest)) == b"abc"
    bit, rest = ops.strip_suffix(es, b"bc")
    assert _dec(toy_k1, bit)[0] == 1 and fhestr.blocks_to_string(P, _dec(toy_k1, rest)) == b"abca"
    bit, rest = ops.strip_suffix(es, b"xx")
    assert _dec(toy_k1, bit)[0] == 0 and fhestr.blocks_to_string(P, _dec(toy_k1, rest)) == s


@pytest.mark.parametrize("a,b", [(b"abc", b"abd"), (b"abc", b"abc"), (b"abc", b"ab"), (b"", b"a"), (b"zz", b"za")])
def test_lexicographic_order(toy_k1, a, b):
    ops = _ops(toy_k1)
    ea, eb = _enc(toy_k1, a, 8), _enc(toy_k1, b, 8)
    for name, want in (("lt", a < b), ("le", a <= b), ("gt", a > b), ("ge", a >= b)):
        assert _dec(toy_k1, getattr(ops, name)(ea, eb))[0] == int(want)
        assert _dec(toy_k1, getattr(ops, name)(ea, b))[0] == int(want)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_circuits_gpu_vs_clear_and_oracle(toy_k1, seed):
    """Random lin/pbs DAGs through the GPU plan executor: decrypt == clear evaluation == oracle run."""
    import fhestr
    from random_circuits import build_random_circuit
    rng = np.random.default_rng(seed)
    plan = fhestr.Plan(gpu_engine(toy_k1))
    evaluate = build_random_circuit(plan, rng, n_ops=60)
    plan.finalize(1)
    values = [int(v) for v in rng.integers(0, 4, size=6)]
    inputs = toy_k1.ck.encrypt_many(values)
    got = _dec(toy_k1, plan.run(inputs)).tolist()
    assert got == evaluate(values)
    assert got == _dec(toy_k1, run_with_oracle(plan, inputs, toy_k1.sk)).tolist()


def test_rfind_encrypted_empty_pattern_on_full_string(toy_k1):
    """bytes.rfind(b"") == len(s), also when s fills its capacity and the pattern is encrypted (found by
    scripts/fuzz_strings.py)."""
    ops = _ops(toy_k1)
    for s, cap in ((b"qy", 2), (b"BABx", 4), (b"ab", 4), (b"", 3)):
        r = _dec(toy_k1, ops.rfind(_enc(toy_k1, s, cap), _enc(toy_k1, b"", 2)))
        assert r[0] == 1 and sum(int(v) * 4 ** i for i, v in enumerate(r[1:])) == len(s)


def test_fuzz_string_ops_against_python():
    """scripts/fuzz_strings.py: random strings / patterns / operations vs Python bytes semantics."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_strings.py"), "400", "7"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


GENERAL_REPLACE = [(b"abcabc", b"bc", b"X"), (b"abcabc", b"b", b"XYZ"), (b"aaaa", b"aa", b"b"), (b"aaa", b"aa", b"xyz"),
                   (b"hello", b"l", b""), (b"hello", b"zz", b"y"), (b"", b"a", b"bc"), (b"abcabc", b"abc", b"abcd")]


@pytest.mark.parametrize("s,frm,to", GENERAL_REPLACE)
def test_general_replace_gpu(toy_k1, s, frm, to):
    """replace with |from| != |to| through the C ABI: clear operands, and encrypted zero padded ones
    (hidden lengths), against bytes.replace incl. growth, shrink, deletion and overlapping candidates."""
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    want = s.replace(frm, to)
    es = _enc(toy_k1, s, 6)
    out_cap = max(len(want), 1)
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.replace(es, frm, to, out_cap=out_cap))) == want
    enc_out = ops.replace(es, _enc(toy_k1, frm, 3), _enc(toy_k1, to, 4), out_cap=10)
    assert fhestr.blocks_to_string(P, _dec(toy_k1, enc_out)) == want


def test_replace_empty_clear_pattern_and_padded_strip_gpu(toy_k1):
    import fhestr
    ops = _ops(toy_k1)
    P = gpu_engine(toy_k1).params
    es = _enc(toy_k1, b"ab", 3)
    assert fhestr.blocks_to_string(P, _dec(toy_k1, ops.replace(es, b"", b"-", out_cap=8))) == b"ab".replace(b"", b"-")
    s = b"abcabc"
    es = _enc(toy_k1, s, 6)
    for pat in (b"abc", b"bc", b"", b"abcabc", b"x"):
        ep = _enc(toy_k1, pat, 6)
        bit, rest = ops.strip_prefix(es, ep)
        assert _dec(toy_k1, bit)[0] == int(s.startswith(pat))
        assert fhestr.blocks_to_string(P, _dec(toy_k1, rest)) == (s[len(pat):] if s.startswith(pat) else s)
        bit, rest = ops.strip_suffix(es, ep)
        assert _dec(toy_k1, bit)[0] == int(s.endswith(pat))
        assert fhestr.blocks_to_string(P, _dec(toy_k1, rest)) == (s[:len(s) - len(pat)] if s.endswith(pat) else s)


def test_p22_general_replace_64_chars(p22):
    """General replace on the real parameter set: growth and shrink on a 64-char string."""
    import fhestr
    ops = _ops(p22)
    P = gpu_engine(p22).params
    s = b"the cat and the hat sat on the mat; the end"
    es = _enc(p22, s, 48)
    for frm, to in ((b"the ", b"a "), (b"at", b"ouse")):
        want = s.replace(frm, to)
        got = fhestr.blocks_to_string(P, _dec(p22, ops.replace(es, frm, to, out_cap=64)))
        assert got == want


def test_page_locked_host_buffers_give_the_same_results(toy_k1):
    """fhe_host_alloc / fhestr.pinned_empty: operands and results in page-locked host memory (the fast path for
    1024-char strings, bench.py p44) -- same ciphertexts as with pageable numpy arrays, buffers freed with the arrays."""
    import fhestr
    eng = gpu_engine(toy_k1)
    s = b"Hello, World"
    pageable = _enc(toy_k1, s, 16)
    locked = fhestr.pinned_empty(pageable.shape)
    locked[...] = pageable
    calls = []

    def alloc(shape):
        calls.append(tuple(shape))
        return fhestr.pinned_empty(shape)

    want = fhestr.FheStringOps(eng).to_lower(pageable)
    got = fhestr.FheStringOps(eng, out_alloc=alloc).to_lower(locked)
    assert calls == [pageable.shape]
    assert np.array_equal(got, want)          # the blind rotation is deterministic: identical words
    assert fhestr.blocks_to_string(eng.params, _dec(toy_k1, got)) == s.lower()
    hit = fhestr.FheStringOps(eng, out_alloc=alloc).contains(locked, b"World")
    assert _dec(toy_k1, hit)[0] == 1
    del locked, got, hit
    zero = fhestr.pinned_empty((0, 3))
    assert zero.shape == (0, 3)
