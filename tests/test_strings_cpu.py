"""Host logic of the FheString plans without a GPU: the C++ planner builds offline plans, the CPU
oracle executes their exported levels, results are compared with Python string semantics.  Also the
world_size-2 gloo test of the sharded executor's protocol (fhestr/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, keyset, to_fhestr_params
from plan_oracle import run_with_oracle


def _plan(op, a_cap, b_cap=0, clear=None, world=1):
    import fhestr
    return fhestr.Plan.string_op(None, op, a_cap, b_cap, clear, world, params=to_fhestr_params(O.TOY_K1))


def _enc(ks, s, cap):
    import fhestr
    return ks.ck.encrypt_many(fhestr.string_to_blocks(to_fhestr_params(ks.params), s, cap))


def _run(ks, op, s, pat, pat_cap=4):
    if isinstance(pat, tuple):  # ("clear", bytes)
        plan = _plan(op + "_clear", 8, 0, pat[1])
        inputs = _enc(ks, s, 8)
    elif pat is None:
        plan = _plan(op, 8)
        inputs = _enc(ks, s, 8)
    else:
        plan = _plan(op, 8, pat_cap)
        inputs = np.concatenate([_enc(ks, s, 8), _enc(ks, pat, pat_cap)])
    return ks.ck.decrypt_many(run_with_oracle(plan, inputs, ks.sk))


CASES = [(b"hello wd", b"hell"), (b"hello wd", b"o wd"), (b"hello", b"lo"), (b"hello", b""), (b"", b""),
         (b"", b"a"), (b"aaa", b"aa"), (b"ab", b"abc"), (b"abab", b"bab")]


@pytest.mark.parametrize("s,pat", CASES)
def test_pattern_ops_offline_plan_vs_python(toy_k1, s, pat):
    for op, want in (("starts_with", s.startswith(pat)), ("ends_with", s.endswith(pat)), ("contains", pat in s),
                     ("eq", s == pat), ("ne", s != pat)):
        assert _run(toy_k1, op, s, pat)[0] == int(want), (op, "encrypted")
        assert _run(toy_k1, op, s, ("clear", pat))[0] == int(want), (op, "clear")
    want = s.find(pat)
    for p in (pat, ("clear", pat)):
        out = _run(toy_k1, "find", s, p)
        assert out[0] == int(want >= 0)
        if want >= 0:
            assert sum(int(d) * 4**i for i, d in enumerate(out[1:])) == want


@pytest.mark.parametrize("s", [b"Hello Wd", b"az AZ@[`", b""])
def test_case_ops_offline_plan_vs_python(toy_k1, s):
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    assert fhestr.blocks_to_string(P, _run(toy_k1, "to_upper", s, None)) == s.upper()
    assert fhestr.blocks_to_string(P, _run(toy_k1, "to_lower", s, None)) == s.lower()


@pytest.mark.parametrize("s", [b"  hi  ", b"\t a b\n", b"abc", b"    ", b"", b" x", b"x \r\n"])
def test_trim_ops_offline_plan_vs_python(toy_k1, s):
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    assert fhestr.blocks_to_string(P, _run(toy_k1, "trim_end", s, None)) == s.rstrip()
    assert fhestr.blocks_to_string(P, _run(toy_k1, "trim_start", s, None)) == s.lstrip()
    assert fhestr.blocks_to_string(P, _run(toy_k1, "strip", s, None)) == s.strip()


@pytest.mark.parametrize("s,frm,to", [(b"abcabc", b"bc", b"XY"), (b"aaaa", b"aa", b"bc"), (b"aaa", b"aa", b"xy"),
                                       (b"hello", b"zz", b"yy"), (b"abababab", b"aba", b"xyz"), (b"", b"a", b"b")])
def test_replace_offline_plan_vs_python(toy_k1, s, frm, to):
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    want = s.replace(frm, to)
    got = fhestr.blocks_to_string(P, _run(toy_k1, "replace", s, ("clear", frm + to)))
    assert got == want, ("clear", got, want)
    plan = _plan("replace", 8, 2 * len(frm))
    inputs = np.concatenate([_enc(toy_k1, s, 8), _enc(toy_k1, frm, len(frm)), _enc(toy_k1, to, len(to))])
    got = fhestr.blocks_to_string(P, toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk)))
    assert got == want, ("encrypted", got, want)


@pytest.mark.parametrize("s", [b"hello", b"", b"abcdefgh", b"a"])
def test_len_is_empty_offline_plan_vs_python(toy_k1, s):
    digits = _run(toy_k1, "len", s, None)
    assert sum(int(d) * 4**i for i, d in enumerate(digits)) == len(s)
    assert _run(toy_k1, "is_empty", s, None)[0] == int(len(s) == 0)


@pytest.mark.parametrize("s,pat", [(b"abcabc", b"bc"), (b"aaaa", b"aa"), (b"hello", b"zz"), (b"abab", b"")])
def test_rfind_and_ignore_case_offline_plan_vs_python(toy_k1, s, pat):
    want = s.rfind(pat)
    for p in (pat, ("clear", pat)):
        out = _run(toy_k1, "rfind", s, p)
        assert out[0] == int(want >= 0)
        if want >= 0:
            assert sum(int(d) * 4**i for i, d in enumerate(out[1:])) == want
    up = s.upper()
    assert _run(toy_k1, "eq_ignore_case", s, up, pat_cap=8)[0] == 1
    assert _run(toy_k1, "eq_ignore_case", s, ("clear", up))[0] == 1
    assert _run(toy_k1, "eq_ignore_case", s, s + b"x", pat_cap=8)[0] == 0


@pytest.mark.parametrize("s,pat", [(b"prefix-x", b"pre"), (b"prefix-x", b"-x"), (b"abc", b"abc"), (b"abc", b"x"), (b"aaa", b"a")])
def test_strip_affix_offline_plan_vs_python(toy_k1, s, pat):
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    out = _run(toy_k1, "strip_prefix", s, ("clear", pat))
    assert out[0] == int(s.startswith(pat))
    assert fhestr.blocks_to_string(P, out[1:]) == (s[len(pat):] if s.startswith(pat) else s)
    out = _run(toy_k1, "strip_suffix", s, ("clear", pat))
    assert out[0] == int(s.endswith(pat))
    assert fhestr.blocks_to_string(P, out[1:]) == (s[: len(s) - len(pat)] if s.endswith(pat) else s)


@pytest.mark.parametrize("a,b", [(b"abc", b"abd"), (b"abc", b"abc"), (b"abc", b"ab"), (b"", b"a"), (b"", b""),
                                  (b"b", b"abcd"), (b"zz", b"za"), (b"Abc", b"abc")])
def test_lexicographic_order_offline_plan_vs_python(toy_k1, a, b):
    for op, want in (("lt", a < b), ("le", a <= b), ("gt", a > b), ("ge", a >= b)):
        assert _run(toy_k1, op, a, b)[0] == int(want), (op, "encrypted")
        assert _run(toy_k1, op, a, ("clear", b))[0] == int(want), (op, "clear")


@pytest.mark.parametrize("a,b", [(b"hello", b"hello"), (b"hello", b"hellp"), (b"ab", b"abc"), (b"", b""), (b"\x7f~}|", b"\x7f~}|"), (b"\x00", b"\x7f")])
def test_reference_shaped_eq_ne_agree_with_packed(toy_k1, a, b):
    for op, want in (("eq", a == b), ("ne", a != b), ("eq_reference", a == b), ("ne_reference", a != b)):
        assert _run(toy_k1, op, a, b, pat_cap=8)[0] == int(want), op


def test_plan_shapes_match_survey_counts():
    # eq enc-enc 256 chars = 1024 + 69 + 5 + 1 PBS, depth 4; enc-clear = 551
    info = _plan("eq_reference", 256, 256).info()
    assert (info["n_pbs"], info["n_levels"], info["n_inputs"], info["n_outputs"]) == (1099, 4, 2048, 1)
    # default: two blocks per PBS through the padding bit (fhe_string.cpp: packed_pair_eq)
    # ... and reductions of 16 bits per lookup (Circuit::pbs_full_box) instead of the reference's 15
    assert _plan("eq", 256, 256).info()["n_pbs"] == 512 + 32 + 2 + 1
    info = _plan("eq_clear", 256, 0, b"x" * 200).info()
    assert (info["n_pbs"], info["n_levels"]) == (547, 4)
    # contains, 16-char encrypted pattern in 256 chars: AND over the 16 chars and OR over the 256 offsets lose a level each,
    # and two pattern characters share one lookup of the (offset, character) level (fhe_string.cpp: group_match)
    plan = _plan("contains", 256, 16)
    assert [plan.level_info(l)["jobs"] for l in range(plan.info()["n_levels"])] == [7968, 1992, 256, 16, 1]
    plan = _plan("contains_reference", 256, 16)
    assert [plan.level_info(l)["jobs"] for l in range(plan.info()["n_levels"])] == [15920, 3991, 497, 256, 18, 2, 1]
    lv = [_plan("eq_reference", 256, 256).level_info(l)["jobs"] for l in range(4)]
    assert lv == [1024, 69, 5, 1]


def test_degree_overflow_is_a_build_error():
    import fhestr
    plan = fhestr.Plan(None, params=to_fhestr_params(O.TOY_K1))
    a, b = plan.input(3), plan.input(3)
    ident = plan.lut(lambda x: x)
    with pytest.raises(fhestr.FheError):
        plan.pbs(plan.lin([(a, 4), (b, 4)]), ident)   # 3*4 + 3*4 = 24 > 15


def test_offline_plan_refuses_to_run():
    import fhestr
    plan = _plan("eq", 4, 4)
    with pytest.raises(fhestr.FheError):
        plan.run(np.zeros((plan.info()["n_inputs"], O.TOY_K1.big_size), dtype=np.uint64))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SHARDED_CASES = [   # (op, a_cap, b_cap, clear, a, b)
    ("contains", 8, 4, None, b"abcabd", b"abd"),
    ("eq", 7, 7, None, b"abcdefg", b"abcdefg"),          # 7 characters over 4 ranks: uneven slices
    ("eq", 7, 7, None, b"abcdefg", b"abcdefh"),
    ("find", 7, 3, None, b"xxabdab", b"abd"),            # prefix scan: ownership decided by the leveliser
    ("to_lower", 5, 0, None, b"HeLLo", None),            # every output block is gathered
    ("replace_clear:2:9", 6, 0, b"bcXYZ", b"abcabc", None),
]


def _gloo_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "fhe-string-bounty_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from fhestr.distributed import ShardedPlanRunner
    from plan_oracle import OracleBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ks = keyset(O.TOY_K1)
        out = []
        for op, a_cap, b_cap, clear, a, b in SHARDED_CASES:
            plan = _plan(op, a_cap, b_cap, clear, world=world)
            inputs = _enc(ks, a, a_cap)                   # deterministic per seed: same ciphertexts on every rank
            if b is not None:
                inputs = np.concatenate([inputs, _enc(ks, b, b_cap)])
            runner = ShardedPlanRunner(plan, rank, world, OracleBackend(plan, ks.sk))
            res = ks.ck.decrypt_many(runner.run(inputs)).tolist()
            out.append((res, runner.gathered_lwes, runner.collectives))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _run_sharded(world):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    return dict(ret)


def _single_rank_results(ks):
    out = []
    for op, a_cap, b_cap, clear, a, b in SHARDED_CASES:
        plan = _plan(op, a_cap, b_cap, clear)
        inputs = _enc(ks, a, a_cap)
        if b is not None:
            inputs = np.concatenate([inputs, _enc(ks, b, b_cap)])
        out.append(ks.ck.decrypt_many(run_with_oracle(plan, inputs, ks.sk)).tolist())
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_runner_gloo(toy_k1, world):
    """N>1 path on CPU (gloo): every rank runs only the jobs it owns, exported ciphertexts are all-gathered,
    all ranks agree with each other and with the single-rank run.  World 4 splits 7- and 5-character
    strings unevenly (some ranks own one character, some two; some own nothing at the small levels)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    single = _single_rank_results(toy_k1)
    assert single[0] == [1] and single[1] == [1] and single[2] == [0]
    assert fhestr.blocks_to_string(P, single[4]) == b"hello"
    assert fhestr.blocks_to_string(P, single[5]) == b"aXYZaXYZ"
    ret = _run_sharded(world)
    assert sorted(ret) == list(range(world))
    for r in range(world):
        assert [res for res, _, _ in ret[r]] == single, f"rank {r}"
    # eq: one block per rank gathered after the local reduction, then the final bit
    eq_gathered, eq_collectives = ret[0][1][1], ret[0][1][2]
    assert eq_gathered == 2 * world and eq_collectives == 2


def test_config1_eq_8_chars_p22_cpu_reference_path(p22):
    """BASELINE.json configs[0] / SURVEY.md 8(d) config 1: FheString::eq on two 8-char ASCII strings,
    PARAM_MESSAGE_2_CARRY_2, CPU path only (the planner's levels executed by the oracle), keys from
    seed 0x5EED0001; "fhe-str!" vs itself -> 1, vs "fhe-str?" -> 0; 32 block PBS + 3 + 1 reduce."""
    import fhestr
    P = to_fhestr_params(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
    plan = fhestr.Plan.string_op(None, "eq_reference", 8, 8, params=P)
    info = plan.info()
    assert (info["n_pbs"], info["n_levels"]) == (32 + 3 + 1, 3)
    enc = lambda s: p22.ck.encrypt_many(fhestr.string_to_blocks(P, s, 8))
    a = enc(b"fhe-str!")
    for other, want in ((b"fhe-str!", 1), (b"fhe-str?", 0)):
        out = run_with_oracle(plan, np.concatenate([a, enc(other)]), p22.sk)
        assert p22.ck.decrypt_many(out)[0] == want


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_circuits_offline_plan_vs_clear_semantics(toy_k1, seed):
    """Planner + leveliser on random lin/pbs DAGs: oracle-executed plan == clear evaluation."""
    import fhestr
    from random_circuits import build_random_circuit
    rng = np.random.default_rng(seed)
    plan = fhestr.Plan(None, params=to_fhestr_params(O.TOY_K1))
    evaluate = build_random_circuit(plan, rng)
    plan.finalize(1)
    values = [int(v) for v in rng.integers(0, 4, size=6)]
    out = run_with_oracle(plan, toy_k1.ck.encrypt_many(values), toy_k1.sk)
    assert toy_k1.ck.decrypt_many(out).tolist() == evaluate(values)


@pytest.mark.parametrize("a,b", [(b"hello", b"wd"), (b"", b"abc"), (b"abc", b""), (b"", b""), (b"abcdefgh", b"ijkl"),
                                 (b"a", b"b"), (b"ab cd", b" ef")])
def test_concat_repeat_offline_plan_vs_python(toy_k1, a, b):
    """concat removes a's padding; repeat is iterated concat (Rust's `+` and str::repeat)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    assert fhestr.blocks_to_string(P, _run(toy_k1, "concat", a, b)) == a + b                    # out cap 8 + 4
    assert fhestr.blocks_to_string(P, _run(toy_k1, "concat", a, ("clear", b))) == a + b
    if len(a) <= 4:
        plan = _plan("repeat_clear", 4, 0, bytes([3]))
        got = toy_k1.ck.decrypt_many(run_with_oracle(plan, _enc(toy_k1, a, 4), toy_k1.sk))
        assert fhestr.blocks_to_string(P, got) == a * 3


# ---- general replace (|from| != |to|), clear and encrypted (padded) patterns ---------------------
GENERAL_REPLACE = [
    (b"abcabc", b"bc", b"X"),          # shrink
    (b"abcabc", b"b", b"XYZ"),         # growth
    (b"aaaa", b"aa", b"b"),            # adjacent occurrences
    (b"aaa", b"aa", b"xyz"),           # overlapping candidates: leftmost wins
    (b"hello", b"l", b""),             # deletion
    (b"hello", b"zz", b"y"),           # no occurrence
    (b"abab", b"ab", b"ab"),           # same length through the general path
    (b"", b"a", b"bc"),                # empty string
    (b"abc", b"abc", b"z"),            # whole string
    (b"abcabc", b"abc", b"abcd"),      # result fills the output capacity
]


@pytest.mark.parametrize("s,frm,to", GENERAL_REPLACE)
def test_general_replace_clear_offline_plan_vs_python(toy_k1, s, frm, to):
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    want = s.replace(frm, to)
    out_cap = max(len(want), 1)
    plan = _plan(f"replace_clear:{len(frm)}:{out_cap}", 6, 0, frm + to)
    got = toy_k1.ck.decrypt_many(run_with_oracle(plan, _enc(toy_k1, s, 6), toy_k1.sk))
    assert fhestr.blocks_to_string(P, got) == want
    assert plan.info()["n_outputs"] == out_cap * 4


@pytest.mark.parametrize("s,frm,to", GENERAL_REPLACE[:7])
def test_general_replace_encrypted_padded_offline_plan_vs_python(toy_k1, s, frm, to):
    """Encrypted `from` / `to` with hidden lengths (capacity 3, zero padded)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    want = s.replace(frm, to)
    out_cap = 10
    plan = _plan(f"replace:3:{out_cap}", 6, 6)
    inputs = np.concatenate([_enc(toy_k1, s, 6), _enc(toy_k1, frm, 3), _enc(toy_k1, to, 3)])
    got = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))
    assert fhestr.blocks_to_string(P, got) == want


def test_replace_empty_patterns(toy_k1):
    """Clear empty `from`: `to` before every character and at the end (bytes.replace(b"", to)); an
    encrypted `from` that decrypts to the empty string replaces nothing (documented in fhestr.h)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    for s in (b"ab", b"", b"abc"):
        want = s.replace(b"", b"-")
        plan = _plan("replace_clear:0:8", 3, 0, b"-")
        got = toy_k1.ck.decrypt_many(run_with_oracle(plan, _enc(toy_k1, s, 3), toy_k1.sk))
        assert fhestr.blocks_to_string(P, got) == want
    plan = _plan("replace:2:6", 4, 4)
    inputs = np.concatenate([_enc(toy_k1, b"abab", 4), _enc(toy_k1, b"", 2), _enc(toy_k1, b"zz", 2)])
    got = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))
    assert fhestr.blocks_to_string(P, got) == b"abab"


@pytest.mark.parametrize("s,pat", [(b"abcabc", b"abc"), (b"abcabc", b"bc"), (b"abc", b""), (b"", b""), (b"ab", b"abc"),
                                   (b"aaa", b"a"), (b"abcabc", b"abcabc")])
def test_strip_prefix_suffix_encrypted_pattern(toy_k1, s, pat):
    """strip_prefix / strip_suffix with an encrypted, zero padded pattern == Rust's strip_prefix /
    Python's removeprefix (+ the "was there" bit)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    for op, had, want in (("strip_prefix", s.startswith(pat), s[len(pat):] if s.startswith(pat) else s),
                          ("strip_suffix", s.endswith(pat), s[:len(s) - len(pat)] if s.endswith(pat) else s)):
        plan = _plan(op, 6, 6)
        inputs = np.concatenate([_enc(toy_k1, s, 6), _enc(toy_k1, pat, 6)])
        out = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))
        assert out[0] == int(had), op
        assert fhestr.blocks_to_string(P, out[1:]) == want, op


# ---- value ranges and noise are checked when a plan is built -------------------------------------
def test_plan_rejects_negative_and_noisy_pbs_inputs():
    import fhestr
    P = to_fhestr_params(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
    plan = fhestr.Plan(None, params=P)
    a, b = plan.input(3), plan.input(3)
    ident = plan.lut(lambda x: x)
    diff = plan.lin([(a, 1), (b, -1)])
    with pytest.raises(fhestr.FheError, match="negative"):
        plan.pbs(diff, ident)                         # a - b may wrap into the padding bit
    plan.pbs(plan.lin([(a, 1), (b, -1)], 3), ident)   # the reference's correcting constant: fine
    plan.pbs(diff, plan.lut(lambda x: int(x == 0)), signed=True)   # declared use of the padding bit
    with pytest.raises(fhestr.FheError, match="overflows"):
        plan.pbs(plan.lin([(a, 4), (b, 2)]), ident)   # 12 + 6 > 15
    m = fhestr.noise_model(P)
    assert 25 <= m["budget"] < 400 and m["log2_pfail_at_budget"] < -39
    narrow = plan.input(0)
    with pytest.raises(fhestr.FheError, match="noise"):
        plan.pbs(plan.lin([(narrow, 15)]), ident)     # 225 nominal variances
    plan.set_noise_budget(0)                          # unchecked, like the reference's unchecked_* ops
    plan.pbs(plan.lin([(narrow, 15)]), ident)


def test_noise_model_matches_the_reference_parameter_sets():
    """The model's worst-case failure probability at the reference's own noise bound (norm2 =
    max_noise_level) is what the parameter sets are generated for: ~2^-40
    (docs/getting_started/security_and_cryptography.md:96)."""
    import fhestr
    for P in (fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS):
        m = fhestr.noise_model(P)
        std = (m["v_ks"] + m["v_ms"]) ** 0.5
        assert 6.8 < m["half_box"] / std < 8.5, P.name
        assert -55 < m["log2_pfail_at_budget"] < -39, (P.name, m)


# ---- sharded plans: owners, exports, what travels ------------------------------------------------
def test_sharded_eq_reduces_locally_and_gathers_one_block_per_rank():
    """FheString::eq on 256 chars over 8 ranks -- every rank compares its 32 characters and
    reduces them to ONE block; two all-gathers of one ciphertext per rank (the per-rank results, then the
    final bit) instead of every level's whole output."""
    import fhestr
    P = to_fhestr_params(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
    plan = fhestr.Plan.string_op(None, "eq", 256, 256, world=8, params=P)
    info = plan.info()
    levels = [plan.level_info(l) for l in range(info["n_levels"])]
    assert info["n_levels"] == 4
    assert [lv["e_max"] for lv in levels] == [0, 0, 1, 1]
    gathered = sum(lv["e_max"] * 8 for lv in levels)
    assert gathered == 16 and gathered * P.big_size * 8 < 1.2e6      # SURVEY's 1.18 MB figure is the ceiling
    assert plan.noise_info()["gathered_lwes"] == 16
    for r in range(8):          # level 1: 64 packed compares per rank, all private
        ri = plan.level_rank_info(0, r)
        assert ri["job_hi"] - ri["job_lo"] == 64 and ri["n_export"] == 0
    single = fhestr.Plan.string_op(None, "eq", 256, 256, world=1, params=P).info()
    assert info["n_pbs"] <= single["n_pbs"] + 16      # rank-aligned reduction groups: a few more small PBS


def test_sharded_contains_keeps_offsets_on_their_rank():
    import fhestr
    P = to_fhestr_params(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
    plan = fhestr.Plan.string_op(None, "contains", 256, 16, world=8, params=P)
    info = plan.info()
    levels = [plan.level_info(l) for l in range(info["n_levels"])]
    gathered = sum(lv["e_max"] * 8 for lv in levels)
    # the pattern's 16 zero-flags are shared by every offset (computed once, exported), then one block per rank
    assert gathered * P.big_size * 8 < 4e6, gathered
    biggest = max(range(len(levels)), key=lambda l: levels[l]["jobs"])
    sizes = [plan.level_rank_info(biggest, r)["job_hi"] - plan.level_rank_info(biggest, r)["job_lo"] for r in range(8)]
    assert max(sizes) <= 1.35 * min(sizes)          # balanced (the last offsets run past the capacity: fewer compares)


def test_string_plans_stay_inside_the_noise_budget():
    """Every FheString plan on the real parameter sets: the noisiest PBS input is within the budget (the
    builder would have refused it otherwise) and its modelled failure probability is reported."""
    import fhestr
    for P, a_cap in ((fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, 64), (fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS, 64)):
        for op, b_cap, clear in (("eq", a_cap, None), ("contains", 8, None), ("find", 8, None), ("to_lower", 0, None),
                                 ("strip", 0, None), ("replace_clear", 0, b"\xff\xfe\xfd\xfc" + b"\xff\xff\xff\xff"),
                                 ("replace_clear:2:80", 0, b"ab\xff\xff\xff"), ("eq_ignore_case", a_cap, None), ("lt", a_cap, None)):
            plan = fhestr.Plan.string_op(None, op, a_cap, b_cap, clear, params=P)
            ni = plan.noise_info()
            assert 0 < ni["max_pbs_input_noise"] <= ni["budget"], (P.name, op, ni)
            assert ni["log2_pfail_worst"] < -38.5, (P.name, op, ni)


# ---- 4-bit blocks (PARAM_MESSAGE_4_CARRY_4 geometry, toy LWE dimension): whole-character lookups and the
#      class-doubling clear-pattern matcher, plans run by the CPU oracle ---------------------------------------------
def _plan4(op, a_cap, b_cap=0, clear=None):
    import fhestr
    return fhestr.Plan.string_op(None, op, a_cap, b_cap, clear, 1, params=to_fhestr_params(O.TOY_N32768))


def test_whole_character_lookups_on_4_bit_blocks():
    import fhestr
    ks = keyset(O.TOY_N32768)
    P = to_fhestr_params(ks.params)
    s = b"Az [`{@Zq"
    for op, want in (("to_lower", s.lower()), ("to_upper", s.upper())):
        plan = _plan4(op, 10)
        assert plan.info()["n_pbs"] == 10 and plan.info()["n_levels"] == 1        # one PBS per character
        assert plan.noise_info()["max_pbs_input_noise"] == 257.0                  # hi * 16 + lo of two nominal blocks
        got = fhestr.blocks_to_string(P, ks.ck.decrypt_many(run_with_oracle(plan, _enc(ks, s, 10), ks.sk)))
        assert got == want
    plan = _plan4("strip", 6)
    got = fhestr.blocks_to_string(P, ks.ck.decrypt_many(run_with_oracle(plan, _enc(ks, b" \tab ", 6), ks.sk)))
    assert got == b"ab"


@pytest.mark.parametrize("hay,pat", [(b"abcabcab", b"abcab"), (b"the cat the", b"the "), (b"aaaaaa", b"aa"), (b"xyzxyz", b"zx"),
                                     (b"abcdefgh", b"cdefgh"), (b"abcdefgh", b"zz"), (b"aabaab", b"aab")])
def test_clear_patterns_by_classes_offline_plan_vs_python(hay, pat):
    import fhestr
    ks = keyset(O.TOY_N32768)
    P = to_fhestr_params(ks.params)
    cap = 12
    enc = _enc(ks, hay, cap)
    plan = _plan4("find_clear", cap, 0, pat)
    out = ks.ck.decrypt_many(run_with_oracle(plan, enc, ks.sk))
    want = hay.find(pat)
    assert int(out[0]) == (want >= 0)
    if want >= 0:
        assert sum(int(d) * P.msg_mod ** i for i, d in enumerate(out[1:])) == want
    plan = _plan4("contains_clear", cap, 0, pat)
    # classify + double: about (1 + ceil(log2 len)) lookups per position, far below one per (position, distinct character)
    assert plan.info()["n_pbs"] <= cap * (2 + len(pat).bit_length()) + 4
    assert int(ks.ck.decrypt_many(run_with_oracle(plan, enc, ks.sk))[0]) == (pat in hay)
    if len(set(pat)) > 1 or len(pat) == 1:
        to = bytes(reversed(pat)).upper()
        plan = _plan4("replace_clear", cap, 0, pat + to)
        got = fhestr.blocks_to_string(P, ks.ck.decrypt_many(run_with_oracle(plan, enc, ks.sk)))
        assert got == hay.replace(pat, to)


def test_full_box_reduction_every_sum_and_every_kind_of_consumer(toy_k1):
    """Circuit::pbs_full_box (fhe_plan_pbs_full_box): T = msg*carry bits in one lookup.  Every sum 0 .. T through `all`
    and `any`, the result read directly, negated (1 - b), summed with its twin and fed to another lookup -- the +1/2
    the consumers owe (Node::half) must arrive in every position.  Oracle-executed (bit-exact integer bookkeeping:
    a wrong constant would shift the next lookup by half a box)."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    T = P.msg_mod * P.carry_mod
    plan = fhestr.Plan(None, params=P)
    bits = [plan.input(1) for _ in range(T)]
    total = plan.lin([(b, 1) for b in bits])
    all_b = plan.pbs_full_box(total, True)
    any_b = plan.pbs_full_box(total, False)
    ident = plan.lut(lambda i: i)
    plan.output(all_b)
    plan.output(any_b)
    plan.output(plan.lin([(all_b, -1)], 1))                              # not all
    plan.output(plan.lin([(all_b, 2), (any_b, 3)], 1))                   # 1 + 2 all + 3 any
    plan.output(plan.pbs(plan.lin([(all_b, 1), (any_b, 1), (bits[0], 1)]), ident))
    with pytest.raises(fhestr.FheError, match="must lie in"):
        plan.pbs_full_box(plan.lin([(b, 1) for b in bits] + [(bits[0], 1)]), True)     # up to T + 1
    plan.finalize(1)
    assert plan.info()["n_levels"] == 2
    for s in range(T + 1):
        values = [1] * s + [0] * (T - s)
        values = values[::-1] if s % 2 else values
        out = toy_k1.ck.decrypt_many(run_with_oracle(plan, toy_k1.ck.encrypt_many(values), toy_k1.sk)).tolist()
        a, o = int(s == T), int(s != 0)
        assert out == [a, o, 1 - a, 1 + 2 * a + 3 * o, a + o + values[0]], (s, out)
    # trivial inputs fold at build time
    plan = fhestr.Plan(None, params=P)
    x = plan.input(1)
    t = plan.pbs_full_box(plan.lin([], T), True)
    plan.output(plan.lin([(x, 1), (t, 1)]))
    plan.finalize(1)
    assert plan.info()["n_pbs"] == 0


@pytest.mark.parametrize("s,pat", [(b"xxxxxxxxxxxxxxxxxab", b"ab"), (b"abxxxxxxxxxxxxxxxab", b"ab"), (b"xxxxxxxxxxxxxxxxxxx", b"ab"),
                                   (b"xxxxxxxxxxxxxxxab", b"ab"), (b"", b"a")])
def test_find_contains_over_more_than_16_offsets(toy_k1, s, pat):
    """20-char haystack, 2-char pattern: 19 .. 21 candidate offsets, so the OR over the offsets and find's prefix scan
    meet runs of exactly T = 16 bits (Circuit::pbs_full_box) next to shorter ones.  Oracle-executed vs Python."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    inputs = np.concatenate([_enc(toy_k1, s, 20), _enc(toy_k1, pat, 2)])
    for op in ("contains", "find", "rfind", "ends_with"):
        plan = fhestr.Plan.string_op(None, op, 20, 2, None, 1, params=P)
        out = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk)).tolist()
        if op == "contains":
            assert out == [int(pat in s)], (op, out)
        elif op == "ends_with":
            assert out == [int(s.endswith(pat))], (op, out)
        else:
            idx = s.find(pat) if op == "find" else s.rfind(pat)
            digits = out[1:]
            got = sum(d * P.msg_mod**i for i, d in enumerate(digits))
            assert out[0] == int(idx >= 0) and (idx < 0 or got == idx), (op, out, idx)


@pytest.mark.parametrize("pat_len", [0, 1, 2, 3, 4, 5])
def test_two_pattern_characters_per_lookup_every_padding_position(toy_k1, pat_len):
    """group_match (fhe_string.cpp): pairs of pattern characters share one lookup, weighted so that padding counts as a
    match and one wrong block of a live character cannot be made up.  Pattern capacity 5 (two pairs + a single), every
    length 0 .. 5 (the padding starts inside a pair, between pairs, nowhere), haystacks that match, that differ in the
    first / second character of a pair only, in one block only, and windows that run past the end of the haystack."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    pat = b"abcde"[:pat_len]
    hays = [b"xxabcdex", b"abcdexxx", b"xxxabcde", b"xxabcdfx", b"xxaccdex", b"xxbbcdex", b"xxabcd", b"abcd", b"", b"xxabqdex",
            b"xxabcdeq"[:8], b"xabcdabc"]
    for op in ("contains", "starts_with", "find"):
        plan = fhestr.Plan.string_op(None, op, 8, 5, None, 1, params=P)
        for hay in hays:
            inputs = np.concatenate([_enc(toy_k1, hay, 8), _enc(toy_k1, pat, 5)])
            out = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk)).tolist()
            if op == "contains":
                assert out == [int(pat in hay)], (op, hay, pat, out)
            elif op == "starts_with":
                assert out == [int(hay.startswith(pat))], (op, hay, pat, out)
            else:
                idx = hay.find(pat)
                got = sum(d * P.msg_mod**i for i, d in enumerate(out[1:]))
                assert out[0] == int(idx >= 0) and (idx < 0 or got == idx), (op, hay, pat, out)


@pytest.mark.parametrize("s,frm", [(b"aaaaaaa", b"aa"), (b"aaaaaa", b"aaa"), (b"abaabaab", b"aba"), (b"aaaaaaaa", b"a"), (b"ababababa", b"ab"),
                                   (b"aabaabaa", b"aab"), (b"xaaaaaax", b"aaa"), (b"aaaaaaaa", b"aaa")])
def test_replace_occurrence_recurrence_two_offsets_per_level(toy_k1, s, frm):
    """occurrences() (fhe_string.cpp) settles two offsets per lookup level: self-overlapping patterns of every true
    length inside a capacity-3 padded encrypted pattern (the blocking of the next offset depends on whether the pattern is
    longer than one character), leftmost non-overlapping selection as bytes.replace does it.  Oracle-executed."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    to = b"XYZ"[:len(frm)]
    want = s.replace(frm, to)
    cap = 9
    # equal-length in-place form, unpadded encrypted operands
    plan = _plan("replace", cap, 2 * len(frm))
    inputs = np.concatenate([_enc(toy_k1, s, cap), _enc(toy_k1, frm, len(frm)), _enc(toy_k1, to, len(to))])
    assert fhestr.blocks_to_string(P, toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))) == want
    assert plan.info()["n_levels"] < cap + 6            # one level per two offsets, not per offset
    # general form, zero padded encrypted operands of capacity 3 (hidden length)
    plan = _plan(f"replace:3:{cap}", cap, 6)
    inputs = np.concatenate([_enc(toy_k1, s, cap), _enc(toy_k1, frm, 3), _enc(toy_k1, to, 3)])
    assert fhestr.blocks_to_string(P, toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))) == want


@pytest.mark.parametrize("s,pat", [(b"xxxxab", b"ab"), (b"abxxab", b"ab"), (b"xxxxxx", b"ab"), (b"xabab", b"ab")])
def test_find_on_one_bit_blocks(s, pat):
    """PARAM_MESSAGE_1_CARRY_1's box holds T = 4 values: find's first-hit test cannot read (match, hit earlier in the run,
    hit in an earlier run) side by side there (find_clear was refused with 'input degree 5 overflows'); the two
    "before" bits are OR-ed first.  Toy twin of the set (N = 512, k = 3), oracle-executed.  Encrypted patterns pack two
    pattern characters per lookup and need a box of 16 values: on this set they are refused at plan build, cleanly."""
    import fhestr
    p = next(q for q in O.TOY_SHAPES if q.name == "TOY_N512_K3")
    assert p.msg_mod * p.carry_mod == 4
    ks = keyset(p)
    P = to_fhestr_params(p)
    inputs = ks.ck.encrypt_many(fhestr.string_to_blocks(P, s, 6))
    for op in ("find", "rfind"):
        idx = s.find(pat) if op == "find" else s.rfind(pat)
        plan = fhestr.Plan.string_op(None, op + "_clear", 6, 0, pat, 1, params=P)
        out = ks.ck.decrypt_many(run_with_oracle(plan, inputs, ks.sk)).tolist()
        got = sum(d * P.msg_mod**i for i, d in enumerate(out[1:]))
        assert out[0] == int(idx >= 0) and (idx < 0 or got == idx), (op, out, idx)
    with pytest.raises(fhestr.FheError, match="overflows"):
        fhestr.Plan.string_op(None, "find", 6, 2, None, 1, params=P)


# ---- many instances of one plan, sharded over ranks (fhestr.distributed.run_instances_sharded) ----
BATCH_ROWS = [b"hello", b"hellp", b"", b"hell", b"hello!!", b"HELLO", b"hello"]      # 7 instances: uneven over 2 and 4 ranks


def _instances_worker(rank, world, port, ret):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fhestr
    from fhestr.distributed import run_instances_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ks = keyset(O.TOY_K1)
    P = to_fhestr_params(O.TOY_K1)
    plan = fhestr.Plan.string_op(None, "eq", 8, 8, None, 1, params=P)      # world 1: instances shard, levels do not
    pat = _enc(ks, b"hello", 8)
    inputs = np.stack([np.concatenate([_enc(ks, r, 8), pat]) for r in BATCH_ROWS])
    calls = []

    def run_batch(x):       # the checker's stand-in for fhe_plan_run_batch: the oracle, one instance at a time
        calls.append(x.shape[0])
        return np.stack([run_with_oracle(plan, inst, ks.sk) for inst in x])

    out = run_instances_sharded(run_batch, inputs, rank, world)
    ret[rank] = (ks.ck.decrypt_many(out[:, 0]).tolist(), calls)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_instances_sharded_over_ranks_gloo(world):
    """Many strings against one pattern over several ranks: every rank runs a contiguous slice of the instances (no
    data-path collective), the outputs are gathered; all ranks hold all results.  7 instances over 2 and 4 ranks."""
    import torch.multiprocessing as mp
    from fhestr.distributed import instance_slice
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_instances_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    want = [int(r == b"hello") for r in BATCH_ROWS]
    slices = [instance_slice(len(BATCH_ROWS), r, world) for r in range(world)]
    assert slices[0][0] == 0 and slices[-1][1] == len(BATCH_ROWS) and all(a[1] == b[0] for a, b in zip(slices, slices[1:]))
    for r in range(world):
        got, calls = ret[r]
        assert got == want, f"rank {r}"
        assert calls == [slices[r][1] - slices[r][0]]


# ---- replace with an ENCRYPTED pattern at scale: the blocked occurrence scan (fhe_string.cpp: occurrences_scan) ----
def _periodic(unit, n):
    return (unit * (n // len(unit) + 1))[:n]


SCAN_CASES = [
    # (string, from, to): self-overlapping patterns are the point -- leftmost non-overlapping occurrences (bytes.replace)
    (_periodic(b"a", 64), b"aa", b"XY"),                       # every offset matches; pairs
    (_periodic(b"ab", 64), b"aba", b"XYZ"),                    # borders of length 1
    (_periodic(b"ab", 63) + b"b", b"abab", b"WXYZ"),           # borders of length 2, a broken tail
    (b"x" * 20 + b"abcd" + b"y" * 17 + b"abcdabcd" + b"z" * 15, b"abcd", b"1234"),   # isolated and adjacent occurrences
    (b"q" * 64, b"abc", b"XYZ"),                               # no occurrence at all
    (b"abc" + b"q" * 58 + b"abc", b"abc", b"XYZ"),             # first and last offset
    (_periodic(b"aab", 50), b"aa", b"ZZ"),                     # shorter than the capacity: padded string
]


@pytest.mark.parametrize("s,frm,to", SCAN_CASES)
def test_replace_encrypted_pattern_blocked_scan_in_place(toy_k1, s, frm, to):
    """64-character strings, unpadded encrypted `from` / `to` of 2 .. 4 characters, rewritten in place: the occurrence
    recurrence runs as a blocked scan (state = characters still covered, one lookup per step, blocks evaluated for every
    incoming state, state maps composed block by block).  Oracle-executed plan == bytes.replace; far fewer levels than
    the two-offsets-per-level recurrence."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    m = len(frm)
    plan = _plan("replace", 64, 2 * m)
    info = plan.info()
    assert info["n_levels"] < 64 // 2, info          # the recurrence alone would need (64 - m) / 2 levels
    inputs = np.concatenate([_enc(toy_k1, s, 64), _enc(toy_k1, frm, m), _enc(toy_k1, to, m)])
    got = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))
    assert fhestr.blocks_to_string(P, got) == s.replace(frm, to)


@pytest.mark.parametrize("s,frm,to", [(_periodic(b"ab", 56), b"aba", b"-"), (_periodic(b"a", 56), b"aa", b"XYZ"),
                                      (b"x" * 20 + b"abcd" + b"y" * 20 + b"abcdabcd", b"abcd", b""),
                                      (_periodic(b"ab", 56), b"", b"Q"), (_periodic(b"abc", 56), b"bc", b"bc")])
def test_replace_encrypted_padded_pattern_blocked_scan(toy_k1, s, frm, to):
    """The same with hidden lengths: `from` / `to` zero padded to capacity 4, any lengths, caller-given output capacity.
    The scan's step then reads g[o] = match[o] * (L - 1) with the encrypted pattern length L."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    want = s.replace(frm, to) if frm else s                 # an encrypted pattern that decrypts to "" replaces nothing
    out_cap = max(len(want), 1)
    plan = _plan(f"replace:4:{out_cap}", 56, 8)
    inputs = np.concatenate([_enc(toy_k1, s, 56), _enc(toy_k1, frm, 4), _enc(toy_k1, to, 4)])
    got = toy_k1.ck.decrypt_many(run_with_oracle(plan, inputs, toy_k1.sk))
    assert fhestr.blocks_to_string(P, got) == want


def test_blocked_scan_depth_at_1024_characters():
    """The plan shape bench.py times as replace_enc_4_in_1024: 2 B + n / B levels for the scan instead of n / 2."""
    import fhestr
    P = to_fhestr_params(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
    plan = fhestr.Plan.string_op(None, "replace", 1024, 8, None, 1, params=P)
    info = plan.info()
    assert info["n_levels"] < 130, info
    print("replace (encrypted 4-character pattern, 1024 characters):", info["n_levels"], "levels,", info["n_pbs"], "PBS")
