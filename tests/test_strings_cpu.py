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
    assert _plan("eq", 256, 256).info()["n_pbs"] == 512 + 35 + 3 + 1
    info = _plan("eq_clear", 256, 0, b"x" * 200).info()
    assert (info["n_pbs"], info["n_levels"]) == (551, 4)
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


def _gloo_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "fhe-string-bounty_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ks = keyset(O.TOY_K1)
        plan = _plan("contains", 8, 4, world=world)
        inputs = np.concatenate([_enc(ks, b"abcabd", 8), _enc(ks, b"abd", 4)])  # deterministic per seed
        out = run_with_oracle(plan, inputs, ks.sk, rank, world)
        ret[rank] = int(ks.ck.decrypt_many(out)[0])
    finally:
        dist.destroy_process_group()


def test_sharded_runner_world2_gloo(toy_k1):
    """N>1 path on CPU: two ranks each compute half of every level, all-gather, agree on the result
    and agree with the single-rank run."""
    import torch.multiprocessing as mp
    world = 2
    single = _run(toy_k1, "contains", b"abcabd", b"abd")[0]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: single, 1: single} and single == 1


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
