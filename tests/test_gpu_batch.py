"""Many independent instances of one plan in a single pass (fhe_plan_run_batch / fhe_str_op_many): level l of all instances
is one gather + keyswitch + blind-rotation launch.  The reference's throughput shape -- many independent inputs per call
(benches/core_crypto/pbs_bench.rs:430-549; rayon over the blocks of an integer, integer/server_key/radix_parallel/
comparison.rs:10-33) -- applied to whole FheString operations.  Every instance must decrypt like the single-instance run
(fhe_plan_run), like the same plan stepped through the CPU oracle, and like Python's bytes semantics; ragged counts."""
import numpy as np
import pytest

import oracle as O
from conftest import gpu_engine, to_fhestr_params
from plan_oracle import run_with_oracle

pytestmark = pytest.mark.gpu

ROWS = [b"hello wd", b"hello", b"", b"xhellox", b"hell", b"HELLO wd", b"o wd", b"hello wd", b"llo", b"wd hello", b"abcdefgh",
        b"hello wd", b"h", b"ell", b"hello  d", b"hellohel", b"d"]


def _enc(ks, P, s, cap):
    import fhestr
    return ks.ck.encrypt_many(fhestr.string_to_blocks(P, s, cap))


@pytest.mark.parametrize("count", [1, 3, 17])
@pytest.mark.parametrize("op,pat", [("eq", b"hello wd"), ("contains", b"ell"), ("find", b"o"), ("starts_with", b"hel")])
def test_plan_batch_matches_single_runs_and_the_oracle(toy_k1, op, pat, count):
    import fhestr
    ks = toy_k1
    eng = gpu_engine(ks)
    P = to_fhestr_params(ks.params)
    a_cap, b_cap = 8, len(pat) + 1            # the pattern is zero padded by one character
    plan = fhestr.Plan.string_op(eng, op, a_cap, b_cap)
    offline = fhestr.Plan.string_op(None, op, a_cap, b_cap, None, 1, params=P)
    pat_ct = _enc(ks, P, pat, b_cap)
    rows = ROWS[:count]
    inputs = np.stack([np.concatenate([_enc(ks, P, r, a_cap), pat_ct]) for r in rows])
    out = plan.run_batch(inputs)
    assert out.shape[0] == count
    for i, r in enumerate(rows):
        got = ks.ck.decrypt_many(out[i]).tolist()
        single = ks.ck.decrypt_many(plan.run(inputs[i])).tolist()
        assert got == single, (op, r)
        if i < 4:                             # the same plan stepped through the CPU oracle (a few instances: it is slow)
            assert got == ks.ck.decrypt_many(run_with_oracle(offline, inputs[i], ks.sk)).tolist(), (op, r)
        if op == "eq":
            assert got == [int(r == pat)]
        elif op == "contains":
            assert got == [int(pat in r)]
        elif op == "starts_with":
            assert got == [int(r.startswith(pat))]
        else:
            idx = r.find(pat)
            assert got[0] == int(idx >= 0) and (idx < 0 or sum(d * P.msg_mod**k for k, d in enumerate(got[1:])) == idx)


def test_plan_batch_on_device_arrays(toy_k1):
    """fhe_plan_run_batch_dev: inputs and outputs resident in HBM, ordered on the engine's stream."""
    import fhestr
    import torch
    ks = toy_k1
    eng = gpu_engine(ks)
    P = to_fhestr_params(ks.params)
    plan = fhestr.Plan.string_op(eng, "to_lower", 8)
    rows = ROWS[:9]
    inputs = np.stack([_enc(ks, P, r, 8) for r in rows])
    d_in = torch.from_numpy(inputs.view(np.int64)).cuda()
    n_out = plan.info()["n_outputs"]
    d_out = torch.zeros((len(rows), n_out, P.big_size), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    plan.run_batch_dev(d_in.data_ptr(), d_out.data_ptr(), len(rows))
    eng.synchronize()
    out = d_out.cpu().numpy().view(np.uint64)
    for i, r in enumerate(rows):
        assert fhestr.blocks_to_string(P, ks.ck.decrypt_many(out[i])) == r.lower()


def test_one_pattern_against_many_rows_p22(p22):
    """FheStringOps.eq_many / contains_many / op_many on PARAM_MESSAGE_2_CARRY_2: 13 rows of 16 characters against one
    encrypted pattern and against clear bytes; unary operation over many rows."""
    import fhestr
    ks = p22
    eng = gpu_engine(ks)
    P = to_fhestr_params(ks.params)
    ops = fhestr.FheStringOps(eng)
    rng = np.random.default_rng(5)
    base = b"the quick brown "
    rows_b = [base, base[:-1] + b"!", b"", base[:7], b"THE QUICK BROWN ", base] + [bytes(rng.integers(97, 123, size=16).astype(np.uint8)) for _ in range(7)]
    rows = np.stack([_enc(ks, P, r, 16) for r in rows_b])
    pat = _enc(ks, P, base, 16)
    assert ks.ck.decrypt_many(ops.eq_many(rows, pat)).tolist() == [int(r == base) for r in rows_b]
    assert ks.ck.decrypt_many(ops.ne_many(rows, base)).tolist() == [int(r != base) for r in rows_b]
    needle = _enc(ks, P, b"quick", 6)
    assert ks.ck.decrypt_many(ops.contains_many(rows, needle)).tolist() == [int(b"quick" in r) for r in rows_b]
    lowered = ops.op_many("to_lower", rows)
    for i, r in enumerate(rows_b):
        assert fhestr.blocks_to_string(P, ks.ck.decrypt_many(lowered[i])) == r.lower()
    # the single-row entry point gives the same answers
    assert ks.ck.decrypt_many(np.stack([ops.eq(rows[i], pat) for i in range(3)])).tolist() == [int(r == base) for r in rows_b[:3]]


def test_batch_argument_errors(toy_k1):
    import fhestr
    eng = gpu_engine(toy_k1)
    P = to_fhestr_params(toy_k1.params)
    plan = fhestr.Plan.string_op(eng, "eq", 4, 4)
    with pytest.raises(fhestr.FheError):
        plan.run_batch(np.zeros((2, 3, P.big_size), dtype=np.uint64))          # wrong inputs per instance
    sharded = fhestr.Plan.string_op(eng, "eq", 4, 4, None, 2)
    n_in = sharded.info()["n_inputs"]
    with pytest.raises(fhestr.FheError, match="world = 1"):
        sharded.run_batch(np.zeros((2, n_in, P.big_size), dtype=np.uint64))
    assert plan.run_batch(np.zeros((0, plan.info()["n_inputs"], P.big_size), dtype=np.uint64)).shape[0] == 0
