"""Radix-integer operations (integer/server_key/radix_parallel/{add,scalar_add,cmux}.rs,
comparator.rs): plans built by the C++ layer, executed by the CPU oracle (and on the GPU), against Python
integers -- the reference's own notion of correctness (radix_parallel/tests_unsigned.rs: decrypt(op(enc)) ==
clear op, incl. wrap-around and the forced a == b case of tests_cases_comparisons.rs:33-39)."""
import numpy as np
import pytest

import oracle as O
from conftest import keyset, to_fhestr_params
from plan_oracle import run_with_oracle

N = 4                    # 4 blocks of 2 bits: 8-bit integers (the reference's FheUint8 shape)
MOD = 1 << (2 * N)
PAIRS = [(0, 0), (255, 1), (1, 255), (170, 85), (200, 100), (15, 240), (128, 128), (3, 252), (77, 77)]


def _run(ks, op, operands, scalar=0, n=N):
    import fhestr
    P = to_fhestr_params(ks.params)
    plan = fhestr.Plan.integer_op(None, op, n, scalar, params=P)
    inputs = np.concatenate([ks.ck.encrypt_many(np.atleast_1d(o)) for o in operands])
    return ks.ck.decrypt_many(run_with_oracle(plan, inputs, ks.sk)), plan


@pytest.mark.parametrize("a,b", PAIRS)
def test_add_sub_and_scalar_forms(toy_k1, a, b):
    import fhestr
    P = to_fhestr_params(toy_k1.params)
    ea, eb = fhestr.int_to_blocks(P, a, N), fhestr.int_to_blocks(P, b, N)
    for op, want in (("add", (a + b) % MOD), ("sub", (a - b) % MOD)):
        out, plan = _run(toy_k1, op, [ea, eb])
        assert fhestr.blocks_to_int(P, out) == want and all(o < 4 for o in out), (op, a, b)   # carries are empty
        out, _ = _run(toy_k1, "scalar_" + op, [ea], scalar=b)
        assert fhestr.blocks_to_int(P, out) == want, ("scalar_" + op, a, b)
    # depth: state + ceil(log2 n) scan steps + extract (add.rs:572-624)
    assert plan.info()["n_levels"] == 1 + 2 + 1


@pytest.mark.parametrize("a,b", PAIRS)
def test_comparisons(toy_k1, a, b):
    import fhestr
    P = to_fhestr_params(toy_k1.params)
    ea, eb = fhestr.int_to_blocks(P, a, N), fhestr.int_to_blocks(P, b, N)
    for op, want in (("eq", a == b), ("ne", a != b), ("gt", a > b), ("ge", a >= b), ("lt", a < b), ("le", a <= b)):
        assert _run(toy_k1, op, [ea, eb])[0][0] == int(want), (op, a, b)
        assert _run(toy_k1, "scalar_" + op, [ea], scalar=b)[0][0] == int(want), ("scalar_" + op, a, b)


def test_cmux_and_extracts(toy_k1):
    import fhestr
    P = to_fhestr_params(toy_k1.params)
    t, f = fhestr.int_to_blocks(P, 0xA7, N), fhestr.int_to_blocks(P, 0x3C, N)
    for cond, want in ((1, 0xA7), (0, 0x3C)):
        out, plan = _run(toy_k1, "cmux", [np.array([cond]), t, f])
        assert fhestr.blocks_to_int(P, out) == want
    assert plan.info()["n_pbs"] == 3 * N and plan.info()["n_levels"] == 2    # cmux.rs: 2 x N zero-outs + N extracts
    full = np.array([15, 9, 4, 3, 0], dtype=np.uint64)                      # blocks with carries
    assert _run(toy_k1, "message_extract", [full], n=5)[0].tolist() == [3, 1, 0, 3, 0]
    assert _run(toy_k1, "carry_extract", [full], n=5)[0].tolist() == [3, 2, 1, 0, 0]


def test_odd_block_counts_and_errors(toy_k1):
    import fhestr
    P = to_fhestr_params(toy_k1.params)
    a, b = 0b10_11_01, 0b10_01_11               # 3 blocks: a trailing unpacked block in the comparator
    ea, eb = fhestr.int_to_blocks(P, a, 3), fhestr.int_to_blocks(P, b, 3)
    assert _run(toy_k1, "gt", [ea, eb], n=3)[0][0] == int(a > b)
    assert _run(toy_k1, "eq", [ea, ea], n=3)[0][0] == 1
    assert fhestr.blocks_to_int(P, _run(toy_k1, "add", [ea, eb], n=3)[0]) == (a + b) % 64
    with pytest.raises(fhestr.FheError, match="unknown integer op"):
        fhestr.Plan.integer_op(None, "mul", 4, params=P)
    with pytest.raises(fhestr.FheError, match="64 bits"):
        fhestr.Plan.integer_op(None, "scalar_add", 40, 1, params=P)


@pytest.mark.gpu
def test_integer_ops_on_the_gpu_p22(p22):
    """FheUint8-shaped operations on the real parameter set through the GPU plan executor."""
    import fhestr
    from conftest import gpu_engine
    eng = gpu_engine(p22)
    P = eng.params
    rng = np.random.default_rng(0x1234)
    enc = lambda v: p22.ck.encrypt_many(fhestr.int_to_blocks(P, v, N))
    for _ in range(4):
        a, b = int(rng.integers(0, 256)), int(rng.integers(0, 256))
        for op, want in (("add", (a + b) % 256), ("sub", (a - b) % 256)):
            out = p22.ck.decrypt_many(fhestr.Plan.integer_op(eng, op, N).run(np.concatenate([enc(a), enc(b)])))
            assert fhestr.blocks_to_int(P, out) == want
        out = p22.ck.decrypt_many(fhestr.Plan.integer_op(eng, "scalar_add", N, b).run(enc(a)))
        assert fhestr.blocks_to_int(P, out) == (a + b) % 256
        for op, want in (("gt", a > b), ("eq", a == b), ("le", a <= b)):
            out = p22.ck.decrypt_many(fhestr.Plan.integer_op(eng, op, N).run(np.concatenate([enc(a), enc(b)])))
            assert out[0] == int(want)
        out = p22.ck.decrypt_many(fhestr.Plan.integer_op(eng, "cmux", N).run(
            np.concatenate([p22.ck.encrypt_many([a & 1]), enc(a), enc(b)])))
        assert fhestr.blocks_to_int(P, out) == (a if a & 1 else b)
