"""BASELINE.json config 5 inside `pytest -m gpu`: FheString::to_lower + replace on a 1024-char string
under PARAM_MESSAGE_4_CARRY_4_KS_PBS exactly as the reference defines it (shortint/parameters/mod.rs:
1063-1077: n = 996, k = 1, N = 32768, PBS base 2^15 x 2 levels, KS base 2^3 x 7 levels; one ASCII char =
2 blocks of 4 bits).  Server keys (3.7 GB) are generated on the device (fhe_engine_generate_keys);
correctness = decrypt(op(enc(s))) == Python's bytes op (the reference's notion,
integer/server_key/radix_parallel/tests_cases_comparisons.rs:33-39).  The same two plans are also
stepped through the CPU oracle on the toy-n instance of the N = 32768 shape."""
import time

import numpy as np
import pytest

import oracle as O
from conftest import gpu_engine, keyset
from plan_oracle import run_with_oracle

pytestmark = pytest.mark.gpu

WORDS = [b"The ", b"quick ", b"BROWN ", b"fox ", b"Jumps ", b"over ", b"the ", b"LAZY ", b"dog. "]


def _text(n_chars, seed=0x5EED0005):
    rng = np.random.default_rng(seed)
    return b"".join(WORDS[int(i)] for i in rng.integers(0, len(WORDS), size=n_chars))[:n_chars]


@pytest.fixture(scope="module")
def p44():
    import fhestr
    P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
    ck = fhestr.ClientKey(P, 0x5EED0005)
    eng = fhestr.Engine(P, 0)
    glwe_sk, small_sk = ck.secret_keys()
    t0 = time.time()
    eng.generate_keys(glwe_sk, small_sk, 0x5EED0005)
    print(f"P44 device keygen: {time.time() - t0:.1f} s")
    yield P, ck, eng
    eng.close()


def test_config5_to_lower_1024_chars(p44):
    import fhestr
    P, ck, eng = p44
    ops = fhestr.FheStringOps(eng)
    s = _text(1000)
    assert any(65 <= c <= 90 for c in s)
    es = ck.encrypt(fhestr.string_to_blocks(P, s, 1024))
    t0 = time.time()
    out = ops.to_lower(es)
    print(f"config 5 to_lower, 1024 chars, P44: {time.time() - t0:.2f} s")
    assert fhestr.blocks_to_string(P, ck.decrypt(out)) == s.lower()


def test_config5_replace_1024_chars(p44):
    import fhestr
    P, ck, eng = p44
    ops = fhestr.FheStringOps(eng)
    s = _text(1000)
    assert s.count(b"the ") >= 3
    es = ck.encrypt(fhestr.string_to_blocks(P, s, 1024))
    t0 = time.time()
    out = ops.replace(es, b"the ", b"THAT")
    print(f"config 5 replace('the ' -> 'THAT'), 1024 chars, P44: {time.time() - t0:.2f} s")
    assert fhestr.blocks_to_string(P, ck.decrypt(out)) == s.replace(b"the ", b"THAT")


@pytest.mark.parametrize("op,clear", [("to_lower", None), ("replace_clear", b"the THAT")])
def test_config5_plans_match_oracle_on_toy_n(op, clear):
    """Same plan, same inputs, N = 32768 / 4-bit blocks with a toy LWE dimension: the GPU executor against
    the oracle stepping through the exported levels."""
    import fhestr
    ks = keyset(O.TOY_N32768)
    eng = gpu_engine(ks)
    s = b"Over the Hills "
    plan = fhestr.Plan.string_op(eng, op, 16, 0, clear)
    inputs = ks.ck.encrypt_many(fhestr.string_to_blocks(eng.params, s, 16))
    got = ks.ck.decrypt_many(plan.run(inputs))
    want = ks.ck.decrypt_many(run_with_oracle(plan, inputs, ks.sk))
    assert np.array_equal(got, want)
    clear_want = s.lower() if op == "to_lower" else s.replace(b"the ", b"THAT")
    assert fhestr.blocks_to_string(eng.params, got) == clear_want
