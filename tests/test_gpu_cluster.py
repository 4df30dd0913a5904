"""GPU parity of blind_rotate_cluster_kernel (csrc/pbs_cluster_kernels.hip.h): several compute units of one XCD
per LWE for N >= 16384, against the CPU oracle and against the one-workgroup-per-LWE kernel on the same keys.

Reference path: fft64/crypto/bootstrap.rs:242-364, ggsw.rs:477-598; parameters shortint/parameters/mod.rs:1063-1077
(N = 32768) and the N = 16384 family.  Bit-exact where the path is integer-only (zero-mask PBS: rotation, sample
extraction), decrypt-exact and within the noise model's 8 sigma of the oracle's f64 path otherwise."""
import numpy as np
import pytest

import oracle as O
from conftest import gpu_engine, keyset, torus_distance, to_fhestr_params

pytestmark = pytest.mark.gpu
import os as _os
ROOT_DIR = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))

SHAPES = [O.TOY_N32768] + [p for p in O.TOY_SHAPES if p.N >= 16384]
# (shape, fhe_engine_set_cluster_mode): 1 = the best multi-CU kernel of the shape -- blind_rotate_xcd_kernel (all CUs of an
# XCD per LWE, csrc/pbs_xcd_kernels.hip.h) for N = 32768 with two levels, blind_rotate_cluster_kernel otherwise; 2 = the
# 8-CU cluster kernel also where the whole-XCD kernel exists
CASES = [(p, 1) for p in SHAPES] + [(O.TOY_N32768, 2)]
CASE_IDS = [f"{p.name}-mode{m}" for p, m in CASES]


def _phases(ks, cts):
    return np.array([ks.ck.decrypt_plaintext(c) for c in cts], dtype=np.uint64)


def _tolerance(p):
    import fhestr
    v_pbs = fhestr.noise_model(to_fhestr_params(p))["v_pbs"]
    return 8.0 * np.sqrt(2.0 * v_pbs) * 2.0**64


@pytest.mark.parametrize("params,mode", CASES, ids=CASE_IDS)
def test_cluster_kernel_against_oracle(params, mode):
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f = lambda x: (3 * x + 1) % M
    lut, _ = ks.sk.generate_lookup_table(f)
    lut_id = eng.upload_lut(lut)
    eng.set_cluster_mode(mode)
    try:
        # zero-mask PBS (no CMUX step runs: LUT rotation, hand-over of the published accumulator, sample extraction)
        small = np.zeros((4, params.small_size), dtype=np.uint64)
        small[:, -1] = np.array([0, 2**64 - 1, 2**63, 0x0123456789ABCDEF], dtype=np.uint64)
        idx = np.full(4, lut_id, dtype=np.uint32)
        assert np.array_equal(eng.pbs(small, idx), np.stack([ks.sk.pbs(s, lut) for s in small]))
        clusters = eng.cluster_info()
        print(f"{params.name}: {clusters} clusters formed")
        assert clusters >= 1
        # full KS + PBS: decrypt-exact, phase within 8 sigma of the oracle's f64 path
        msgs = np.array([0, 1, M // 2, M - 1, 5, 7])
        enc = ks.ck.encrypt_many(msgs, O.Rng(99, 3))
        got = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
        assert np.array_equal(ks.ck.decrypt_many(got), np.array([f(int(m)) for m in msgs]))
        want = ks.sk.apply_lookup_table_batch(enc, lut)
        dist = torus_distance(_phases(ks, got), _phases(ks, want))
        tol = _tolerance(params)
        print(f"{params.name}: max phase distance cluster kernel vs oracle = 2^{np.log2(dist.max() + 1):.1f} (8 sigma = 2^{np.log2(tol):.1f})")
        assert dist.max() < tol
        # the one-workgroup-per-LWE kernel on the same inputs: same messages, phases as close
        eng.set_cluster_mode(0)
        ref = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
        assert np.array_equal(ks.ck.decrypt_many(ref), ks.ck.decrypt_many(got))
        assert torus_distance(_phases(ks, got), _phases(ks, ref)).max() < tol
    finally:
        eng.set_cluster_mode(-1)


@pytest.mark.parametrize("params,mode", [(O.TOY_N32768, 1), (O.TOY_N32768, 2), ([p for p in O.TOY_SHAPES if p.N == 16384][0], 1)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_cluster_kernel_more_lwes_than_clusters(params, mode):
    """Every cluster walks several LWEs (epoch flags, workspace and mask table reused), ragged count, two tables."""
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    fs = [lambda x: (3 * x + 1) % M, lambda x: (M - 1 - x)]
    ids = [eng.upload_lut(ks.sk.generate_lookup_table(f)[0]) for f in fs]
    eng.set_cluster_mode(mode)
    try:
        B = 2 * 64 + 3
        rng = np.random.default_rng(17)
        msgs = rng.integers(0, M, size=B)
        which = rng.integers(0, 2, size=B)
        enc = ks.ck.encrypt_many(msgs, O.Rng(123, 4))
        got = eng.apply_lookup_table(enc, np.array([ids[w] for w in which], dtype=np.uint32))
        want = np.array([fs[w](int(m)) for m, w in zip(msgs, which)])
        assert np.array_equal(ks.ck.decrypt_many(got), want)
        # a_i == 0 is skipped by the whole cluster (bootstrap.rs:281): zero out some mask elements after the keyswitch
        small = eng.keyswitch(enc[:5])
        small[:, 1] = 0
        idx = np.full(5, ids[0], dtype=np.uint32)
        a = eng.pbs(small, idx)
        eng.set_cluster_mode(0)
        b = eng.pbs(small, idx)
        assert np.array_equal(ks.ck.decrypt_many(a), ks.ck.decrypt_many(b))
    finally:
        eng.set_cluster_mode(-1)


TESTHOOKS_LIB = _os.path.join(ROOT_DIR, "build", "testhooks", "libfhestr_testhooks.so")


@pytest.mark.parametrize("mode", [1, 2])
def test_a_missing_hand_over_is_recovered_or_reported_never_hung(mode):
    """Fault injection: one workgroup of one cluster never publishes one of its epoch flags.  The waits of that cluster
    give up after the (lowered) poll limit and the launch drains.  Default: the engine notices before the call returns
    and runs the batch again on the one-workgroup kernel -- correct results, `cluster_fallbacks` counts it.  With
    FHESTR_CLUSTER_FALLBACK=0 (round 3's behaviour) the next host-visible completion point returns an error and the engine
    is usable again afterwards.  A hand-over that never comes must never hang the GPU nor hand out a wrong ciphertext.
    The hook (FHESTR_CLUSTER_TEST_FAULT) exists only in the -DFHESTR_TEST_HOOKS build of the library
    (`make -C fhe-string-bounty_amd testhooks`, built by __graft_entry__.build()); the product library ignores it."""
    import os
    import subprocess
    import sys
    assert os.path.exists(TESTHOOKS_LIB), "build/testhooks/libfhestr_testhooks.so missing: run __graft_entry__.build()"
    code = r'''
import sys, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd"); sys.path.insert(0, ".")
import fhestr, oracle as O
sys.path.insert(0, "tests")
from conftest import to_fhestr_params
p = O.TOY_N32768
ck = O.ClientKey(p, 0x5EED0001); sk = O.ServerKey(ck)
eng = fhestr.Engine(to_fhestr_params(p), 0); eng.load_keys(sk.bsk, sk.ksk)
M = p.msg_mod * p.carry_mod
lut, _ = eng.generate_lookup_table(lambda x: (x + 1) % M)
cts = ck.encrypt_many([1, 2, 3], O.Rng(1, 1))
eng.set_cluster_mode(int(sys.argv[1]))
try:
    out = eng.apply_lookup_table(cts, np.full(3, lut, dtype=np.uint32))
    print("NO ERROR", "decrypts", [int(ck.decrypt(c)) for c in out], "fallbacks", eng.cluster_fallbacks())
except fhestr.FheError as e:
    print("ERROR:", e)
'''
    base = {k: v for k, v in os.environ.items() if k not in ("FHESTR_LIB", "FHESTR_CLUSTER_TEST_FAULT")}
    hooks = dict(base, FHESTR_LIB=TESTHOOKS_LIB, FHESTR_CLUSTER_TEST_FAULT="5", FHESTR_CLUSTER_SPIN_LIMIT="4096")
    r = subprocess.run([sys.executable, "-c", code, str(mode)], env=hooks, cwd=ROOT_DIR, capture_output=True, text=True, timeout=300)
    assert "NO ERROR decrypts [2, 3, 4] fallbacks 1" in r.stdout, r.stdout + r.stderr[-1000:]
    r = subprocess.run([sys.executable, "-c", code, str(mode)], env=dict(hooks, FHESTR_CLUSTER_FALLBACK="0"), cwd=ROOT_DIR,
                       capture_output=True, text=True, timeout=300)
    assert "ERROR:" in r.stdout and "hand-over timed out" in r.stdout, r.stdout + r.stderr[-1000:]
    # the product library has no such hook: the same environment runs clean, nothing to recover from
    prod = dict(base, FHESTR_CLUSTER_TEST_FAULT="5", FHESTR_CLUSTER_SPIN_LIMIT="4096")
    r = subprocess.run([sys.executable, "-c", code, str(mode)], env=prod, cwd=ROOT_DIR, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NO ERROR decrypts [2, 3, 4] fallbacks 0" in r.stdout, r.stdout + r.stderr[-1000:]
    # and the test build without the fault works too
    hooks.pop("FHESTR_CLUSTER_TEST_FAULT")
    r = subprocess.run([sys.executable, "-c", code, str(mode)], env=hooks, cwd=ROOT_DIR, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NO ERROR decrypts [2, 3, 4] fallbacks 0" in r.stdout, r.stdout + r.stderr[-1000:]


@pytest.mark.parametrize("shape,mode", [(O.TOY_N32768, 1), (O.TOY_N32768, 2), ([p for p in O.TOY_SHAPES if p.N == 16384][0], 1)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_cluster_launch_while_another_engine_holds_the_compute_units(p22, shape, mode):
    """the multi-CU kernels need their whole grid resident at once.  Here a second engine keeps every CU
    busy with long blind_rotate_wide_kernel batches (2,048 LWEs of PARAM_MESSAGE_2_CARRY_2, about 16 ms each, two workgroups per
    CU) while cluster batches are launched from another stream at varying offsets into them.  Every cluster result must be
    correct -- its workgroups queue until CUs free up, the formation wait is bounded but long enough, and a launch that gave
    up all the same is run again on the one-workgroup kernel before the call returns (cluster_settle; counted) -- a wrong
    ciphertext without an error is the one outcome that may not happen, and since round 4 an error is none either.
    The other engine's results are checked too."""
    import time
    import fhestr
    import torch
    eng_a = gpu_engine(p22)
    ks_b = keyset(shape)
    eng_b = gpu_engine(ks_b)
    pa = p22.params
    Ma = pa.msg_mod * pa.carry_mod
    lut_a, _ = eng_a.generate_lookup_table(lambda x: (x + 3) % Ma)
    rng = np.random.default_rng(71)
    msgs_a = rng.integers(0, Ma, size=2048)
    d_in = torch.from_numpy(p22.ck.encrypt_many(msgs_a, O.Rng(71, 1)).view(np.int64)).cuda()
    d_idx = torch.full((2048,), int(lut_a), dtype=torch.int32, device="cuda")
    d_out = torch.zeros_like(d_in)
    Mb = shape.msg_mod * shape.carry_mod
    fb = lambda x: (5 * x + 1) % Mb
    lut_b, _ = eng_b.generate_lookup_table(fb)
    torch.cuda.synchronize()
    eng_b.set_cluster_mode(mode)
    timeouts = 0
    fallbacks_before = eng_b.cluster_fallbacks()
    try:
        for trial in range(8):
            count = (3, 8, 17, 40)[trial % 4]
            msgs_b = rng.integers(0, Mb, size=count)
            enc_b = ks_b.ck.encrypt_many(msgs_b, O.Rng(72, trial))
            for _ in range(3):                           # ~50 ms of wide-kernel work queued on the other engine's stream
                eng_a.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), 2048)
            time.sleep(0.003 * (trial % 3))              # land at different points of the running batch
            try:
                got = eng_b.apply_lookup_table(enc_b, np.full(count, lut_b, dtype=np.uint32))
                assert np.array_equal(ks_b.ck.decrypt_many(got), np.array([fb(int(m)) for m in msgs_b])), f"trial {trial}: silent garbage"
            except fhestr.FheError as e:
                assert "hand-over timed out" in str(e), str(e)
                timeouts += 1
            eng_a.synchronize()
            assert np.array_equal(p22.ck.decrypt_many(d_out.cpu().numpy().view(np.uint64)[:64]), (msgs_a[:64] + 3) % Ma)
    finally:
        eng_b.set_cluster_mode(-1)
    print(f"{shape.name} mode {mode}: {timeouts} of 8 launches reported a hand-over time-out, "
          f"{eng_b.cluster_fallbacks() - fallbacks_before} were re-run on the one-workgroup kernel, all results handed out were correct")
    assert timeouts == 0       # since round 4 a launch that gave up is re-run before the call returns (cluster_settle)
    # after the contention: a clean launch works
    enc_b = ks_b.ck.encrypt_many([1, 2, 3], O.Rng(73, 1))
    got = eng_b.apply_lookup_table(enc_b, np.full(3, lut_b, dtype=np.uint32))
    assert np.array_equal(ks_b.ck.decrypt_many(got), np.array([fb(m) for m in (1, 2, 3)]))
