"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical keys and inputs.

Bit-exact where the reference is integer-only (keyswitch, modulus switch, rotation, sample
extraction, linear ops, LUT generation); decrypt-level plus a torus-distance tolerance where the
reference itself is f64-FFT based (its own tests assert no more: fft64/math/fft/tests.rs:166-173,
fft64/crypto/tests.rs:5-13, algorithms/test/lwe_programmable_bootstrapping.rs:152-156)."""
import numpy as np
import pytest

import oracle as O
from conftest import gpu_engine, keyset, to_fhestr_params, torus_distance

pytestmark = pytest.mark.gpu

PARAM_SETS = [O.TOY_K1, O.TOY_K2, O.PARAM_MESSAGE_2_CARRY_2_KS_PBS]


def _phase_tolerance(p):
    """Bound on |phase_gpu - phase_oracle| (phase = b - <a, s>, the only quantity two f64 implementations
    can be compared on: one flipped decomposition digit re-randomises the mask coefficients while moving
    the phase by noise only).

    With a 2^23 base the f64 FFT's rounding (~2^40 per accumulator coefficient and CMUX step) is as large
    as the decomposition granule (2^41), so after a few of the n steps the two accumulators round
    differently and the two outputs are, in effect, two independent draws of the same PBS output noise:
    their phase difference has variance 2 V_pbs.  V_pbs is the engine's noise model (csrc/noise_model.h:
    GGSW noise + decomposition rounding + FFT rounding; its value is itself checked against the measured
    PBS output noise in tests/test_gpu_noise.py).  8 standard deviations of that difference:
    PARAM_MESSAGE_2_CARRY_2 -> 2^52.1 against 2^50.3 .. 2^51.0 measured over 48 samples, delta/2 = 2^58."""
    import fhestr
    from conftest import to_fhestr_params
    v_pbs = fhestr.noise_model(to_fhestr_params(p))["v_pbs"]
    return 8.0 * np.sqrt(2.0 * v_pbs) * 2.0**64


def _phases(ks, cts):
    return np.array([ks.ck.decrypt_plaintext(c) for c in cts], dtype=np.uint64)


@pytest.mark.parametrize("params", PARAM_SETS, ids=lambda p: p.name)
def test_keyswitch_bit_exact(params):
    ks = keyset(params)
    eng = gpu_engine(ks)
    rng = np.random.default_rng(1)
    B = 37  # ragged: not a multiple of the kernel's sample tile
    cts = rng.integers(0, 2**64, size=(B, params.big_size), dtype=np.uint64)
    cts[0, :] = 0
    cts[1, :-1] = 0  # trivial ciphertext
    cts[2, :] = 2**64 - 1
    got = eng.keyswitch(cts)
    want = np.stack([ks.sk.keyswitch(c) for c in cts])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ks_level,ks_base_log", [(22, 1), (17, 2), (11, 2), (16, 1)])
def test_keyswitch_with_many_levels_bit_exact(ks_level, ks_base_log):
    """Keyswitch decompositions of the compact-public-key parameter sets (shortint/parameters/parameters_compact_pk.rs:
    22 levels of base 2, 11 of base 4, ...) on a toy twin: more than 16 levels do not fit a 16-slot group of the matrix-core
    kernel and take the byte-plane kernel (they divided by zero at key load), up to 16 take the matrix cores with
    one mask element per group.  Bit-exact against the oracle, ragged batch; then one KS + PBS round trip."""
    import fhestr
    from conftest import to_fhestr_params
    p = O.Params(8, 1, 256, 15, 2, ks_base_log, ks_level, 4, 1, 1e-13, 1e-17, f"TOY_N256_KS{ks_level}x{ks_base_log}")
    ck = O.ClientKey(p, 0x5EED0077)
    sk = O.ServerKey(ck)
    eng = fhestr.Engine(to_fhestr_params(p), 0)
    try:
        eng.load_keys(sk.bsk, sk.ksk)
        rng = np.random.default_rng(3)
        cts = rng.integers(0, 2**64, size=(37, p.big_size), dtype=np.uint64)
        cts[0, :] = 0
        cts[1, :] = 2**64 - 1
        assert np.array_equal(eng.keyswitch(cts), np.stack([sk.keyswitch(c) for c in cts]))
        M = p.msg_mod * p.carry_mod
        lut, _ = eng.generate_lookup_table(lambda x: (x + 1) % M)
        enc = ck.encrypt_many(list(range(M)) * 3)
        out = eng.apply_lookup_table(enc, np.full(len(enc), lut, dtype=np.uint32))
        assert np.array_equal(ck.decrypt_many(out), np.array([(m + 1) % M for m in list(range(M)) * 3]))
    finally:
        eng.close()


@pytest.mark.parametrize("params", PARAM_SETS, ids=lambda p: p.name)
def test_lut_generation_matches_oracle(params):
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f = lambda x: (5 * x + 3) % M
    lut_id, deg = eng.generate_lookup_table(f)
    want, want_deg = ks.sk.generate_lookup_table(f)
    assert deg == want_deg
    assert np.array_equal(eng.download_lut(lut_id), want)


@pytest.mark.parametrize("params", PARAM_SETS, ids=lambda p: p.name)
def test_pbs_zero_mask_bit_exact(params):
    """a_i == 0 skips every CMUX (bootstrap.rs:281): only modulus switch, LUT rotation and sample
    extraction run -- all integer, so ciphertext bits must match the oracle exactly."""
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    lut, _ = ks.sk.generate_lookup_table(lambda x: (3 * x + 1) % M)
    lut_id = eng.upload_lut(lut)
    rng = np.random.default_rng(2)
    B = 19
    small = np.zeros((B, params.small_size), dtype=np.uint64)
    small[:, -1] = rng.integers(0, 2**64, size=B, dtype=np.uint64)
    small[0, -1] = 0
    small[1, -1] = 2**64 - 1          # modulus switch returns 2N
    small[2, -1] = 2**63
    got = eng.pbs(small, np.full(B, lut_id, dtype=np.uint32))
    want = np.stack([ks.sk.pbs(s, lut) for s in small])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("params", PARAM_SETS, ids=lambda p: p.name)
def test_ks_pbs_decrypts_and_tracks_oracle(params):
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f0 = lambda x: x
    f1 = lambda x: (x * x + 1) % M
    l0, d0 = ks.sk.generate_lookup_table(f0)
    l1, d1 = ks.sk.generate_lookup_table(f1)
    id0, id1 = eng.upload_lut(l0), eng.upload_lut(l1)
    reps = 3
    msgs = np.array([m for m in range(M)] * reps)
    rng = O.Rng(0x5EED0002, 9)
    cts = ks.ck.encrypt_many(msgs, rng)
    which = (np.arange(len(msgs)) // M) % 2
    idx = np.where(which == 0, id0, id1).astype(np.uint32)
    got = eng.apply_lookup_table(cts, idx)
    dec = ks.ck.decrypt_many(got)
    want_clear = np.array([f0(m) if w == 0 else f1(m) for m, w in zip(msgs, which)])
    assert np.array_equal(dec, want_clear)
    # phase-level: GPU f64 path vs the oracle's f64 path stay within the FFT tolerance
    luts = np.stack([l0, l1])
    want = ks.sk.apply_lookup_table_batch(cts, luts, which.astype(np.uint32))
    dist = torus_distance(_phases(ks, got), _phases(ks, want))
    print(f"{params.name}: max phase distance GPU vs oracle-fft = 2^{np.log2(dist.max() + 1):.1f}, "
          f"tolerance 2^{np.log2(_phase_tolerance(params)):.1f}, delta/2 = 2^{np.log2(params.delta / 2):.0f}")
    assert dist.max() < _phase_tolerance(params)
    assert _phase_tolerance(params) < params.delta / 4


@pytest.mark.parametrize("params", PARAM_SETS, ids=lambda p: p.name)
def test_ks_pbs_tracks_exact_integer_oracle(params):
    """Against the exact-integer external product (schoolbook negacyclic products mod 2^64, no FFT at all);
    on the real parameter set too (8 ciphertexts: ~2 s of CPU each)."""
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    lut, _ = ks.sk.generate_lookup_table(lambda x: (M - 1 - x))
    lut_id = eng.upload_lut(lut)
    msgs = list(range(M)) if params.N <= 256 else [0, 1, 5, 7, 8, 11, 14, 15]
    M = len(msgs)
    cts = ks.ck.encrypt_many(msgs, O.Rng(77, 1))
    got = eng.apply_lookup_table(cts, np.full(M, lut_id, dtype=np.uint32))
    want = ks.sk.apply_lookup_table_batch(cts, lut, exact=True)
    assert np.array_equal(ks.ck.decrypt_many(got), ks.ck.decrypt_many(want))
    dist = torus_distance(_phases(ks, got), _phases(ks, want))
    print(f"{params.name}: max phase distance GPU vs exact-integer oracle = 2^{np.log2(dist.max() + 1):.1f}")
    assert dist.max() < _phase_tolerance(params)


def test_lincomb_bit_exact():
    params = O.TOY_K1
    ks = keyset(params)
    eng = gpu_engine(ks)
    rng = np.random.default_rng(3)
    pool = rng.integers(0, 2**64, size=(11, params.big_size), dtype=np.uint64)
    jobs = [([(0, 1)], 0), ([(1, 4), (2, 1)], 0), ([(3, -1), (4, 1)], 5 * params.delta),
            ([], 7 * params.delta), ([(i, i - 5) for i in range(11)], 2**64 - 1)]
    got = eng.lincomb(pool, jobs)
    with np.errstate(over="ignore"):
        for row, (terms, c) in zip(got, jobs):
            want = np.zeros(params.big_size, dtype=np.uint64)
            for s, a in terms:
                want += pool[s] * np.uint64(a % 2**64)
            want[-1] += np.uint64(c % 2**64)
            assert np.array_equal(row, want)


def test_p22_batch_256_decrypts(p22):
    """BASELINE.json config 2: 256 independent LWEs, PARAM_MESSAGE_2_CARRY_2, per-LWE LUTs."""
    eng = gpu_engine(p22)
    M = 16
    rng = np.random.default_rng(0x5EED0002)
    tables = rng.integers(0, M, size=(16, M))
    ids = [eng.generate_lookup_table(lambda x, t=t: int(t[x]))[0] for t in tables]
    msgs = rng.integers(0, M, size=256)
    sel = rng.integers(0, 16, size=256)
    cts = p22.ck.encrypt_many(msgs, O.Rng(0x5EED0002, 1))
    got = eng.apply_lookup_table(cts, np.array([ids[s] for s in sel], dtype=np.uint32))
    dec = p22.ck.decrypt_many(got)
    assert np.array_equal(dec, tables[sel, msgs])


@pytest.mark.parametrize("selector", [2, 3, 4, 18, 19], ids=lambda s: f"variant{s}")
def test_p22_every_blind_rotate_variant(p22, selector):
    """All instantiated blind-rotation variants (points per thread 4/8/16, split and wide layouts)
    give decrypt-exact results and bit-exact zero-mask PBS on PARAM_MESSAGE_2_CARRY_2."""
    eng = gpu_engine(p22, selector)
    lut, _ = p22.sk.generate_lookup_table(lambda x: (x * 7 + 3) % 16)
    lut_id = eng.upload_lut(lut)
    msgs = np.arange(32) % 16
    cts = p22.ck.encrypt_many(msgs, O.Rng(123, selector))
    got = eng.apply_lookup_table(cts, np.full(32, lut_id, dtype=np.uint32))
    assert np.array_equal(p22.ck.decrypt_many(got), (msgs * 7 + 3) % 16)
    small = np.zeros((5, p22.params.small_size), dtype=np.uint64)
    small[:, -1] = np.array([0, 2**64 - 1, 2**63, 12345678901234567, 2**62 + 5], dtype=np.uint64)
    want = np.stack([p22.sk.pbs(s, lut) for s in small])
    assert np.array_equal(eng.pbs(small, np.full(5, lut_id, dtype=np.uint32)), want)


@pytest.mark.parametrize("params,selector", [(O.TOY_K1, 18), (O.TOY_K2, 18)], ids=["toy_k1_wide", "toy_k2_wide"])
def test_toy_wide_variants(params, selector):
    ks = keyset(params)
    eng = gpu_engine(ks, selector)
    M = params.msg_mod * params.carry_mod
    lut, _ = ks.sk.generate_lookup_table(lambda x: (M - 1 - x))
    lut_id = eng.upload_lut(lut)
    cts = ks.ck.encrypt_many(list(range(M)) * 2, O.Rng(5, 5))
    got = eng.apply_lookup_table(cts, np.full(2 * M, lut_id, dtype=np.uint32))
    assert np.array_equal(ks.ck.decrypt_many(got), np.array([M - 1 - m for m in range(M)] * 2))


@pytest.mark.parametrize("params", [O.TOY_N8192, O.TOY_N32768], ids=lambda p: p.name)
def test_large_polynomial_sizes(params):
    """N = 8192 / 32768 (geometry of PARAM_MESSAGE_3_CARRY_3 / _4_CARRY_4): accumulator and spectra
    live in an HBM workspace, four-step FFT (pbs_large_kernels.hip.h)."""
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f = lambda x: (3 * x + 1) % M
    lut, _ = ks.sk.generate_lookup_table(f)
    lut_id = eng.upload_lut(lut)
    # keyswitch + zero-mask PBS: bit-exact
    rng = np.random.default_rng(8)
    cts = rng.integers(0, 2**64, size=(5, params.big_size), dtype=np.uint64)
    assert np.array_equal(eng.keyswitch(cts), np.stack([ks.sk.keyswitch(c) for c in cts]))
    small = np.zeros((4, params.small_size), dtype=np.uint64)
    small[:, -1] = np.array([0, 2**64 - 1, 2**63, 0x0123456789ABCDEF], dtype=np.uint64)
    idx = np.full(4, lut_id, dtype=np.uint32)
    assert np.array_equal(eng.pbs(small, idx), np.stack([ks.sk.pbs(s, lut) for s in small]))
    # full KS+PBS: decrypt-exact, phase close to the oracle's f64 path
    msgs = np.array([0, 1, M // 2, M - 1, 5, 7])
    enc = ks.ck.encrypt_many(msgs, O.Rng(99, 3))
    got = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
    assert np.array_equal(ks.ck.decrypt_many(got), np.array([f(int(m)) for m in msgs]))
    want = ks.sk.apply_lookup_table_batch(enc, lut)
    dist = torus_distance(_phases(ks, got), _phases(ks, want))
    tol = _phase_tolerance(params)
    print(f"{params.name}: max phase distance GPU vs oracle-fft = 2^{np.log2(dist.max() + 1):.1f} (8 sigma = 2^{np.log2(tol):.1f})")
    assert dist.max() < tol
    # more LWEs than the GPU holds workgroups at once (per-LWE workspaces, cluster slots and mask tables are reused):
    # every output against the oracle, for both large-N kernels
    B = 2 * 256 + 3
    rng = np.random.default_rng(21)
    many = rng.integers(0, M, size=B)
    enc = ks.ck.encrypt_many(many, O.Rng(98, 5))
    want = ks.sk.apply_lookup_table_batch(enc, lut, threads=16)
    want_ph = _phases(ks, want)
    for mode in (1, 0):
        eng.set_cluster_mode(mode)
        try:
            got = eng.apply_lookup_table(enc, np.full(B, lut_id, dtype=np.uint32))
        finally:
            eng.set_cluster_mode(-1)
        assert np.array_equal(ks.ck.decrypt_many(got), np.array([f(int(m)) for m in many]))
        assert torus_distance(_phases(ks, got), want_ph).max() < tol


def test_p44_real_parameters_against_the_oracle_on_device_generated_keys():
    """PARAM_MESSAGE_4_CARRY_4_KS_PBS exactly as the reference defines it (shortint/parameters/mod.rs:1063-1077:
    n = 996, N = 32768, 2 PBS levels).  Server keys are generated on the device and exported; the CPU oracle
    (fft64/crypto/bootstrap.rs:242-364 restated, f64 path) then bootstraps the same ciphertexts with exactly that key
    material: decrypt equality and phases within 8 sigma of the noise model, for the cluster kernel (several CUs per LWE)
    and the one-workgroup-per-LWE kernel.  A layout error shared by device keygen and device PBS cannot cancel here: the
    oracle reads the exported keys in the reference's layouts."""
    import fhestr
    P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
    oP = O.Params(P.n, P.k, P.N, P.pbs_base_log, P.pbs_level, P.ks_base_log, P.ks_level, P.msg_mod, P.carry_mod,
                  P.lwe_std, P.glwe_std, P.name)
    ck = fhestr.ClientKey(P, 0x5EED0005)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    try:
        bsk, ksk = eng.generate_keys(g, s, 0x5EED0005, export=True)
        osk = O.ServerKey.from_keys(oP, bsk, ksk, threads=8)
        del bsk
        M = P.msg_mod * P.carry_mod
        f = lambda x: (x * x + 3) % M
        lut_id, _ = eng.generate_lookup_table(f)
        lut, _ = osk.generate_lookup_table(f)
        assert np.array_equal(eng.download_lut(lut_id), lut)
        msgs = np.array([0, 1, 255, 128, 77, 200, 16, 15])
        enc = ck.encrypt(msgs)
        assert np.array_equal(eng.keyswitch(enc[:3]), np.stack([osk.keyswitch(c) for c in enc[:3]]))      # integer path: bit-exact
        want = osk.apply_lookup_table_batch(enc[:4], lut, threads=4)
        big_sel = np.flatnonzero(g == 1)
        phase = lambda cts: (cts[:, -1] - cts[:, big_sel].sum(axis=1, dtype=np.uint64))
        tol = 8.0 * np.sqrt(2.0 * fhestr.noise_model(P)["v_pbs"]) * 2.0**64
        for mode in (1, 0):
            eng.set_cluster_mode(mode)
            out = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
            assert ck.decrypt(out).tolist() == [f(int(m)) for m in msgs]
            assert ck.decrypt(want).tolist() == [f(int(m)) for m in msgs[:4]]
            with np.errstate(over="ignore"):
                dist = torus_distance(phase(out[:4]), phase(want))
            print(f"P44 cluster mode {mode}: max phase distance GPU vs oracle = 2^{np.log2(dist.max() + 1):.1f} (8 sigma = 2^{np.log2(tol):.1f})")
            assert dist.max() < tol
        # zero-mask PBS on the real dimensions: bit-exact (LUT rotation + sample extraction, no CMUX)
        small = np.zeros((2, P.small_size), dtype=np.uint64)
        small[:, -1] = np.array([0x0123456789ABCDEF, 2**63 + 12345], dtype=np.uint64)
        for mode in (1, 0):
            eng.set_cluster_mode(mode)
            got = eng.pbs(small, np.full(2, lut_id, dtype=np.uint32))
            assert np.array_equal(got, np.stack([osk.pbs(c, lut) for c in small]))
    finally:
        eng.close()


@pytest.mark.parametrize("params", O.TOY_SHAPES, ids=lambda p: p.name)
def test_every_reference_parameter_shape(params):
    """One toy-n instance of every (N, k, level) shape in shortint/parameters/mod.rs that the engine
    instantiates: keyswitch + zero-mask PBS bit-exact, KS+PBS decrypt-exact."""
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f = lambda x: (M - 1 - x)
    lut, _ = ks.sk.generate_lookup_table(f)
    lut_id = eng.upload_lut(lut)
    rng = np.random.default_rng(11)
    cts = rng.integers(0, 2**64, size=(3, params.big_size), dtype=np.uint64)
    assert np.array_equal(eng.keyswitch(cts), np.stack([ks.sk.keyswitch(c) for c in cts]))
    small = np.zeros((3, params.small_size), dtype=np.uint64)
    small[:, -1] = np.array([0, 2**64 - 1, 0x0123456789ABCDEF], dtype=np.uint64)
    idx = np.full(3, lut_id, dtype=np.uint32)
    assert np.array_equal(eng.pbs(small, idx), np.stack([ks.sk.pbs(s, lut) for s in small]))
    msgs = np.array(sorted({0, 1, M // 2, M - 1}))
    enc = ks.ck.encrypt_many(msgs, O.Rng(7, 7))
    got = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
    assert np.array_equal(ks.ck.decrypt_many(got), np.array([f(int(m)) for m in msgs]))


def test_pbs_then_keyswitch_small_key_order(toy_k1):
    """*_PBS_KS parameter sets keep ciphertexts under the small key: bootstrap, then keyswitch
    (shortint/server_key/mod.rs:859-932).  Checked against the oracle's two stages."""
    import ctypes as C
    ks, params = toy_k1, toy_k1.params
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    f = lambda x: (x + 5) % M
    lut, _ = ks.sk.generate_lookup_table(f)
    lut_id = eng.upload_lut(lut)
    rng = O.Rng(321, 1)
    small = np.zeros((M, params.small_size), dtype=np.uint64)
    for m in range(M):   # encrypt under the small key with the small-key noise
        O.lib().orc_lwe_encrypt(ks.ck.small_sk, params.n, m * params.delta, params.lwe_std, rng.ptr, small[m])
    got = eng.apply_lookup_table_small_key(small, np.full(M, lut_id, dtype=np.uint32))
    dec = [int(O.lib().orc_decode(C.byref(params.c()), ks.ck.decrypt_small_plaintext(c))) for c in got]
    assert dec == [f(m) for m in range(M)]
    want = np.stack([ks.sk.keyswitch(ks.sk.pbs(s, lut)) for s in small])
    assert [int(O.lib().orc_decode(C.byref(params.c()), ks.ck.decrypt_small_plaintext(c))) for c in want] == dec


def test_p22_output_noise_matches_oracle(p22):
    """Statistical parity: the error of the bootstrapped phase (phase - delta*f(m)) has the same
    spread on the GPU as in the oracle's f64 and exact-integer paths, and sits far below delta/2."""
    eng = gpu_engine(p22)
    params = p22.params
    lut, _ = p22.sk.generate_lookup_table(lambda x: x)
    lut_id = eng.upload_lut(lut)
    rng = np.random.default_rng(42)
    msgs = rng.integers(0, 16, size=192)
    cts = p22.ck.encrypt_many(msgs, O.Rng(4242, 1))
    got = eng.apply_lookup_table(cts, np.full(len(msgs), lut_id, dtype=np.uint32))

    def errors(outs, ms):
        ph = _phases(p22, outs).astype(np.uint64)
        with np.errstate(over="ignore"):
            e = (ph - ms.astype(np.uint64) * np.uint64(params.delta)).astype(np.int64)
        return e.astype(np.float64)

    e_gpu = errors(got, msgs)
    e_fft = errors(p22.sk.apply_lookup_table_batch(cts[:64], lut), msgs[:64])
    assert np.abs(e_gpu).max() < params.delta / 8
    s_gpu, s_fft = e_gpu.std(), e_fft.std()
    print(f"PBS output noise std: GPU 2^{np.log2(s_gpu):.2f} (n=192), oracle-fft 2^{np.log2(s_fft):.2f} (n=64), "
          f"delta/2 = 2^{np.log2(params.delta / 2):.0f}")
    assert 0.6 < s_gpu / s_fft < 1.6
    assert abs(e_gpu.mean()) < 4 * s_gpu / np.sqrt(len(e_gpu)) + 1.0   # unbiased


@pytest.mark.parametrize("B", [1, 3, 255, 257, 600], ids=lambda b: f"B{b}")
def test_ragged_batch_sizes(toy_k1, B):
    """Batch sizes around the one-LWE-per-CU boundary (256 CUs) and the kernels' sample tiles; per-LWE
    LUT choice.  Keyswitch stays bit-exact, the PBS decrypt-exact."""
    ks = toy_k1
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    luts = [ks.sk.generate_lookup_table(f)[0] for f in (lambda x: x, lambda x: (x + 1) % M, lambda x: (3 * x) % M)]
    ids = np.array([eng.upload_lut(l) for l in luts], dtype=np.uint32)
    rng = np.random.default_rng(B)
    msgs = rng.integers(0, M, size=B)
    sel = rng.integers(0, 3, size=B)
    cts = ks.ck.encrypt_many(msgs, O.Rng(77, B))
    assert np.array_equal(eng.keyswitch(cts), np.stack([ks.sk.keyswitch(c) for c in cts]))
    got = ks.ck.decrypt_many(eng.apply_lookup_table(cts, ids[sel]))
    want = np.where(sel == 0, msgs, np.where(sel == 1, (msgs + 1) % M, (3 * msgs) % M))
    assert np.array_equal(got, want)


def test_empty_batches_and_error_reporting(toy_k1):
    """Empty inputs are no-ops; bad LUT ids, missing keys and missing LUTs are reported, not run."""
    import fhestr
    from conftest import to_fhestr_params
    ks = toy_k1
    eng = gpu_engine(ks)
    p = ks.params
    empty_big = np.zeros((0, p.big_size), dtype=np.uint64)
    assert eng.keyswitch(empty_big).shape == (0, p.small_size)
    assert eng.apply_lookup_table(empty_big, np.zeros(0, dtype=np.uint32)).shape == (0, p.big_size)
    assert eng.pbs(np.zeros((0, p.small_size), dtype=np.uint64), np.zeros(0, dtype=np.uint32)).shape == (0, p.big_size)
    cts = ks.ck.encrypt_many([1, 2])
    with pytest.raises(fhestr.FheError, match="lut_idx"):
        eng.apply_lookup_table(cts, np.array([0, 10**6], dtype=np.uint32))
    fresh = fhestr.Engine(to_fhestr_params(p), 0)
    try:
        with pytest.raises(fhestr.FheError, match="lookup table|keys"):
            fresh.apply_lookup_table(cts, np.zeros(2, dtype=np.uint32))
        fresh.generate_lookup_table(lambda x: x)
        with pytest.raises(fhestr.FheError, match="keys"):
            fresh.apply_lookup_table(cts, np.zeros(2, dtype=np.uint32))
    finally:
        fresh.close()


def test_p22_device_key_generation_matches_cpu_keys(p22):
    """fhe_engine_generate_keys on PARAM_MESSAGE_2_CARRY_2: same secret keys + seed as the CPU
    key set => bit-identical KSK / standard BSK (the oracle's), and a working engine."""
    import fhestr
    from conftest import to_fhestr_params
    eng = fhestr.Engine(to_fhestr_params(p22.params), 0)
    try:
        bsk, ksk = eng.generate_keys(p22.ck.glwe_sk, p22.ck.small_sk, p22.ck.seed, export=True)
        assert np.array_equal(ksk, p22.sk.ksk.ravel())
        assert np.array_equal(bsk, p22.sk.bsk.ravel())
        lut_id, _ = eng.generate_lookup_table(lambda x: (x * x) % 16)
        msgs = np.arange(64) % 16
        cts = p22.ck.encrypt_many(msgs, O.Rng(99, 1))
        got = eng.apply_lookup_table(cts, np.full(64, lut_id, dtype=np.uint32))
        assert np.array_equal(p22.ck.decrypt_many(got), (msgs * msgs) % 16)
    finally:
        eng.close()


@pytest.mark.gpu
def test_pipelined_calls_are_bit_identical_to_serial_ones(p22):
    """fhe_engine_set_pipeline: the keyswitch of call k+1 in the shadow of the blind rotation of call k.  Independent
    batches, a chain (output of one call is the input of the next), and one buffer used for every output: the same
    ciphertexts bit for bit as with serial calls."""
    import torch
    ks = p22
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    lut, _ = eng.generate_lookup_table(lambda x: (3 * x + 1) % M)
    rng = np.random.default_rng(17)
    B = 96
    batches = [torch.from_numpy(ks.ck.encrypt_many(rng.integers(0, M, size=B)).view(np.int64)).cuda() for _ in range(5)]
    idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")

    def run(pipelined):
        eng.set_pipeline(pipelined)
        outs = [torch.zeros_like(b) for b in batches]
        for b, o in zip(batches, outs):                                    # independent
            eng.apply_lookup_table_dev(b.data_ptr(), idx.data_ptr(), o.data_ptr(), B)
        chain = [torch.zeros_like(batches[0]) for _ in range(4)]
        src = batches[0]
        for c in chain:                                                    # chained: input = previous output
            eng.apply_lookup_table_dev(src.data_ptr(), idx.data_ptr(), c.data_ptr(), B)
            src = c
        same = torch.zeros_like(batches[0])
        for b in batches:                                                  # every output into one buffer
            eng.apply_lookup_table_dev(b.data_ptr(), idx.data_ptr(), same.data_ptr(), B)
        eng.synchronize()
        eng.set_pipeline(False)
        return [t.cpu().numpy() for t in outs + chain + [same]]

    serial, piped = run(False), run(True)
    assert all(np.array_equal(a, b) for a, b in zip(serial, piped))
    # and a serial call right after pipelined ones still sees a free small-ciphertext buffer
    got = eng.apply_lookup_table(batches[1].cpu().numpy().view(np.uint64), np.full(B, lut, dtype=np.uint32))
    assert np.array_equal(got.view(np.int64), serial[1])


@pytest.mark.gpu
def test_overlapped_batches_mode_matches_the_large_batch_kernel(p22):
    """fhe_engine_set_pipeline(2): consecutive 96-LWE calls alternate between two streams on the two-LWEs-per-CU kernel.
    Independent batches, a chain (every call reads the previous call's output), one buffer written by every call and a
    call that overwrites the previous call's input: bit-identical to the same calls made one after the other on that
    kernel (variant selector 16 | 2), decrypt-identical to the default serial path."""
    import torch
    ks = p22
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    lut, _ = eng.generate_lookup_table(lambda x: (7 * x + 1) % M)
    rng = np.random.default_rng(29)
    B = 96
    batches = [torch.from_numpy(ks.ck.encrypt_many(rng.integers(0, M, size=B)).view(np.int64)).cuda() for _ in range(6)]
    idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")

    def run(mode):
        eng.set_pipeline(mode)
        outs = [torch.zeros_like(b) for b in batches]
        for b, o in zip(batches, outs):
            eng.apply_lookup_table_dev(b.data_ptr(), idx.data_ptr(), o.data_ptr(), B)
        chain = [torch.zeros_like(batches[0]) for _ in range(5)]
        src = batches[0]
        for c in chain:
            eng.apply_lookup_table_dev(src.data_ptr(), idx.data_ptr(), c.data_ptr(), B)
            src = c
        same = torch.zeros_like(batches[0])
        for b in batches:
            eng.apply_lookup_table_dev(b.data_ptr(), idx.data_ptr(), same.data_ptr(), B)
        scratch = batches[1].clone()        # call k reads scratch, call k+1 overwrites it
        war_out = torch.zeros_like(scratch)
        eng.apply_lookup_table_dev(scratch.data_ptr(), idx.data_ptr(), war_out.data_ptr(), B)
        eng.apply_lookup_table_dev(batches[2].data_ptr(), idx.data_ptr(), scratch.data_ptr(), B)
        eng.synchronize()
        eng.set_pipeline(0)
        return [t.cpu().numpy() for t in outs + chain + [same, war_out, scratch]]

    default_serial = run(0)
    overlapped = run(2)
    eng.set_variant(16 | 2)                 # every serial call on the two-LWEs-per-CU kernel
    try:
        wide_serial = run(0)
    finally:
        eng.set_variant(0)
    assert all(np.array_equal(a, b) for a, b in zip(wide_serial, overlapped))
    dec = lambda arrs: [ks.ck.decrypt_many(a.view(np.uint64)) for a in arrs]
    assert all(np.array_equal(a, b) for a, b in zip(dec(default_serial), dec(overlapped)))


@pytest.mark.gpu
def test_overlapped_mode_three_interleaved_dependent_chains(p22):
    """mode 2 used to remember only the LAST call of each stream.  Three chains interleaved call by call
    (A0 B0 C0 A1 B1 C1 ...) put chain A's calls on streams 0, 1, 0, 1 ...: A1 (stream 1) reads what A0 (stream 0) wrote three
    calls earlier, by which time stream 0's record described C0 -- the wait was skipped.  Every call of the run is remembered now.  Bit-identical to the same calls made serially on the same kernel, several times over."""
    import torch
    ks = p22
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    lut, _ = eng.generate_lookup_table(lambda x: (5 * x + 2) % M)
    rng = np.random.default_rng(31)
    B = 96
    heads = [torch.from_numpy(ks.ck.encrypt_many(rng.integers(0, M, size=B)).view(np.int64)).cuda() for _ in range(3)]
    idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")
    depth = 4

    def run(mode):
        eng.set_pipeline(mode)
        bufs = [[torch.zeros_like(h) for _ in range(depth)] for h in heads]
        for d in range(depth):
            for c in range(3):
                src = heads[c] if d == 0 else bufs[c][d - 1]
                eng.apply_lookup_table_dev(src.data_ptr(), idx.data_ptr(), bufs[c][d].data_ptr(), B)
        eng.synchronize()
        eng.set_pipeline(0)
        return [b.cpu().numpy() for chain in bufs for b in chain]

    eng.set_variant(16 | 2)                 # the serial reference on the kernel mode 2 uses
    try:
        want = run(0)
        for _ in range(5):
            got = run(2)
            assert all(np.array_equal(a, b) for a, b in zip(want, got))
    finally:
        eng.set_variant(0)
    f = lambda x: (5 * x + 2) % M
    msgs = ks.ck.decrypt_many(heads[0].cpu().numpy().view(np.uint64))
    for d in range(depth):
        msgs = np.array([f(int(m)) for m in msgs])
        assert np.array_equal(ks.ck.decrypt_many(want[d].view(np.uint64)), msgs)


@pytest.mark.gpu
def test_pipelined_call_right_after_a_serial_one_and_stream_ordered_inputs(p22):
    """(a) with the throughput mode on, a batch that takes the serial path (more LWEs than CUs) followed at
    once, without synchronisation, by one that takes the pipelined path -- both use the engine's small-ciphertext buffer;
    (b) an input produced by work enqueued on the engine stream between two pipelined calls, handed over with
    fhe_engine_pipeline_input_event.  Bit-identical to serial calls."""
    import torch
    ks = p22
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    lut, _ = eng.generate_lookup_table(lambda x: (5 * x + 2) % M)
    rng = np.random.default_rng(23)
    big_b, small_b = 512, 96
    big = torch.from_numpy(ks.ck.encrypt_many(rng.integers(0, M, size=big_b)).view(np.int64)).cuda()
    small = torch.from_numpy(ks.ck.encrypt_many(rng.integers(0, M, size=small_b)).view(np.int64)).cuda()
    idx = torch.full((big_b,), int(lut), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def run(pipelined):
        eng.set_pipeline(pipelined)
        o_big, o_small, o_late = torch.zeros_like(big), torch.zeros_like(small), torch.zeros_like(small)
        for _ in range(3):           # serial-path batch, then at once a pipelined one
            eng.apply_lookup_table_dev(big.data_ptr(), idx.data_ptr(), o_big.data_ptr(), big_b)
            eng.apply_lookup_table_dev(small.data_ptr(), idx.data_ptr(), o_small.data_ptr(), small_b)
        # (b) the input of the next call is written by copies on the engine stream, enqueued after the previous call
        stream = torch.cuda.ExternalStream(eng.stream)
        staged = torch.zeros_like(small)
        ballast_src = torch.zeros(64 << 20, dtype=torch.int64, device="cuda")      # 512 MB: keeps the stream busy for a while
        ballast_dst = torch.empty_like(ballast_src)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for _ in range(4):
                ballast_dst.copy_(ballast_src, non_blocking=True)
            staged.copy_(small, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        if pipelined:
            eng.pipeline_input_event(ev.cuda_event)
        eng.apply_lookup_table_dev(staged.data_ptr(), idx.data_ptr(), o_late.data_ptr(), small_b)
        eng.synchronize()
        eng.set_pipeline(False)
        return [t.cpu().numpy() for t in (o_big, o_small, o_late)]

    serial, piped = run(False), run(True)
    assert all(np.array_equal(a, b) for a, b in zip(serial, piped))
    assert np.array_equal(serial[1], serial[2])
    assert np.array_equal(ks.ck.decrypt_many(piped[1].view(np.uint64))[:8], ks.ck.decrypt_many(serial[2].view(np.uint64))[:8])


@pytest.mark.gpu
def test_host_threads_share_one_engine(toy_k1):
    """The reference's ServerKey is Sync (shortint/engine/mod.rs:23-25,184-189): several host threads may bootstrap with the
    same key.  Here every C ABI entry point takes the engine's lock: four threads hammer one engine (host-buffer calls,
    their own tables, a plan each) and every result decrypts right."""
    import threading
    import fhestr
    ks = toy_k1
    eng = gpu_engine(ks)
    p = ks.params
    M = p.msg_mod * p.carry_mod
    errors = []

    def worker(t):
        try:
            f = lambda x, t=t: (x + t) % M
            lut, _ = eng.generate_lookup_table(f)
            rng = np.random.default_rng(100 + t)
            for it in range(6):
                msgs = rng.integers(0, M, size=9 + t)
                cts = ks.ck.encrypt_many(msgs, O.Rng(55 + t, it))
                out = eng.apply_lookup_table(cts, np.full(len(msgs), lut, dtype=np.uint32))
                if not np.array_equal(ks.ck.decrypt_many(out), np.array([f(int(m)) for m in msgs])):
                    errors.append((t, it, "lut"))
                assert np.array_equal(eng.keyswitch(cts), np.stack([ks.sk.keyswitch(c) for c in cts]))
            plan = fhestr.Plan.string_op(eng, "eq", 4, 4)
            a = ks.ck.encrypt_many(fhestr.string_to_blocks(to_fhestr_params(p), b"ab%dz" % t, 4), O.Rng(77 + t, 1))
            res = plan.run(np.concatenate([a, a]))
            if int(ks.ck.decrypt_many(res)[0]) != 1:
                errors.append((t, "eq"))
            plan.close()
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.gpu
def test_blind_rotation_against_an_exact_integer_recurrence():
    """Known-answer test of the rotate-subtract-decompose-accumulate chain (polynomial_algorithms.rs:315-354,425-490,
    decomposer.rs:98-118, bootstrap.rs:242-331) that needs neither the oracle nor its FFT: with an all-zero GLWE
    secret, noiseless GGSWs are just `bit * q / B` at one coefficient, and the blind rotation of the body polynomial is
        B <- LUT * X^-ms(b);   B <- B + s_i * closest_representable(B * X^ms(a_i) - B)    for every a_i != 0
    in exact wrapping integers.  The GPU's f64 path may differ by its transform rounding (a few 2^13 per step) and,
    rarely, by one decomposition digit (2^41) where that rounding crosses a digit boundary: any indexing or sign
    mistake in the rotation would be off by ~2^60."""
    import fhestr
    P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
    n, N, k, bl = P.n, P.N, P.k, P.pbs_base_log
    rng = np.random.default_rng(0xA5)
    s = rng.integers(0, 2, size=n, dtype=np.uint64)
    bsk = np.zeros((n, 1, k + 1, k + 1, N), dtype=np.uint64)
    bsk[:, 0, k, k, 0] = s << np.uint64(64 - bl)            # last row, body polynomial, coefficient 0 (ggsw_encryption.rs:122-126)
    eng = fhestr.Engine(P, 0)
    try:
        eng.load_keys(bsk.reshape(-1), np.zeros(P.ksk_len, dtype=np.uint64))
        lut = np.zeros((k + 1, N), dtype=np.uint64)
        lut[k] = rng.integers(0, 1 << 63, size=N, dtype=np.uint64) * np.uint64(2)
        lut_id = eng.upload_lut(lut.reshape(-1))
        B = 48
        cts = rng.integers(0, 1 << 63, size=(B, n + 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(B, n + 1), dtype=np.uint64)
        cts[:, 5] = 0                                        # an a_i == 0 is skipped (bootstrap.rs:281)
        got = eng.pbs(cts, np.full(B, lut_id, dtype=np.uint32))

        def ms(x):                                           # fast_pbs_modulus_switch, common.rs:26-43
            return ((int(x) >> (64 - 11 - 2)) + 1) >> 1
        def monomial_mul(p, d):                              # p * X^d mod X^N + 1, d in [0, 2N]
            d %= 2 * N
            out = np.roll(p, d % N).copy()
            out[: d % N] = np.uint64(0) - out[: d % N]
            return (np.uint64(0) - out) if d >= N else out
        def closest(x):                                      # decomposer.rs:105-117 for one level of bl bits
            sh = np.uint64(63 - bl)
            return (((x >> sh) + np.uint64(1)) & ~np.uint64(1)) << sh
        worst, off = 0, 0
        with np.errstate(over="ignore"):
            for c, out in zip(cts, got):
                body = monomial_mul(lut[k], 2 * N - ms(c[n]))
                for i in range(n):
                    if c[i] == 0 or s[i] == 0:
                        continue
                    body = body + closest(monomial_mul(body, ms(c[i])) - body)
                assert not out[:N].any()                     # the mask polynomial stays zero: sample extraction of zeros
                d = int(out[N]) - int(body[0])
                d = abs((d + (1 << 63)) % (1 << 64) - (1 << 63))
                worst = max(worst, d)
                off += d > 1 << 26
        assert worst < 1 << 46 and off <= B // 4, (worst, off)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("params", [O.TOY_K2] + [p for p in O.TOY_SHAPES if p.N in (512, 1024, 4096)], ids=lambda p: p.name)
def test_two_workgroups_per_cu_kernels_of_every_shape(params):
    """Batches of 2 and 4 LWEs per CU (+ 3) on every shape that has a two-LWEs-per-CU kernel or takes more than one round
    of its only kernel: the time-sliced priorities of blind_rotate_wide_kernel (BlindRotateArgs::fair_shift; a ticket per
    hardware CU, parity through LDS) are on for these counts.  Decrypt-exact, per-LWE tables, and the first LWEs agree
    with a small launch of the same ciphertexts within the noise model's 8 sigma."""
    import fhestr
    ks = keyset(params)
    eng = gpu_engine(ks)
    M = params.msg_mod * params.carry_mod
    fs = [lambda x: (M - 1 - x), lambda x: (x + 1) % M]
    ids = np.array([eng.upload_lut(ks.sk.generate_lookup_table(f)[0]) for f in fs], dtype=np.uint32)
    for B in (2 * 256 + 3, 4 * 256 + 3):
        rng = np.random.default_rng(B)
        msgs = rng.integers(0, M, size=B)
        sel = rng.integers(0, 2, size=B)
        cts = ks.ck.encrypt_many(msgs, O.Rng(91, B))
        got = eng.apply_lookup_table(cts, ids[sel])
        want = np.where(sel == 0, M - 1 - msgs, (msgs + 1) % M)
        assert np.array_equal(ks.ck.decrypt_many(got), want)
        few = eng.apply_lookup_table(cts[:5], ids[sel[:5]])
        tol = 8.0 * np.sqrt(2.0 * fhestr.noise_model(to_fhestr_params(params))["v_pbs"]) * 2.0**64
        phase = lambda c: np.array([ks.ck.decrypt_plaintext(x) for x in c], dtype=np.uint64)
        assert torus_distance(phase(got[:5]), phase(few)).max() < tol


@pytest.mark.gpu
def test_keep_busy_replicas_change_nothing(p22):
    """fhe_engine_set_keep_busy: launches of at most half the CUs' worth of LWEs carry replica workgroups that store
    nothing -- outputs are bit-identical to the plain launch (same kernel, same inputs), for 1, 3, 35 and 128 LWEs, and
    a 129-LWE launch (no room for replicas) is untouched."""
    ks = p22
    eng = gpu_engine(ks)
    M = ks.params.msg_mod * ks.params.carry_mod
    lut, _ = ks.sk.generate_lookup_table(lambda x: (x + 5) % M)
    lut_id = eng.upload_lut(lut)
    rng = np.random.default_rng(3)
    try:
        for B in (1, 3, 35, 128, 129):
            msgs = rng.integers(0, M, size=B)
            cts = ks.ck.encrypt_many(msgs, O.Rng(41, B))
            idx = np.full(B, lut_id, dtype=np.uint32)
            eng.set_keep_busy(False)
            plain = eng.apply_lookup_table(cts, idx)
            eng.set_keep_busy(True)
            busy = eng.apply_lookup_table(cts, idx)
            assert np.array_equal(plain, busy), B
            assert np.array_equal(ks.ck.decrypt_many(busy), (msgs + 5) % M)
    finally:
        eng.set_keep_busy(False)
