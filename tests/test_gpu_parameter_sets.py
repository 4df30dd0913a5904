"""Every shortint parameter set the reference defines (tests/golden/reference_parameter_sets.json: the constants of
shortint/parameters/mod.rs and multi_bit.rs, read by tests/golden/make_param_table.py) through the engine with
its real dimensions: server keys generated on the device, every message through a random lookup table, decrypted.
The reference holds no ciphertext vectors, so this is decrypt-level: parity of the arithmetic itself
is pinned on the oracle in test_gpu_parity.py / test_multibit.py."""
import json
import os

import numpy as np
import pytest

TABLE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_parameter_sets.json")))


def _params(name):
    import fhestr
    r = TABLE[name]
    return fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"],
                         r["ks_base_log"], r["ks_level"], r["message_modulus"], r["carry_modulus"],
                         r["lwe_modular_std_dev"], r["glwe_modular_std_dev"], name, r.get("grouping_factor", 0))


def test_table_is_the_whole_reference_list():
    kinds = {"ks_pbs": 0, "pbs_ks": 0, "multi_bit": 0}
    for name, r in TABLE.items():
        kinds["multi_bit" if "grouping_factor" in r else "pbs_ks" if r["encryption_key_choice"] == "Small" else "ks_pbs"] += 1
    assert kinds == {"ks_pbs": 36, "pbs_ks": 4, "multi_bit": 6}
    r = TABLE["PARAM_MESSAGE_2_CARRY_2_KS_PBS"]            # shortint/parameters/mod.rs:703-717
    assert (r["lwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["ks_level"]) == (742, 2048, 23, 5)


def _fields(a):
    return (a.n, a.k, a.N, a.pbs_base_log, a.pbs_level, a.ks_base_log, a.ks_level, a.msg_mod, a.carry_mod, a.lwe_std,
            a.glwe_std, max(a.grouping, 1))


def test_product_parameter_constants_match_the_table():
    import fhestr
    seen = 0
    for name in TABLE:
        if hasattr(fhestr, name):
            assert _fields(getattr(fhestr, name)) == _fields(_params(name)), name
            seen += 1
    assert seen >= 9


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TABLE))
def test_reference_parameter_set_decrypts(name):
    import fhestr
    P = _params(name)
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, 0x5E7)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    try:
        eng.generate_keys(g, s, 0x5E7)
        rng = np.random.default_rng(11)
        table = rng.integers(0, M, size=M)
        lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
        msgs = np.arange(max(M, 8)) % M if M <= 64 else rng.integers(0, M, size=64)
        cts = ck.encrypt(msgs)
        idx = np.full(len(msgs), lut, dtype=np.uint32)
        if TABLE[name]["encryption_key_choice"] == "Small":
            # PBS -> KS order on small-key ciphertexts (shortint/server_key/mod.rs:859-932); the client encrypts under
            # the big key, so: keyswitch in, PBS+KS with the table, identity PBS out to a big-key ciphertext
            ident, _ = eng.generate_lookup_table(lambda x: x)
            small = eng.apply_lookup_table_small_key(eng.keyswitch(cts), idx)
            out = eng.pbs(small, np.full(len(msgs), ident, dtype=np.uint32))
        else:
            out = eng.apply_lookup_table(cts, idx)
        assert np.array_equal(ck.decrypt(out), table[msgs])
    finally:
        eng.close()


# One parameter set per blind-rotation kernel family (csrc/engine.hip, variants()): N = 256 (k = 5), 512 (k = 3 and k = 2 with
# two levels), 1024 (k = 2), 2048, 4096 (one and two levels), 8192 (seq kernel, one and two levels), 16384 and 32768 (cluster
# kernel, two and three levels).
FAMILIES = ["PARAM_MESSAGE_1_CARRY_0_KS_PBS", "PARAM_MESSAGE_1_CARRY_1_KS_PBS", "PARAM_MESSAGE_2_CARRY_0_KS_PBS",
            "PARAM_MESSAGE_2_CARRY_1_KS_PBS", "PARAM_MESSAGE_2_CARRY_2_KS_PBS", "PARAM_MESSAGE_2_CARRY_3_KS_PBS",
            "PARAM_MESSAGE_1_CARRY_4_KS_PBS", "PARAM_MESSAGE_5_CARRY_1_KS_PBS", "PARAM_MESSAGE_3_CARRY_3_KS_PBS",
            "PARAM_MESSAGE_3_CARRY_4_KS_PBS", "PARAM_MESSAGE_1_CARRY_6_KS_PBS", "PARAM_MESSAGE_3_CARRY_5_KS_PBS"]


def _oracle_params(P):
    import oracle as O
    return O.Params(P.n, P.k, P.N, P.pbs_base_log, P.pbs_level, P.ks_base_log, P.ks_level, P.msg_mod, P.carry_mod,
                    P.lwe_std, P.glwe_std, P.name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FAMILIES)
def test_kernel_family_against_the_oracle_on_exported_device_keys(name):
    """Real dimensions, device-generated keys EXPORTED in the reference's layouts and handed to the CPU
    oracle: the oracle's keyswitch of the same ciphertexts is bit-identical, its KS + PBS decrypts to the same messages and
    its output phases sit within 8 sigma (noise model) of the device's.  Decrypt-only checks on device keys would let a
    layout error shared by device keygen and device PBS cancel; this cannot."""
    import fhestr
    import oracle as O
    from conftest import torus_distance
    P = _params(name)
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, 0x5E8)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    try:
        bsk, ksk = eng.generate_keys(g, s, 0x5E8, export=True)
        osk = O.ServerKey.from_keys(_oracle_params(P), bsk, ksk, threads=8)
        f = lambda x: (5 * x + 1) % M
        lut_id, _ = eng.generate_lookup_table(f)
        lut, _ = osk.generate_lookup_table(f)
        msgs = np.array([0, M - 1, M // 2, 1])
        enc = ck.encrypt(msgs)
        assert np.array_equal(eng.keyswitch(enc), np.stack([osk.keyswitch(c) for c in enc]))
        got = eng.apply_lookup_table(enc, np.full(len(msgs), lut_id, dtype=np.uint32))
        want = osk.apply_lookup_table_batch(enc, lut, threads=4)
        assert ck.decrypt(got).tolist() == [f(int(m)) for m in msgs] == ck.decrypt(want).tolist()
        big_sel = np.flatnonzero(g == 1)
        phase = lambda cts: (cts[:, -1] - cts[:, big_sel].sum(axis=1, dtype=np.uint64))
        tol = 8.0 * np.sqrt(2.0 * fhestr.noise_model(P)["v_pbs"]) * 2.0**64
        with np.errstate(over="ignore"):
            dist = torus_distance(phase(got), phase(want))
        print(f"{name}: max phase distance GPU vs oracle = 2^{np.log2(dist.max() + 1):.1f} (8 sigma = 2^{np.log2(tol):.1f})")
        assert dist.max() < tol
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["PARAM_MESSAGE_2_CARRY_1_KS_PBS", "PARAM_MESSAGE_2_CARRY_3_KS_PBS"])
def test_device_keygen_is_bit_identical_to_the_oracle_keygen(name):
    """N = 1024 (k = 2) and N = 4096 with their real dimensions: the device-generated KSK and standard-domain BSK equal
    the oracle's (same secret keys, same ChaCha20 streams) word for word -- as test_gpu_parity.py asserts for P22."""
    import fhestr
    import oracle as O
    P = _params(name)
    ock = O.ClientKey(_oracle_params(P), 0x5E9)
    osk = O.ServerKey(ock, threads=16, fourier=False)
    eng = fhestr.Engine(P, 0)
    try:
        bsk, ksk = eng.generate_keys(ock.glwe_sk, ock.small_sk, 0x5E9, export=True)
        assert np.array_equal(ksk, osk.ksk) and np.array_equal(bsk, osk.bsk)
    finally:
        eng.close()


@pytest.mark.gpu
def test_dense_kernel_of_n1024_against_the_oracle_and_the_two_per_cu_kernel():
    """N = 1024, k = 2 (PARAM_MESSAGE_2_CARRY_1_KS_PBS): beyond two LWEs per CU the engine takes the dense blind-rotation
    kernel (one exchange-plane set, four workgroups per CU, FftSwap9; pbs_dense_kernels.hip.h).  Same algorithm, another
    order of the transforms' roundings: the 800 outputs of one call decrypt to the table and their PHASES sit within 8 sigma
    of the difference of two PBS noises (noise model) of those of the same ciphertexts sent in calls of 400 (two-per-CU
    kernel) -- the bar of the oracle test above.  Neither words nor noise samples can be compared between two kernels: one
    rounding that moves one decomposition digit in one of the 742 steps adds a whole key row to the accumulator, a fresh
    mask, and every later digit -- hence the noise sample -- differs (measured with scripts/dense_diff.py: every word
    differs, phase distance median 2^48.6 = sigma, between the two older kernels as well).
    The CPU oracle, handed the exported device keys, bootstraps eight of the 800 ciphertexts (the first and last of the
    launch among them): same messages, phases within the same bar of the dense kernel's."""
    import fhestr
    import torch
    import oracle as O
    from conftest import torus_distance
    P = _params("PARAM_MESSAGE_2_CARRY_1_KS_PBS")
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, 0x5EA)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    try:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        bsk, ksk = eng.generate_keys(g, s, 0x5EA, export=True)
        rng = np.random.default_rng(12)
        table = rng.integers(0, M, size=M)
        lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
        B = 3 * cus + 32                       # three per CU and a ragged tail
        half = B // 2
        assert B > 2 * cus and cus < half <= 2 * cus
        msgs = rng.integers(0, M, size=B)
        cts = ck.encrypt(msgs)
        idx = np.full(B, lut, dtype=np.uint32)
        dense = eng.apply_lookup_table(cts, idx)
        assert np.array_equal(ck.decrypt(dense), table[msgs])
        halves = np.concatenate([eng.apply_lookup_table(cts[:half], idx[:half]), eng.apply_lookup_table(cts[half:], idx[half:])])
        assert np.array_equal(ck.decrypt(halves), table[msgs])
        big_sel = np.flatnonzero(g == 1)
        phase = lambda cts: (cts[:, -1] - cts[:, big_sel].sum(axis=1, dtype=np.uint64))
        with np.errstate(over="ignore"):
            dist = torus_distance(phase(dense), phase(halves)).max()
        tol = 8.0 * np.sqrt(2.0 * fhestr.noise_model(P)["v_pbs"]) * 2.0**64
        print(f"dense vs two-per-CU kernel: max phase distance 2^{np.log2(dist + 1):.1f}, 8 sigma = 2^{np.log2(tol):.1f}")
        assert dist < tol
        small = eng.apply_lookup_table(cts[:64], idx[:64])
        assert np.array_equal(ck.decrypt(small), table[msgs[:64]])
        # the oracle on the same keys and ciphertexts
        osk = O.ServerKey.from_keys(_oracle_params(P), bsk, ksk, threads=8)
        olut, _ = osk.generate_lookup_table(lambda x: int(table[x]))
        pick = np.array([0, 1, cus, 2 * cus + 1, 3 * cus - 1, 3 * cus, B - 2, B - 1])
        want = osk.apply_lookup_table_batch(cts[pick], olut, threads=8)
        assert np.array_equal(ck.decrypt(want), table[msgs[pick]])
        with np.errstate(over="ignore"):
            odist = torus_distance(phase(dense[pick]), phase(want)).max()
        print(f"dense kernel vs oracle: max phase distance 2^{np.log2(odist + 1):.1f}")
        assert odist < tol
    finally:
        eng.close()
