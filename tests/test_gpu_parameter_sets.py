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
