"""Every parameter constant of the reference's shortint layer against fhe_params_supported (no device needed): the
46 sets of shortint/parameters/{mod,multi_bit}.rs must be accepted; the 56 "experimental" compact-public-key sets of
parameters_compact_pk.rs (other keyswitch decompositions: up to 22 levels, bases up to 2^25; N up to 65536) must be
accepted or refused with a reason -- never crash the host (ks_level > 16 divided by zero at key load)."""
import json
import os

import pytest

import fhestr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params(name, r):
    return fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"],
                         r["ks_base_log"], r["ks_level"], r["message_modulus"], r["carry_modulus"], r["lwe_modular_std_dev"],
                         r["glwe_modular_std_dev"], name, r.get("grouping_factor", 1))


def _table(fname):
    return json.load(open(os.path.join(GOLDEN, fname)))


def test_every_classic_and_multi_bit_set_is_supported():
    for name, r in _table("reference_parameter_sets.json").items():
        ok, why = fhestr.params_supported(_params(name, r))
        assert ok, f"{name}: {why}"


def test_compact_pk_sets_are_supported_or_cleanly_refused():
    table = _table("reference_parameter_sets_compact_pk.json")
    assert len(table) == 56
    supported, refused = [], {}
    for name, r in table.items():
        ok, why = fhestr.params_supported(_params(name, r))
        if ok:
            supported.append(name)
        else:
            assert why, name
            refused[name] = why
    # refusals are of two kinds only: a keyswitch base above 2^7 (digits no longer fit a byte), or a blind-rotation shape
    # the library has no kernel for (N = 65536; three levels at N = 1024 / 8192)
    for name, why in refused.items():
        assert "keyswitch decomposition" in why or "no blind-rotation kernel" in why, (name, why)
    # the set ADVICE r3 names: N = 16384, two PBS levels, 22 keyswitch levels of base 2 -- byte-plane keyswitch kernel
    assert "PARAM_MESSAGE_3_CARRY_4_COMPACT_PK_PBS_KS" in supported
    many_levels = [n for n in supported if table[n]["ks_level"] > 16]
    assert many_levels, "no accepted set exercises ks_level > 16"
    print(f"compact-pk sets: {len(supported)} accepted, {len(refused)} refused")
    assert len(supported) >= 30


@pytest.mark.parametrize("ks_level,ks_base_log", [(22, 1), (17, 2), (62, 1)])
def test_many_keyswitch_levels_are_accepted(ks_level, ks_base_log):
    p = fhestr.Params(742, 1, 2048, 23, 1, ks_base_log, ks_level, 4, 4, 7e-6, 3e-16, "P22_MANY_KS_LEVELS")
    ok, why = fhestr.params_supported(p)
    assert ok, why


def test_bad_keyswitch_decompositions_are_refused():
    for ks_level, ks_base_log in [(0, 3), (5, 0), (2, 8), (32, 2)]:
        ok, why = fhestr.params_supported(fhestr.Params(742, 1, 2048, 23, 1, ks_base_log, ks_level, 4, 4, 7e-6, 3e-16, "BAD"))
        assert not ok and "keyswitch" in why
