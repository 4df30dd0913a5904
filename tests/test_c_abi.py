"""The C-ABI library builds, loads and exports every symbol include/fhestr.h declares; the client
side (CPU) agrees bit-for-bit with the oracle's harness; GPU entry points fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest

import oracle as O
from conftest import ROOT, to_fhestr_params


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fhestr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", text))
    for n in re.findall(r"^FHE_STR_BINARY_DECL\((\w+)\)", text, flags=re.M):   # macro-declared pairs
        names |= {f"fhe_str_{n}", f"fhe_str_{n}_clear"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import fhestr
    lib = fhestr.lib()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fhestr.h but not exported"
    assert set(fhestr.EXPORTS) <= set(names)


def test_client_keys_match_oracle_harness():
    import fhestr
    p = O.TOY_K1
    ck = fhestr.ClientKey(to_fhestr_params(p), 1234)
    bsk, ksk = ck.gen_server_keys(2)
    ock = O.ClientKey(p, 1234)
    osk = O.ServerKey(ock, fourier=False)
    g, s = ck.secret_keys()
    assert np.array_equal(g, ock.glwe_sk) and np.array_equal(s, ock.small_sk)
    assert np.array_equal(bsk, osk.bsk)
    assert np.array_equal(ksk, osk.ksk)
    msgs = np.arange(p.msg_mod * p.carry_mod)
    cts = ck.encrypt(msgs)
    assert np.array_equal(ck.decrypt(cts), msgs)
    assert np.array_equal(ock.decrypt_many(cts), msgs)          # oracle decrypts product ciphertexts
    assert np.array_equal(ck.decrypt(ock.encrypt_many(msgs)), msgs)  # and vice versa


def test_engine_fails_loudly_without_gpu():
    import torch
    import fhestr
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fhestr.FheError):
        fhestr.Engine(fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, 0)


def test_engine_rejects_unsupported_parameters():
    import fhestr
    bad = fhestr.Params(10, 1, 100, 23, 1, 3, 5, 4, 4, 1e-9, 1e-12, "bad N")
    with pytest.raises(fhestr.FheError):
        fhestr.Engine(bad, 0)


def test_null_pointers_and_bad_handles_return_errors():
    """Reference convention (c_api/utils.rs:3-28): never crash across the ABI, return non-zero and
    leave a message."""
    import ctypes as C
    import fhestr
    L = fhestr.lib()
    out = C.c_void_p(123)
    assert L.fhe_engine_create(None, 0, C.byref(out)) != 0 and not out.value        # out nulled first
    assert b"null pointer" in L.fhe_last_error()
    assert L.fhe_engine_create(C.byref(fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS.c()), 0, None) != 0
    assert L.fhe_engine_load_keys(None, None, None) != 0
    assert L.fhe_ks_pbs_batch(None, None, None, None, 0) != 0
    assert L.fhe_plan_info(None, None) != 0
    assert L.fhe_engine_destroy(None) == 0 and L.fhe_plan_destroy(None) == 0     # destroying nothing is fine
    ck = C.c_void_p()
    assert L.fhe_client_key_create(None, 1, C.byref(ck)) != 0


def test_plan_builder_rejects_misuse():
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    plan = fhestr.Plan(None, params=P)
    a = plan.input(3)
    with pytest.raises(fhestr.FheError):
        plan.pbs(a, 99)                       # LUT id not created through this plan
    with pytest.raises(fhestr.FheError):
        plan.lin([(12345, 1)])                # unknown node
    ident = plan.lut(lambda x: x)
    out = plan.pbs(a, ident)
    plan.output(out)
    with pytest.raises(fhestr.FheError):
        plan.info()                           # not finalised yet
    plan.finalize(1)
    assert plan.info()["n_pbs"] == 1
    with pytest.raises(fhestr.FheError):
        plan.input(3)                         # building after finalize
    with pytest.raises(fhestr.FheError):
        fhestr.Plan.string_op(None, "no_such_op", 4, params=P)
    with pytest.raises(fhestr.FheError):
        fhestr.Plan.string_op(None, "replace_clear", 4, clear=b"abc", params=P)   # odd length: from/to differ


def test_trivial_inputs_fold_at_build_time():
    """apply_lookup_table on a trivial ciphertext is a clear table lookup (shortint/server_key/mod.rs:763-791):
    the planner folds it, so a circuit over constants needs no PBS at all."""
    import fhestr
    P = to_fhestr_params(O.TOY_K1)
    plan = fhestr.Plan(None, params=P)
    t = plan.lin([], 5)
    sq = plan.lut(lambda x: (x * x) % 16)
    plan.output(plan.pbs(t, sq))
    plan.finalize(1)
    info = plan.info()
    assert info["n_pbs"] == 0 and info["n_levels"] == 0
    lv = plan.export_level(0)
    assert lv["jobs"] == 1 and lv["terms"] == 0 and int(lv["cst"][0]) == 9 * P.delta   # 25 % 16 = 9
