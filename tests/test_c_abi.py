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
