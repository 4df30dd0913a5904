"""tests/exact_pbs.py (numpy, exact limb-FFT products) against the C oracle's schoolbook exact path: bit for bit."""
import numpy as np
import pytest

import oracle as O
from conftest import keyset
from exact_pbs import decompose, negacyclic_mul_exact, pbs_exact


def test_negacyclic_limb_product_is_exact():
    rng = np.random.default_rng(1)
    N = 64
    d = rng.integers(-(1 << 14), 1 << 14, size=N)
    key = rng.integers(0, 2**64, size=N, dtype=np.uint64)
    want = np.zeros(N, dtype=object)
    for i in range(N):
        for j in range(N):
            t = int(d[j]) * int(key[(i - j) % N])
            want[i] += -t if j > i else t
    assert [int(v) % 2**64 for v in want] == [int(v) for v in negacyclic_mul_exact(d, key)]


def test_decompose_recomposes():
    rng = np.random.default_rng(2)
    x = rng.integers(0, 2**64, size=1000, dtype=np.uint64)
    for bl, L in ((15, 2), (23, 1), (11, 3), (8, 2)):
        digs = decompose(x, bl, L)
        rec = np.zeros(len(x), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for it, dg in enumerate(digs):                 # it = 0 is level L: weight 2^(64 - bl * L)
                rec += dg.astype(np.uint64) << np.uint64(64 - bl * (L - it))
        err = (rec - x).astype(np.int64)
        assert np.abs(err).max() <= 1 << (63 - bl * L)
        assert all(np.abs(dg).max() <= 1 << (bl - 1) for dg in digs)


@pytest.mark.parametrize("params", [O.TOY_K1, O.TOY_K2, next(p for p in O.TOY_SHAPES if p.name == "TOY_N512_K2_L2")], ids=lambda p: p.name)
def test_numpy_exact_pbs_equals_the_oracles_schoolbook_path(params):
    ks = keyset(params)
    M = params.msg_mod * params.carry_mod
    lut, _ = ks.sk.generate_lookup_table(lambda x: (3 * x + 1) % M)
    cts = ks.ck.encrypt_many([0, 1, M - 1], O.Rng(5, 5))
    for ct in cts:
        small = ks.sk.keyswitch(ct)
        assert np.array_equal(pbs_exact(params, ks.sk.bsk, small, lut), ks.sk.pbs(small, lut, exact=True))
    small = ks.sk.keyswitch(cts[0]).copy()
    small[1] = 0                                           # a_i == 0 is skipped
    assert np.array_equal(pbs_exact(params, ks.sk.bsk, small, lut), ks.sk.pbs(small, lut, exact=True))
