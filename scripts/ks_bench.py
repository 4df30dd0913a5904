"""Keyswitch kernel time by batch size, both implementations (FHESTR_KS_MFMA=1: int8 matrix product on the matrix cores,
0: byte-plane v_dot4 kernel), bit-compared against each other:  python scripts/ks_bench.py [p44]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr  # noqa: E402

P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS if "p44" in sys.argv[1:] else fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
rng = np.random.default_rng(0)
ck = fhestr.ClientKey(P, 3)
g, s = ck.secret_keys()
outs = {}
for mode in (os.environ.get("KS_MODES", "1,0").split(",")):
    os.environ["FHESTR_KS_MFMA"] = mode
    eng = fhestr.Engine(P, 0)
    eng.generate_keys(g, s, 3)
    eng.generate_lookup_table(lambda x: x)
    for B in (1, 3, 35, 256, 512, 1024, 4096):
        cts = rng.integers(0, 2**64, size=(B, P.big_size), dtype=np.uint64) if B != 256 else ck.encrypt(rng.integers(0, 16, size=B))
        small = eng.keyswitch(cts)
        outs.setdefault(B, []).append((cts, small))
        eng.apply_lookup_table(cts)
        eng.apply_lookup_table(cts)
        ks, br = eng.last_kernel_ms()
        print(f"{P.name} KS_MFMA={mode} B={B}: keyswitch {ks * 1e3:.1f} us, blind rotation {br:.3f} ms", flush=True)
    eng.close()
    rng = np.random.default_rng(0)
for B, pair in (outs.items() if len(next(iter(outs.values()))) > 1 else []):
    same_in = np.array_equal(pair[0][0], pair[1][0])
    print(f"B={B}: outputs bit-identical: {bool(same_in and np.array_equal(pair[0][1], pair[1][1]))}")
