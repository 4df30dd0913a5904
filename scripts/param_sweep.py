#!/usr/bin/env python3
"""KS+PBS kernel time of every reference parameter set (tests/golden/reference_parameter_sets.json) at one batch size:

    python3 scripts/param_sweep.py [B [N [name-part]]]      (default 256, every polynomial size and set; GPU box)

Device-generated keys, decrypt-checked; prints keyswitch / blind-rotation ms (HIP events in the engine) and PBS/s."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
import fhestr  # noqa: E402

TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_parameter_sets.json")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY_N = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ONLY_NAME = sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for name in sorted(TABLE, key=lambda k: (TABLE[k]["polynomial_size"], k)):
    r = TABLE[name]
    if r["encryption_key_choice"] == "Small" or (ONLY_N and r["polynomial_size"] != ONLY_N) or ONLY_NAME not in name:
        continue
    P = fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"],
                      r["ks_base_log"], r["ks_level"], r["message_modulus"], r["carry_modulus"],
                      r["lwe_modular_std_dev"], r["glwe_modular_std_dev"], name, r.get("grouping_factor", 0))
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, 3)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    eng.generate_keys(g, s, 3)
    rng = np.random.default_rng(1)
    table = rng.integers(0, M, size=M)
    lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
    msgs = rng.integers(0, M, size=B)
    cts = ck.encrypt(msgs)
    idx = np.full(B, lut, dtype=np.uint32)
    out = eng.apply_lookup_table(cts, idx)
    ok = bool(np.array_equal(ck.decrypt(out), table[msgs]))
    eng.apply_lookup_table(cts, idx)
    ks, br = eng.last_kernel_ms()
    print(f"{name:52s} N={P.N:5d} k={P.k} l={P.pbs_level} n={P.n:4d} g={max(P.grouping,1)}: keyswitch {ks:8.3f} ms, blind rotation {br:9.3f} ms "
          f"-> {B / ((ks + br) * 1e-3):9.0f} PBS/s, correct {ok}", flush=True)
    eng.close()
