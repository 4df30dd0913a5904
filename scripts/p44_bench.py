#!/usr/bin/env python3
"""PARAM_MESSAGE_4_CARRY_4_KS_PBS KS+PBS step time by batch size, decrypt-checked (GPU box):

    python3 scripts/p44_bench.py [--modes 1,0] [--n16384] [B ...]     (default 64 256; FHESTR_LIB selects the build)

--modes: blind-rotation kernel per pass (1 = cluster kernel, several CUs per LWE; 0 = one workgroup per LWE; -1 = automatic).

Prints the blind-rotation and keyswitch kernel ms (HIP events inside the engine) and the PBS/s of each batch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
import fhestr  # noqa: E402
import torch  # noqa: E402

argv = sys.argv[1:]
modes = [-1]
if "--modes" in argv:
    k = argv.index("--modes")
    modes = [int(m) for m in argv[k + 1].split(",")]
    del argv[k:k + 2]
P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
if "--n16384" in argv:       # PARAM_MESSAGE_3_CARRY_4_KS_PBS (shortint/parameters/mod.rs): N = 16384, two levels
    argv.remove("--n16384")
    P = fhestr.Params(930, 1, 16384, 15, 2, 3, 6, 8, 16, 2.2649232786295453e-07, 2.168404344971009e-19, "PARAM_MESSAGE_3_CARRY_4_KS_PBS")
M = P.msg_mod * P.carry_mod
ck = fhestr.ClientKey(P, 77)
g, sm = ck.secret_keys()
eng = fhestr.Engine(P, 0)
eng.generate_keys(g, sm, 77)
rng = np.random.default_rng(5)
table = rng.integers(0, M, size=M)
lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
for mode, B in [(m, b) for m in modes for b in ([int(a) for a in argv] or [64, 256])]:
    eng.set_cluster_mode(mode)
    msgs = rng.integers(0, M, size=B)
    d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
    d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")
    d_out = torch.zeros_like(d_in)
    eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize()
    eng.kernel_times(reset=True)
    reps = 2
    for _ in range(reps):
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize()
    ks_ms, br_ms, calls = eng.kernel_times(reset=True)
    ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
    print(f"{fhestr.kernel_revision()} {P.name} cluster mode {mode} B={B}: keyswitch {ks_ms / calls:.2f} ms, blind rotation {br_ms / calls:.2f} ms -> "
          f"{B / ((ks_ms + br_ms) / calls) * 1e3:.0f} PBS/s, correct {ok}", flush=True)
eng.close() if hasattr(eng, "close") else None
