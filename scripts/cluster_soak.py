"""Soak of blind_rotate_cluster_kernel / blind_rotate_xcd_kernel under uneven load (GPU box): many launches of 256 and of odd batch sizes on
PARAM_MESSAGE_4_CARRY_4 (N = 32768) and PARAM_MESSAGE_3_CARRY_4 (N = 16384), a second stream hammering HBM with copies
of changing size at the same time (so the clusters of different XCDs run at different speeds and the L2s see foreign
traffic), every output decrypted.  A stale hand-over (a consumer reading an exchange buffer before the producer's bytes
are in L2) corrupts an accumulator and decrypts to garbage; the kernel's own bounded waits report through the engine.

    python3 scripts/cluster_soak.py [launches]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import fhestr  # noqa: E402
import torch  # noqa: E402
from noise_budget import reference_params  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
stop = False


def ballast():
    s = torch.cuda.Stream()
    a = torch.empty(96 << 20, dtype=torch.int64, device="cuda")      # 768 MB: past the Infinity Cache
    b = torch.empty_like(a)
    rng = np.random.default_rng(1)
    with torch.cuda.stream(s):
        while not stop:
            n = int(rng.integers(1 << 20, 96 << 20))
            b[:n].copy_(a[:n], non_blocking=True)
            if rng.random() < 0.3:
                s.synchronize()
                time.sleep(float(rng.random()) * 2e-3)
    s.synchronize()


th = threading.Thread(target=ballast)
th.start()
total = bad = 0
try:
    for name in ("PARAM_MESSAGE_4_CARRY_4_KS_PBS", "PARAM_MESSAGE_3_CARRY_4_KS_PBS"):
        P = reference_params(name)
        M = P.msg_mod * P.carry_mod
        ck = fhestr.ClientKey(P, 9)
        g, s = ck.secret_keys()
        eng = fhestr.Engine(P, 0)
        eng.generate_keys(g, s, 9)
        rng = np.random.default_rng(2)
        table = rng.integers(0, M, size=M)
        lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
        t0 = time.time()
        for it in range(launches):
            eng.set_cluster_mode((1, 2, -1)[it % 3])     # whole-XCD kernel (where it exists), 8-CU clusters, automatic
            B = 256 if it % 4 == 0 else int(rng.integers(1, 300))
            msgs = rng.integers(0, M, size=B)
            out = eng.apply_lookup_table(ck.encrypt(msgs), np.full(B, lut, dtype=np.uint32))
            wrong = int((ck.decrypt(out) != table[msgs]).sum())
            total += B
            bad += wrong
            if wrong:
                print(f"{name} launch {it} B={B}: {wrong} wrong", flush=True)
        print(f"{name}: {launches} launches in {time.time() - t0:.1f} s, clusters {eng.cluster_info()}, running totals {total} PBS, {bad} wrong", flush=True)
        eng.close()
finally:
    stop = True
    th.join()
print("SOAK", "FAILED" if bad else "ok", total, "PBS", bad, "wrong")
sys.exit(1 if bad else 0)
