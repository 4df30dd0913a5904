"""Soak of the N = 1024 (k = 2) kernels: random batch sizes across the three layouts (one LWE per CU up to 256, two per CU
up to 512, the dense four-per-CU kernel (pbs_dense_kernels.hip.h) beyond), 16 random tables, fresh ciphertexts, every output decrypted.

    python3 scripts/soak_n1024.py [launches]      (default 80; GPU box)"""
import sys
import numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 80
P = fhestr.PARAM_MESSAGE_2_CARRY_1_KS_PBS
M = P.msg_mod * P.carry_mod
ck = fhestr.ClientKey(P, 0x50AD)
eng = fhestr.Engine(P, 0)
eng.generate_keys(*ck.secret_keys(), 0x50AD)
rng = np.random.default_rng(2)
tables = rng.integers(0, M, size=(16, M))
luts = np.array([eng.generate_lookup_table(lambda x, t=t: int(t[x]))[0] for t in tables], dtype=np.uint32)
bad = total = 0
by_layout = {"one per CU": 0, "two per CU": 0, "dense": 0}
for it in range(NL):
    B = int(rng.choice([int(rng.integers(1, 257)), int(rng.integers(257, 513)), int(rng.integers(513, 4097))], p=[0.2, 0.2, 0.6]))
    by_layout["one per CU" if B <= 256 else "two per CU" if B <= 512 else "dense"] += B
    msgs = rng.integers(0, M, size=B)
    sel = rng.integers(0, 16, size=B)
    out = eng.apply_lookup_table(ck.encrypt(msgs), luts[sel])
    got = ck.decrypt(out)
    want = tables[sel, msgs]
    bad += int((got != want).sum())
    total += B
    if it % 20 == 19:
        print(f"  {it + 1} launches, {total} PBS, {bad} wrong", flush=True)
print(f"{P.name}: {NL} launches, {total} PBS ({by_layout}), {bad} wrong")
eng.close()
sys.exit(1 if bad else 0)
