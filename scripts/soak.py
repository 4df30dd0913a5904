"""Soak test: many random KS+PBS through every entry point variant, every output decrypted and checked.

    python3 scripts/soak.py [batches] [batch]     (default 100 x 1024 = 102,400 PBS; GPU box)

PARAM_MESSAGE_2_CARRY_2_KS_PBS, 16 random tables, fresh ciphertexts; alternates serial calls, pipelined calls
(fhe_engine_set_pipeline modes 1 and 2) on 256-LWE chunks and large batches (wide kernel).  The parameter set's failure
probability is 2^-40: any mismatch here is a bug, not noise."""
import sys
import numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr, torch

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
M = 16
ck = fhestr.ClientKey(P, 0x50AC)
eng = fhestr.Engine(P, 0)
eng.generate_keys(*ck.secret_keys(), 0x50AC)
rng = np.random.default_rng(1)
tables = rng.integers(0, M, size=(16, M))
luts = np.array([eng.generate_lookup_table(lambda x, t=t: int(t[x]))[0] for t in tables], dtype=np.uint32)
bad = total = 0
for it in range(NB):
    msgs = rng.integers(0, M, size=B)
    sel = rng.integers(0, 16, size=B)
    cts = ck.encrypt(msgs)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_idx = torch.from_numpy(luts[sel].view(np.int32)).cuda()
    d_out = torch.zeros_like(d_in)
    mode = it % 4
    if mode == 0:                       # one large batch (wide kernel above 256 LWEs)
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    else:                               # 256-LWE chunks: serial, keyswitch-shadow pipeline, overlapped on two streams
        eng.set_pipeline(mode - 1)
        for lo in range(0, B, 256):
            n = min(256, B - lo)
            eng.apply_lookup_table_dev(d_in[lo:].data_ptr(), d_idx[lo:].data_ptr(), d_out[lo:].data_ptr(), n)
        eng.synchronize()
        eng.set_pipeline(0)
    eng.synchronize()
    got = ck.decrypt(d_out.cpu().numpy().view(np.uint64))
    bad += int((got != tables[sel, msgs]).sum())
    total += B
    if (it + 1) % 20 == 0:
        print(f"{total} PBS, {bad} mismatches", flush=True)
print(f"soak: {total} PBS, {bad} mismatches")
sys.exit(1 if bad else 0)
