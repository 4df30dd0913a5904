"""Why does FheString::eq's 512-LWE first level take 4.9 ms when back-to-back 512-LWE launches take 4.15 ms?
Blind-rotation time of the two-LWEs-per-CU kernel at B = 497 / 511 / 512 / 513, back to back or with a host
synchronisation before every launch, one shared table or 16 random ones.  (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
luts = [eng.generate_lookup_table(lambda x, t=t: (x + t) % 16)[0] for t in range(16)]
rng = np.random.default_rng(0)
for B in (497, 511, 512, 513, 1024):
    msgs = np.arange(B) % 16
    d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
    for many in (False, True):
        sel = rng.integers(0, 16, size=B) if many else np.zeros(B, dtype=np.int64)
        d_idx = torch.from_numpy(np.array(luts, dtype=np.uint32)[sel].view(np.int32)).cuda(); d_out = torch.zeros_like(d_in)
        for sync_each in (False, True):
            for _ in range(3): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
            eng.synchronize(); eng.kernel_times(reset=True)
            for _ in range(10):
                eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
                if sync_each: eng.synchronize()
            eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
            ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), (msgs + sel) % 16))
            print(f"B={B} tables={'16' if many else '1'} sync_each={sync_each}: keyswitch {ks / c * 1e3:.0f} us, blind rotation {br / c:.3f} ms, correct {ok}", flush=True)
