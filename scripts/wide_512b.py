"""Follow-up to wide_512.py: (H1) does a 512-LWE launch pay for following launches that leave the GPU nearly idle?
(H2) does it pay when the two workgroups of a CU hold identical ciphertexts (bench.py's sweep tiled its 256 inputs)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
lut = eng.generate_lookup_table(lambda x: (x + 3) % 16)[0]
B = 512
msgs = np.arange(B) % 16
cts = ck.encrypt(msgs)
d_in = torch.from_numpy(cts.view(np.int64)).cuda()
d_dup = torch.from_numpy(np.concatenate([cts[:256], cts[:256]]).view(np.int64)).cuda()
d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
def run(d, n): eng.apply_lookup_table_dev(d.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), n)
for name, d in (("distinct", d_in), ("halves identical", d_dup)):
    for _ in range(3): run(d, B)
    eng.synchronize(); eng.kernel_times(reset=True)
    for _ in range(10): run(d, B)
    eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
    print(f"back to back, {name}: blind rotation {br / c:.3f} ms", flush=True)
for small in ((1, 1, 1), (35, 3, 1), (256,), (256, 256, 256)):
    tot = 0.0
    for rep in range(6):
        for n in small: run(d_in, n)
        eng.synchronize(); eng.kernel_times(reset=True)
        run(d_in, B)
        eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
        if rep: tot += br
    print(f"512-LWE launch after launches of {small} LWEs: blind rotation {tot / 5:.3f} ms", flush=True)
# same without the host synchronisation in between (what a plan does): time the whole sequence, subtract the small launches
for small in ((35, 3, 1),):
    for _ in range(2):
        for n in small: run(d_in, n)
        run(d_in, B)
    eng.synchronize(); eng.kernel_times(reset=True)
    for rep in range(5):
        run(d_in, B)
        for n in small: run(d_in, n)
    eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
    print(f"plan-like sequence 512,{small}: blind rotation total per round {br / 5:.3f} ms", flush=True)
