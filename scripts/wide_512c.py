"""Follow-up 2: what a launch costs right after launches that left the GPU nearly idle (1-LWE launches), per size."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
lut = eng.generate_lookup_table(lambda x: (x + 3) % 16)[0]
B = 2048
cts = ck.encrypt(np.arange(B) % 16)
d_in = torch.from_numpy(cts.view(np.int64)).cuda()
d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
def run(n): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), n)
for n in (64, 256, 512, 1024, 2048):
    for before in ((256, 256), (1, 1, 1)):
        tot = 0.0
        for rep in range(6):
            for m in before: run(m)
            eng.synchronize(); eng.kernel_times(reset=True)
            run(n)
            eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
            if rep: tot += br
        print(f"fair={os.environ.get('FHESTR_WIDE_FAIR', 'default')} {n}-LWE launch after launches of {before} LWEs: blind rotation {tot / 5:.3f} ms", flush=True)
