"""PARAM_MESSAGE_2_CARRY_2: blind-rotation kernel time at 512 / 1024 / 4096 LWEs (the two-LWEs-per-CU kernel), decrypt-checked."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
lut, _ = eng.generate_lookup_table(lambda x: (x + 3) % 16)
for B in (256, 512, 768, 1024, 4096):
    msgs = np.arange(B) % 16
    d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
    d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
    for _ in range(3): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize(); eng.kernel_times(reset=True)
    for _ in range(10): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
    ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), (msgs + 3) % 16))
    print(f"{os.environ.get('FHESTR_LIB', 'default')} B={B}: keyswitch {ks / c * 1e3:.0f} us, blind rotation {br / c:.3f} ms -> {B / ((ks + br) / c) * 1e3:.0f} PBS/s, correct {ok}", flush=True)
