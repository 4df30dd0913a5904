"""Serial 256-LWE steps, PARAM_MESSAGE_2_CARRY_2: kernel times with the MFMA keyswitch and with the dot4 one (same process)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for rep in range(2):
    for mode in ("1", "0"):
        os.environ["FHESTR_KS_MFMA"] = mode
        eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
        lut, _ = eng.generate_lookup_table(lambda x: x)
        d_in = torch.from_numpy(ck.encrypt(np.arange(B) % 16).view(np.int64)).cuda()
        d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
        for _ in range(10): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
        eng.synchronize(); eng.kernel_times(reset=True)
        for _ in range(40): eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
        eng.synchronize(); ks, br, c = eng.kernel_times(reset=True)
        print(f"KS_MFMA={mode} B={B}: keyswitch {ks / c * 1e3:.1f} us, blind rotation {br / c:.4f} ms", flush=True)
        eng.close()
