"""Timing of PARAM_MESSAGE_4_CARRY_4_KS_PBS (N = 32768) with random keys (timing only)."""
import sys, time, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.Params(996, 1, 32768, 15, 2, 3, 7, 16, 16, 6.767666038309478e-08, 2.168404344971009e-19, "PARAM_MESSAGE_4_CARRY_4_KS_PBS")
rng = np.random.default_rng(0)
t = time.time()
bsk = rng.integers(0, 2**64, size=P.bsk_len, dtype=np.uint64)
ksk = rng.integers(0, 2**64, size=P.ksk_len, dtype=np.uint64)
print("random keys", time.time() - t, "s; bsk GB", bsk.nbytes / 1e9, "ksk GB", ksk.nbytes / 1e9, flush=True)
eng = fhestr.Engine(P, 0)
t = time.time(); eng.load_keys(bsk, ksk); print("load+convert", time.time() - t, "s", flush=True)
eng.generate_lookup_table(lambda x: x)
for B in (16, 64, 256):
    cts = rng.integers(0, 2**64, size=(B, P.big_size), dtype=np.uint64)
    t = time.time(); eng.apply_lookup_table(cts); wall = time.time() - t
    ks, br = eng.last_kernel_ms()
    print(f"B={B}: ks {ks:.1f} ms, blind_rotate {br:.1f} ms, wall {wall*1e3:.0f} ms -> {B/((ks+br)*1e-3):.0f} PBS/s", flush=True)
