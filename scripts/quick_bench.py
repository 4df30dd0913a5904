import sys, time, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
rng = np.random.default_rng(0)
bsk = rng.integers(0, 2**64, size=P.bsk_len, dtype=np.uint64)
ksk = rng.integers(0, 2**64, size=P.ksk_len, dtype=np.uint64)
for lp in (0, 2, 18):
    eng = fhestr.Engine(P, 0, lp)
    eng.load_keys(bsk, ksk)
    eng.generate_lookup_table(lambda x: x)
    for B in (256, 512, 1024, 2048):
        cts = rng.integers(0, 2**64, size=(B, P.big_size), dtype=np.uint64)
        eng.apply_lookup_table(cts)
        t = time.time(); eng.apply_lookup_table(cts); wall = time.time() - t
        ks, br = eng.last_kernel_ms()
        print(f"log2pts={lp} B={B}: ks {ks:.3f} ms, blind_rotate {br:.3f} ms, wall {wall*1e3:.1f} ms -> {B/((ks+br)*1e-3):.0f} PBS/s (kernel)", flush=True)
    eng.close()
