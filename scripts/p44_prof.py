"""One PARAM_MESSAGE_4_CARRY_4 launch per batch size (device-generated keys), for rocprofv3 runs."""
import sys, time, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
ck = fhestr.ClientKey(P, 0x5EED0044)
g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0)
t = time.time(); eng.generate_keys(g, s, 0x5EED0044); print(f"device keygen {time.time()-t:.1f} s", flush=True)
lut, _ = eng.generate_lookup_table(lambda x: (x + 1) % 256)
rng = np.random.default_rng(1)
for B in [int(a) for a in sys.argv[1:]] or [256]:
    msgs = rng.integers(0, 256, size=B)
    cts = ck.encrypt(msgs)
    out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
    ks, br = eng.last_kernel_ms()
    ok = np.array_equal(ck.decrypt(out), (msgs + 1) % 256)
    print(f"B={B}: ks {ks:.1f} ms, blind_rotate {br:.1f} ms -> {B/((ks+br)*1e-3):.0f} PBS/s, correct {ok}", flush=True)
