"""End-to-end PARAM_MESSAGE_4_CARRY_4_KS_PBS (N = 32768) with real keys: keygen (CPU client), KS+PBS on GPU, decrypt."""
import sys, time, os, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.Params(996, 1, 32768, 15, 2, 3, 7, 16, 16, 6.767666038309478e-08, 2.168404344971009e-19, "PARAM_MESSAGE_4_CARRY_4_KS_PBS")
ck = fhestr.ClientKey(P, 0x5EED0005)
t = time.time(); bsk, ksk = ck.gen_server_keys(min(32, os.cpu_count())); print("keygen s", time.time() - t, "threads", min(32, os.cpu_count()), flush=True)
eng = fhestr.Engine(P, 0); eng.load_keys(bsk, ksk)
M = 256
f = lambda x: (x * x + 3) % M
lid, _ = eng.generate_lookup_table(f)
msgs = np.array([0, 1, 2, 15, 16, 100, 200, 255, 128, 127, 64, 33, 77, 254, 3, 9])
cts = ck.encrypt(msgs)
t = time.time(); out = eng.apply_lookup_table(cts, np.full(len(msgs), lid, dtype=np.uint32)); print("16 PBS s", time.time() - t)
dec = ck.decrypt(out)
print("decrypted", dec.tolist()); print("expected ", [f(int(m)) for m in msgs]); print("OK" if dec.tolist() == [f(int(m)) for m in msgs] else "MISMATCH")
