"""BASELINE.json config 5 on one GPU: replace + to_lower on a 1024-char string, PARAM_MESSAGE_4_CARRY_4_KS_PBS
(N = 32768) with real keys; decrypt-checked against Python."""
import sys, time, os, numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.Params(996, 1, 32768, 15, 2, 3, 7, 16, 16, 6.767666038309478e-08, 2.168404344971009e-19, "PARAM_MESSAGE_4_CARRY_4_KS_PBS")
ck = fhestr.ClientKey(P, 0x5EED0005)
eng = fhestr.Engine(P, 0)
g, sm = ck.secret_keys()
t = time.time(); eng.generate_keys(g, sm, 0x5EED0005); print(f"device keygen {time.time() - t:.1f} s", flush=True)
ops = fhestr.FheStringOps(eng)
rng = np.random.default_rng(0x5EED0005)
words = [b"The ", b"quick ", b"BROWN ", b"fox ", b"Jumps ", b"over ", b"the ", b"LAZY ", b"dog. "]
s = b"".join(words[int(i)] for i in rng.integers(0, len(words), size=400))[:1000]
es = ck.encrypt(fhestr.string_to_blocks(P, s, 1024))
dec = lambda ct: fhestr.blocks_to_string(P, ck.decrypt(ct))
for name, fn, want in (("to_lower", lambda: ops.to_lower(es), s.lower()),
                       ("replace('the '->'THAT')", lambda: ops.replace(es, b"the ", b"THAT"), s.replace(b"the ", b"THAT"))):
    t = time.time(); out = fn(); dt = time.time() - t
    print(f"{name}: {dt * 1e3:.0f} ms, correct = {dec(out) == want}", flush=True)
for op, b_cap, clear in (("to_lower", 0, None), ("replace_clear", 0, b"the THAT")):
    print(op, fhestr.Plan.string_op(eng, op, 1024, b_cap, clear).info())
