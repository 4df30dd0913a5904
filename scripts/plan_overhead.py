"""How much of a string operation's wall time is not KS / blind-rotation kernel time (launch gaps, linear combinations,
input upload): eq and contains on 256-char strings, inputs from host (GPU box).  Round 2: eq 14.26 ms wall vs 13.47 ms
of kernels, contains 126.05 vs 125.31 ms -- nothing a hipGraph capture would win back."""
import sys, time
sys.path.insert(0, "fhe-string-bounty_amd")
import numpy as np, torch, fhestr
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 1); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 1)
rng = np.random.default_rng(0)
hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
enc = lambda t, cap: ck.encrypt(fhestr.string_to_blocks(P, t, cap))
text = bytes(rng.integers(0x41, 0x7B, size=1000, dtype=np.uint8))
for op, acap, bcap, second in (("eq", 256, 256, hay), ("contains", 256, 16, hay[100:116]), ("to_lower", 1024, 0, None)):
    plan = fhestr.Plan.string_op(eng, op, acap, bcap)
    inputs = np.concatenate([enc(hay, 256), enc(second, bcap)]) if second is not None else enc(text, 1024)
    print(op, "levels:", [plan.level_info(l)["local_size"] for l in range(plan.info()["n_levels"])])
    info = plan.info()
    plan.run(inputs)
    eng.synchronize(); eng.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        plan.run(inputs)
    eng.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    ks, br, calls = eng.kernel_times(reset=True)
    print(op, f"wall {wall:.2f} ms, keyswitch {ks/5:.2f} ms + blind rotation {br/5:.2f} ms = {(ks+br)/5:.2f} ms over {calls/5:.0f} KS+PBS launches; levels {info['n_levels']}")
