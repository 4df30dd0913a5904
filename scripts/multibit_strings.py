"""FheString::eq (256 vs 256 chars) and ::contains (16 in 256), both encrypted, on the multi-bit PBS engine
(PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS) next to the classic one; device-generated keys."""
import sys, time
import numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

rng = np.random.default_rng(0x5EED0003)
hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
pat = hay[100:116]
for P in (fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS,
          fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS):
    ck = fhestr.ClientKey(P, 0x5EED0002)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    eng.generate_keys(g, s, 0x5EED0002)
    ops = fhestr.FheStringOps(eng)
    enc = lambda b, cap: ck.encrypt(fhestr.string_to_blocks(P, b, cap))
    dec = lambda ct: ck.decrypt(np.asarray(ct).reshape(-1, P.big_size))
    eh, eh2, ep = enc(hay, 256), enc(hay, 256), enc(pat, 16)
    for name, fn, want in (("eq_256", lambda: ops.eq(eh, eh2), 1), ("contains_16_in_256", lambda: ops.contains(eh, ep), 1)):
        fn()
        t = time.time(); out = fn(); dt = time.time() - t
        print(f"{P.name} {name}: {dt * 1e3:.1f} ms (inputs from host), correct {int(dec(out)[0]) == want}", flush=True)
    eng.close()
