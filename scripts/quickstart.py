"""The README's quick start, runnable (GPU box)."""
import sys; sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, fhestr.random_seed())            # client side: secret keys, encrypt / decrypt
eng = fhestr.Engine(P, 0)                                  # server side, GPU 0
eng.generate_keys(*ck.secret_keys(), fhestr.random_seed()) # or eng.load_keys(bsk_std, ksk) / eng.load_seeded_keys(...)
ops = fhestr.FheStringOps(eng)
enc = lambda s, cap: ck.encrypt(fhestr.string_to_blocks(P, s, cap))
hay, pat = enc(b"the quick brown fox", 32), enc(b"brown", 8)
print(ck.decrypt(ops.contains(hay, pat).reshape(1, -1))[0])                    # 1
print(fhestr.blocks_to_string(P, ck.decrypt(ops.replace(hay, b"quick", b"slow", out_cap=32))))   # b"the slow brown fox"
