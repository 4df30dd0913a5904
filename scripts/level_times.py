"""Per-level kernel times of FheString::eq (256 chars) on PARAM_MESSAGE_2_CARRY_2: where the 13 ms go."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 1); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 1)
rng = np.random.default_rng(0)
hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
enc = lambda t, cap: ck.encrypt(fhestr.string_to_blocks(P, t, cap))
for op, bcap, second in (("eq", 256, hay), ("contains", 16, hay[100:116])):
    plan = fhestr.Plan.string_op(eng, op, 256, bcap)
    info = plan.info()
    inputs = torch.from_numpy(np.concatenate([enc(hay, 256), enc(second, bcap)]).view(np.int64)).cuda()
    pool = torch.zeros((info["pool_slots"], P.big_size), dtype=torch.int64, device="cuda")
    pool[:info["n_inputs"]].copy_(inputs)
    torch.cuda.synchronize()
    for rep in range(3):
        rows = []
        for l in range(info["n_levels"]):
            eng.synchronize(); eng.kernel_times(reset=True)
            plan.run_level_rank_dev(pool.data_ptr(), l, 0)
            eng.synchronize()
            ks, br, c = eng.kernel_times(reset=True)
            rows.append((plan.level_info(l)["local_size"], round(ks * 1e3), round(br, 3)))
    print(op, "levels (LWEs, keyswitch us, blind rotation ms):", rows, "sum", round(sum(r[1] / 1e3 + r[2] for r in rows), 2), "ms")
