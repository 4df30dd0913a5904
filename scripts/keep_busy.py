"""fhe_engine_set_keep_busy: FheString::eq / contains (256 chars, PARAM_MESSAGE_2_CARRY_2) with and without replicas on
the idle CUs during the small levels; per-level kernel times.  (GPU box)"""
import os, sys, time
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
    out = torch.zeros((info["n_outputs"], P.big_size), dtype=torch.int64, device="cuda")
    pool[:info["n_inputs"]].copy_(inputs)
    for busy in (False, True, False, True):
        eng.set_keep_busy(busy)
        torch.cuda.synchronize()
        for rep in range(12):
            if rep == 2:
                eng.synchronize(); t0 = time.perf_counter()
            for l in range(info["n_levels"]):
                plan.run_level_rank_dev(pool.data_ptr(), l, 0)
            plan.gather_outputs_dev(pool.data_ptr(), out.data_ptr())
        eng.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        ok = int(ck.decrypt(out.cpu().numpy().view(np.uint64))[0]) == 1
        print(f"{op}: keep_busy={busy}: {ms:.2f} ms per op (back to back, resident), correct {ok}", flush=True)
eng.set_keep_busy(False)
