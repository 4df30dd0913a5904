"""Per-level times of FheString plans on PARAM_MESSAGE_2_CARRY_2 (GPU box): LWEs per level, keyswitch and blind-rotation kernel
time, and the wall time of the level (host synchronised on both sides: launch gaps, linear parts, table gathers included).

    python3 scripts/level_times.py [op:a_cap:b_cap ...]      default: eq:256:256 contains:256:16 to_lower:1024:0"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 1); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 1)
rng = np.random.default_rng(0)
specs = [a.split(":") for a in sys.argv[1:]] or [["eq", "256", "256"], ["contains", "256", "16"], ["to_lower", "1024", "0"]]
for op, a_cap, b_cap in specs:
    a_cap, b_cap = int(a_cap), int(b_cap)
    hay = bytes(rng.integers(0x41, 0x7B, size=a_cap, dtype=np.uint8))
    second = hay[:b_cap] if b_cap == a_cap else hay[a_cap // 3:a_cap // 3 + b_cap]
    enc = lambda t, cap: ck.encrypt(fhestr.string_to_blocks(P, t, cap))
    plan = fhestr.Plan.string_op(eng, op, a_cap, b_cap)
    info = plan.info()
    parts = [enc(hay, a_cap)] + ([enc(second, b_cap)] if b_cap else [])
    inputs = torch.from_numpy(np.concatenate(parts).view(np.int64)).cuda()
    assert inputs.shape[0] == info["n_inputs"], (inputs.shape, info["n_inputs"])
    pool = torch.zeros((info["pool_slots"], P.big_size), dtype=torch.int64, device="cuda")
    pool[:info["n_inputs"]].copy_(inputs)
    torch.cuda.synchronize()
    for rep in range(3):
        rows = []
        for l in range(info["n_levels"]):
            eng.synchronize(); eng.kernel_times(reset=True)
            t0 = time.perf_counter()
            plan.run_level_rank_dev(pool.data_ptr(), l, 0)
            eng.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            ks, br, c = eng.kernel_times(reset=True)
            rows.append((plan.level_info(l)["local_size"], round(ks * 1e3), round(br, 3), round(wall, 3)))
    n = sum(r[0] for r in rows)
    kern = sum(r[1] / 1e3 + r[2] for r in rows); wall = sum(r[3] for r in rows)
    print(f"{op} {a_cap}/{b_cap}: {len(rows)} levels, {n} PBS; kernels {kern:.2f} ms, level walls {wall:.2f} ms -> {n / wall:.1f} k PBS/s")
    if len(rows) <= 12:
        print("   (LWEs, keyswitch us, blind rotation ms, wall ms):", rows)
    else:
        sizes = sorted(set(r[0] for r in rows))
        print("   by level size:", {sz: (sum(1 for r in rows if r[0] == sz), round(sum(r[3] for r in rows if r[0] == sz), 2)) for sz in sizes})
