"""Checker: execute an exported plan (fhe_plan_export_level) with the CPU oracle, optionally sharded
over torch.distributed ranks exactly the way the GPU executor shards it.  Test infrastructure."""
import numpy as np

import oracle as O


def export_plan(plan):
    info = plan.info()
    levels = [plan.export_level(l) for l in range(info["n_levels"] + 1)]
    return dict(info=info, levels=levels)


def lincomb(pool, lv, jobs):
    """out[j] = sum coeff*pool[src] ; body += cst  (mod 2^64), for j in jobs."""
    out = np.zeros((len(jobs), pool.shape[1]), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for r, j in enumerate(jobs):
            for t in range(lv["off"][j], lv["off"][j + 1]):
                out[r] += pool[lv["src"][t]] * np.uint64(int(lv["coeff"][t]) % 2**64)
            out[r, -1] += lv["cst"][j]
    return out


def run_with_oracle(exported, inputs, sk: O.ServerKey, lut_tables, rank=0, world=1, all_gather=None):
    """lut_tables: {lut_id: accumulator}.  all_gather(region_rows, own_rows) fills region in place."""
    info = exported["info"]
    p = sk.params
    pool = np.zeros((info["pool_slots"], p.big_size), dtype=np.uint64)
    pool[: info["n_inputs"]] = inputs
    for lv in exported["levels"][:-1]:
        per = lv["per_rank"]
        lo, hi = rank * per, min(lv["jobs"], (rank + 1) * per)
        jobs = list(range(lo, hi)) if lo < hi else []
        if jobs:
            staged = lincomb(pool, lv, jobs)
            ids = sorted(set(int(lv["lut"][j]) for j in jobs))
            luts = np.stack([lut_tables[i] for i in ids])
            idx = np.array([ids.index(int(lv["lut"][j])) for j in jobs], dtype=np.uint32)
            pool[lv["base"] + lo: lv["base"] + hi] = sk.apply_lookup_table_batch(staged, luts, idx)
        if world > 1:
            region = pool[lv["base"]: lv["base"] + per * world]
            all_gather(region, region[rank * per: (rank + 1) * per].copy())
    out_lv = exported["levels"][-1]
    return lincomb(pool, out_lv, list(range(out_lv["jobs"])))
