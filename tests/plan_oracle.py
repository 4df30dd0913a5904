"""Checker backend: execute a plan's exported levels (fhe_plan_export_level / _export_lut) with the
CPU oracle, optionally all-gathering through torch.distributed (gloo).  Test infrastructure."""
import numpy as np

import oracle as O


def lincomb(pool, lv, jobs):
    """out[j] = sum coeff*pool[src] ; body += cst  (mod 2^64), for j in jobs."""
    out = np.zeros((len(jobs), pool.shape[1]), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for r, j in enumerate(jobs):
            for t in range(lv["off"][j], lv["off"][j + 1]):
                out[r] += pool[lv["src"][t]] * np.uint64(int(lv["coeff"][t]) % 2**64)
            out[r, -1] += lv["cst"][j]
    return out


class OracleBackend:
    def __init__(self, plan, sk: O.ServerKey, group=None):
        n_levels = plan.info()["n_levels"]
        self.levels = [plan.export_level(l) for l in range(n_levels + 1)]
        self.luts = plan.export_luts()
        self.plan = plan
        self.sk = sk
        self.group = group
        self.big = sk.params.big_size

    def alloc_pool(self, slots):
        return np.zeros((slots, self.big), dtype=np.uint64)

    def load_inputs(self, pool, inputs, n_inputs):
        pool[:n_inputs] = np.asarray(inputs, dtype=np.uint64).reshape(n_inputs, self.big)

    def run_level(self, pool, level, rank):
        lv = self.levels[level]
        ri = self.plan.level_rank_info(level, rank)
        jobs = list(range(ri["job_lo"], ri["job_hi"]))
        if not jobs:
            return
        staged = lincomb(pool, lv, jobs)
        ids = sorted(set(int(lv["lut"][j]) for j in jobs))
        luts = np.stack([self.luts[i] for i in ids])
        idx = np.array([ids.index(int(lv["lut"][j])) for j in jobs], dtype=np.uint32)
        pool[lv["local_base"]: lv["local_base"] + len(jobs)] = self.sk.apply_lookup_table_batch(staged, luts, idx)

    def all_gather(self, pool, local_base, e_max, recv_base, world):
        import torch
        import torch.distributed as dist
        recv = torch.from_numpy(pool[recv_base: recv_base + e_max * world].view(np.int64))
        own = torch.from_numpy(pool[local_base: local_base + e_max].view(np.int64)).clone()
        dist.all_gather_into_tensor(recv, own, group=self.group)

    def gather_outputs(self, pool, n_outputs):
        lv = self.levels[-1]
        return lincomb(pool, lv, list(range(lv["jobs"])))


def run_with_oracle(plan, inputs, sk, rank=0, world=1, group=None):
    from fhestr.distributed import ShardedPlanRunner
    return ShardedPlanRunner(plan, rank, world, OracleBackend(plan, sk, group)).run(inputs)
