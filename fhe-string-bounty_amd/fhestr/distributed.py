"""Multi-GPU execution of a plan: one process per GPU, keys replicated, every level's batch of
KS+PBS jobs split into contiguous per-rank slices, one all-gather of the level's outputs per level
(RCCL over xGMI through torch.distributed's "nccl" backend).

The reference has no distributed code; the shardable unit is the independent
KS+PBS of one ciphertext.  The control flow below is backend-agnostic: the
product backend (GpuBackend) launches HIP kernels through the C ABI; tests may inject a checker
backend (CPU oracle + gloo) to exercise the sharding protocol without a GPU.
"""
from __future__ import annotations

import numpy as np


def slice_bounds(jobs: int, per_rank: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of a level's jobs owned by `rank` (may be empty)."""
    lo = min(jobs, rank * per_rank)
    return lo, min(jobs, lo + per_rank)


class ShardedPlanRunner:
    def __init__(self, plan, rank: int, world: int, backend):
        info = plan.info()
        if info["world"] != world:
            raise ValueError(f"plan was finalised for world={info['world']}, runner has world={world}")
        self.plan, self.rank, self.world, self.backend, self.info = plan, rank, world, backend, info
        self.levels = [plan.level_info(l) for l in range(info["n_levels"])]

    def run(self, inputs):
        b = self.backend
        pool = b.alloc_pool(self.info["pool_slots"])
        b.load_inputs(pool, inputs, self.info["n_inputs"])
        for l, lv in enumerate(self.levels):
            lo, hi = slice_bounds(lv["jobs"], lv["per_rank"], self.rank)
            if lo < hi:
                b.run_level_slice(pool, l, lo, hi)
            if self.world > 1:
                b.all_gather(pool, lv["base"], lv["per_rank"], self.rank, self.world)
        return b.gather_outputs(pool, self.info["n_outputs"])


class GpuBackend:
    """Product backend: pool in HBM (torch tensor), HIP kernels via the C ABI, RCCL all-gather."""

    def __init__(self, plan, device, group=None):
        import torch
        self.torch = torch
        self.plan = plan
        self.device = device
        self.group = group
        self.big = plan.params.big_size
        # kernels and collectives on the same stream: no host synchronisation between them
        plan.engine.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def alloc_pool(self, slots):
        return self.torch.zeros((slots, self.big), dtype=self.torch.int64, device=self.device)

    def load_inputs(self, pool, inputs, n_inputs):
        if isinstance(inputs, self.torch.Tensor):      # already resident in HBM (int64 view of the u64 words)
            pool[:n_inputs].copy_(inputs.reshape(n_inputs, self.big))
            return
        arr = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(n_inputs, self.big)
        pool[:n_inputs].copy_(self.torch.from_numpy(arr.view(np.int64)))

    def run_level_slice(self, pool, level, lo, hi):
        self.plan.run_level_slice_dev(pool.data_ptr(), level, lo, hi)

    def all_gather(self, pool, base, per_rank, rank, world):
        import torch.distributed as dist
        region = pool[base: base + per_rank * world]
        own = region[rank * per_rank: (rank + 1) * per_rank].clone()
        dist.all_gather_into_tensor(region, own, group=self.group)

    def gather_outputs(self, pool, n_outputs):
        out = self.torch.empty((n_outputs, self.big), dtype=self.torch.int64, device=self.device)
        self.plan.gather_outputs_dev(pool.data_ptr(), out.data_ptr())
        self.torch.cuda.current_stream(self.device).synchronize()
        return out.cpu().numpy().view(np.uint64)
