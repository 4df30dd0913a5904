"""Multi-GPU execution of a plan: one process per GPU, keys replicated.  Every KS+PBS of a plan has an
owner rank (csrc/circuit.h): a rank runs its own jobs of a level into its local pool region, and only
what another rank -- or the final output gather -- consumes is exchanged: one all-gather of `e_max`
ciphertexts per rank for a level that exports anything (RCCL over xGMI through torch.distributed's
"nccl" backend), no collective at all for a level that does not.  For FheString::eq on 256 chars over 8
GPUs that is two all-gathers of one ciphertext per rank (reduce locally, gather the
reduced blocks).

The reference has no distributed code.  The control flow below is backend-agnostic: the
product backend (GpuBackend) launches HIP kernels through the C ABI; tests may inject a checker backend
(CPU oracle + gloo) to exercise the sharding protocol without a GPU.
"""
from __future__ import annotations

import numpy as np


class ShardedPlanRunner:
    def __init__(self, plan, rank: int, world: int, backend):
        info = plan.info()
        if info["world"] != world:
            raise ValueError(f"plan was finalised for world={info['world']}, runner has world={world}")
        self.plan, self.rank, self.world, self.backend, self.info = plan, rank, world, backend, info
        self.levels = [plan.level_info(l) for l in range(info["n_levels"])]
        # ciphertexts this rank receives over the whole plan, and the bytes that is
        self.gathered_lwes = sum(lv["e_max"] * world for lv in self.levels) if world > 1 else 0
        self.gathered_bytes = self.gathered_lwes * plan.params.big_size * 8
        self.collectives = sum(1 for lv in self.levels if lv["e_max"]) if world > 1 else 0

    def run(self, inputs, device_outputs=False):
        """inputs: host array or a tensor resident in HBM.  device_outputs=True (GpuBackend): the outputs stay in HBM too
        (a torch int64 tensor [n_outputs][big_size], complete when this returns) -- a 1024-char string is 67 MB of
        ciphertext, and a caller that feeds the next operation has no use for a host copy."""
        b = self.backend
        pool = b.alloc_pool(self.info["pool_slots"])
        b.load_inputs(pool, inputs, self.info["n_inputs"])
        for l, lv in enumerate(self.levels):
            b.run_level(pool, l, self.rank)
            if self.world > 1 and lv["e_max"]:
                b.all_gather(pool, lv["local_base"], lv["e_max"], lv["recv_base"], self.world)
        if device_outputs:
            return b.gather_outputs(pool, self.info["n_outputs"], to_host=False)
        return b.gather_outputs(pool, self.info["n_outputs"])


def instance_slice(instances: int, rank: int, world: int):
    """Contiguous share of `instances` independent plan instances rank `rank` runs (the first ranks take one more)."""
    base, extra = divmod(instances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def run_instances_sharded(run_batch, inputs, rank: int, world: int, group=None, gather=True):
    """Many instances of ONE plan over several GPUs: instances are independent, so every rank runs its slice with
    fhe_plan_run_batch (`run_batch(inputs[lo:hi]) -> outputs`; a plan finalised for world 1, keys replicated) and NO
    data-path collective is needed -- this is the form in which 8 GPUs help FheString::eq.  With
    gather=True the outputs are then all-gathered so that every rank returns all of them (control plane: a few ciphertexts
    per instance); gather=False returns the rank's own slice."""
    inputs = np.asarray(inputs, dtype=np.uint64)
    n = inputs.shape[0]
    lo, hi = instance_slice(n, rank, world)
    mine = run_batch(inputs[lo:hi]) if hi > lo else None
    if world == 1 or not gather:
        return mine
    import torch
    import torch.distributed as dist
    share = max(instance_slice(n, r, world)[1] - instance_slice(n, r, world)[0] for r in range(world))
    shape = None if mine is None else mine.shape[1:]
    shapes = [None] * world
    dist.all_gather_object(shapes, shape, group=group)
    shape = next(s for s in shapes if s is not None)
    padded = np.zeros((share,) + tuple(shape), dtype=np.uint64)
    if mine is not None:
        padded[: hi - lo] = mine
    everyone = torch.empty((world * share,) + tuple(shape), dtype=torch.int64)
    dist.all_gather_into_tensor(everyone, torch.from_numpy(padded.view(np.int64)), group=group)
    everyone = everyone.numpy().view(np.uint64).reshape((world, share) + tuple(shape))
    return np.concatenate([everyone[r, : instance_slice(n, r, world)[1] - instance_slice(n, r, world)[0]] for r in range(world)])


class GpuBackend:
    """Product backend: pool in HBM (torch tensor), HIP kernels via the C ABI, RCCL all-gather.
    staged=True gathers through host memory instead (gloo group: rehearsing several ranks on one GPU)."""

    def __init__(self, plan, device, group=None, staged=False, reuse_pool=True):
        import torch
        self.torch = torch
        self.plan = plan
        self.device = device
        self.group = group
        self.staged = staged
        self.reuse_pool = reuse_pool     # False: every alloc_pool call returns its own tensor (several simulated ranks)
        self.big = plan.params.big_size
        # kernels and collectives on the same stream: no host synchronisation between them
        plan.engine.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def alloc_pool(self, slots):
        # one pool per backend, reused across runs (a fresh 100+ MB allocation per operation costs milliseconds when the
        # caching allocator has to go back to the driver); zeroed each time like a fresh one
        if not self.reuse_pool:
            return self.torch.zeros((slots, self.big), dtype=self.torch.int64, device=self.device)
        pool = getattr(self, "_pool", None)
        if pool is None or pool.shape[0] != slots:
            pool = self._pool = self.torch.empty((slots, self.big), dtype=self.torch.int64, device=self.device)
        pool.zero_()
        return pool

    def load_inputs(self, pool, inputs, n_inputs):
        if isinstance(inputs, self.torch.Tensor):      # already resident in HBM (int64 view of the u64 words)
            pool[:n_inputs].copy_(inputs.reshape(n_inputs, self.big))
            return
        arr = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(n_inputs, self.big)
        pool[:n_inputs].copy_(self.torch.from_numpy(arr.view(np.int64)))

    def run_level(self, pool, level, rank):
        self.plan.run_level_rank_dev(pool.data_ptr(), level, rank)

    def all_gather(self, pool, local_base, e_max, recv_base, world):
        import torch.distributed as dist
        if self.staged:
            self.torch.cuda.current_stream(self.device).synchronize()
            mine = pool[local_base: local_base + e_max].cpu()
            everyone = self.torch.empty((e_max * world, self.big), dtype=mine.dtype)
            dist.all_gather_into_tensor(everyone, mine, group=self.group)
            pool[recv_base: recv_base + e_max * world].copy_(everyone)
            return
        dist.all_gather_into_tensor(pool[recv_base: recv_base + e_max * world],
                                    pool[local_base: local_base + e_max], group=self.group)

    def gather_outputs(self, pool, n_outputs, to_host=True):
        out = self.torch.empty((n_outputs, self.big), dtype=self.torch.int64, device=self.device)
        self.plan.gather_outputs_dev(pool.data_ptr(), out.data_ptr())
        stream = self.torch.cuda.current_stream(self.device)
        if not to_host:
            stream.synchronize()
            self._check_engine()
            return out
        if n_outputs * self.big * 8 < (4 << 20):
            stream.synchronize()
            self._check_engine()
            return out.cpu().numpy().view(np.uint64)
        # large outputs (to_lower / replace on 1024 chars: 67 MB): a pageable download runs at 3 GB/s; stage through a
        # page-locked buffer kept by the backend (full PCIe rate) and hand out a copy
        host = getattr(self, "_host_out", None)
        if host is None or host.shape[0] != n_outputs:
            host = self._host_out = self.torch.empty((n_outputs, self.big), dtype=self.torch.int64, pin_memory=True)
        host.copy_(out, non_blocking=True)
        stream.synchronize()
        self._check_engine()
        return host.numpy().view(np.uint64).copy()

    def _check_engine(self):
        """Before results leave: the multi-CU blind-rotation kernels (N >= 16384) never hang on a hand-over that does not
        come (a foreign kernel holding CUs), they finish with garbage and raise a sticky status -- fhe_engine_synchronize
        reads it and raises (this backend only synchronised torch's stream and returned such outputs silently)."""
        self.plan.engine.synchronize()
