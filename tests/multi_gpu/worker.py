"""One rank of tests/test_gpu_multi_gpu.py: a real RCCL group, one GPU per process.

Launched by torch.distributed.run before anything touched a GPU.  Every rank generates the same server keys on its
own device (deterministic ChaCha20 streams from one seed), runs FheString eq / contains / find through the sharded
plan runner with the product backend (HIP kernels + all_gather_into_tensor over xGMI) and, on its own, the same
operation as a one-rank plan; both must decrypt to the Python `bytes` answer and to each other.  Rank 0 writes the
verdict as JSON."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    out_path = sys.argv[1]
    import torch
    import torch.distributed as dist
    import fhestr
    from fhestr.distributed import GpuBackend, ShardedPlanRunner

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
    ck = fhestr.ClientKey(P, 0x5EED0007)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, local)
    eng.generate_keys(g, s, 0x5EED0007)
    rng = np.random.default_rng(7)            # same inputs on every rank
    hay = bytes(rng.integers(0x61, 0x7B, size=64, dtype=np.uint8))
    pat = hay[21:29]
    other = bytearray(hay); other[40] ^= 1; other = bytes(other)
    enc = lambda b, cap: ck.encrypt(fhestr.string_to_blocks(P, b, cap))
    cases = [
        ("eq", 64, 64, np.concatenate([enc(hay, 64), enc(hay, 64)]), lambda d: int(d[0]) == 1),
        ("eq", 64, 64, np.concatenate([enc(hay, 64), enc(other, 64)]), lambda d: int(d[0]) == 0),
        ("contains", 64, 8, np.concatenate([enc(hay, 64), enc(pat, 8)]), lambda d: int(d[0]) == 1),
        ("find", 64, 8, np.concatenate([enc(hay, 64), enc(pat, 8)]),
         lambda d: int(d[0]) == 1 and sum(int(v) * P.msg_mod ** i for i, v in enumerate(d[1:])) == hay.find(pat)),
    ]
    verdict = {"world": world, "cases": []}
    dev = torch.device("cuda", local)
    ok_all = True
    for op, a_cap, b_cap, inputs, check in cases:
        plan_w = fhestr.Plan.string_op(eng, op, a_cap, b_cap, world=world)
        runner = ShardedPlanRunner(plan_w, rank, world, GpuBackend(plan_w, dev))
        t0 = time.perf_counter()
        sharded = ck.decrypt(runner.run(inputs))
        ms = (time.perf_counter() - t0) * 1e3
        plan_1 = fhestr.Plan.string_op(eng, op, a_cap, b_cap, world=1)
        single = ck.decrypt(ShardedPlanRunner(plan_1, 0, 1, GpuBackend(plan_1, dev)).run(inputs))
        ok = bool(check(sharded)) and bool(np.array_equal(sharded, single))
        flags = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)           # every rank agrees
        ok_all = ok_all and bool(flags.item())
        verdict["cases"].append({"op": op, "ok_all_ranks": bool(flags.item()), "collectives": runner.collectives,
                                 "gathered_bytes_per_rank": runner.gathered_bytes, "first_run_ms_rank0": ms})
        plan_w.close(); plan_1.close()
    eng.set_stream(None)
    verdict["ok"] = ok_all
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()
    sys.exit(0 if ok_all else 1)


if __name__ == "__main__":
    main()
