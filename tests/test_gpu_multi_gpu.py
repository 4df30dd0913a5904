"""SURVEY.md 8(e) on real hardware: when the box shows two or more GPUs, one fresh process per GPU runs FheString
eq / contains / find through the sharded plan runner with the RCCL all-gather (backend "nccl" over xGMI) and every
rank is checked against the one-rank result.  Skips on a one-GPU box (the gloo world-2/4 CPU tests and the
all-ranks-on-one-GPU test cover the sharding protocol there).  The reference has no multi-GPU path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_string_ops_over_rccl(tmp_path):
    import torch
    n = torch.cuda.device_count()       # does not initialise the GPU on this image
    if n < 2:
        pytest.skip(f"{n} GPU visible: the RCCL path needs at least 2")
    world = min(n, 4)
    out = tmp_path / "verdict.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", "29617",
           os.path.join(ROOT, "tests", "multi_gpu", "worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    verdict = json.loads(out.read_text())
    assert verdict["ok"] and verdict["world"] == world
    assert all(c["ok_all_ranks"] for c in verdict["cases"])
    # eq exchanges only reduced blocks: at most two all-gathers of one ciphertext per rank
    eq = verdict["cases"][0]
    assert eq["collectives"] <= 2
