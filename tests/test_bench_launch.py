"""`python bench.py --gpus N` must start its own ranks (VERDICT r3 "missing" 1): the launcher half of bench.py runs here
on CPU -- command construction, a real torch.distributed.run of two stand-in ranks, exit-code propagation.  The parent
must not import torch or the library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_rank.py")


def test_launch_command_shape():
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "5", "--print-launch"], 4, 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")


def test_print_launch_does_not_touch_torch():
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '8', '--steps', '3', '--print-launch']; import bench; bench.main(); "
            "assert 'torch' not in sys.modules and 'fhestr' not in sys.modules")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "3"]


def _self_launch(extra_env):
    code = ("import sys, bench; rc = bench.self_launch(['--gpus', '2', '--steps', '5'], 2, script=%r); "
            "assert 'torch' not in sys.modules; sys.exit(rc)" % FAKE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_relays_rank0_line():
    out = _self_launch({})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["argv"] == ["--gpus", "2", "--steps", "5"] and rec["master"] == "127.0.0.1"


def test_self_launch_propagates_a_failing_rank():
    out = _self_launch({"FAKE_RANK_FAIL": "1"})
    assert out.returncode != 0
