#!/usr/bin/env python3
"""A/B the blind-rotation kernel across builds of libfhestr.so on the GPU box (AB_ARGS: extra bench.py flags;
a build name may carry its own flags after a '+', e.g. default+--no-pipeline):

    python3 scripts/ab_bench.py build/ab/libA.so build/ab/libB.so ...   ("default" = the in-tree build)

Each build runs `bench.py --steps 20 --warmup 3` (headline only) in its own process, alternating twice so
clock drift shows; prints the kernel ms and PBS/s of every run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["default"]
extra = os.environ.get("AB_ARGS", "").split()
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in libs:
        env = dict(os.environ)
        lib, _, own = lib.partition("+")
        own = own.split() if own else []
        if lib != "default":
            env["FHESTR_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-sweep",
                            "--no-strings", "--no-p44"] + extra + own, cwd=ROOT, env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
        if not line:
            print(lib, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)
            continue
        j = json.loads(line[-1])
        print(f"{lib + ' ' + ' '.join(own):40s} round {rnd}: {j['ms_per_step']:.4f} ms/step  blind_rotate {j['kernel_ms']['blind_rotate']:.4f} ms  keyswitch {j['kernel_ms']['keyswitch']:.4f} ms  "
              f"{j['value']:.0f} PBS/s  verified {j['verified_decrypt']}", flush=True)
