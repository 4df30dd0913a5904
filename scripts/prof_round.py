#!/usr/bin/env python3
"""Collect the rocprofv3 evidence for one kernel revision on the GPU box (run through gpurun):

    python3 scripts/prof_round.py <tag> [--p44]

  1. `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`      -> <tag>_kernel_stats.csv, <tag>_bench.json
  2. one `rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py ...` pass per counter set
     (never combined with other trace domains; FETCH_SIZE and WRITE_SIZE in separate passes:
     MI355X_MICROARCH.md, "rocprofv3 PMC slots")                      -> <tag>_pmc_<set>.txt
  3. <tag>_counters.json: per-kernel, per-launch averages; FETCH_SIZE doubled (gfx950 tallies the
     128-B requests of 16-B/lane streaming loads at 64 B), WRITE_SIZE as is.

Everything lands in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
This script itself never touches the GPU (the profiled program is always `python3 bench.py` / a script
directly behind `--`)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {
    "a": "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS",
    "b": "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "f": "FETCH_SIZE",
    "w": "WRITE_SIZE",
    "h": "TCC_HIT_sum TCC_MISS_sum",
}
SKIP = ("fillBuffer", "at::native", "copyBuffer", "elementwise", "rocclr")


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(open(log).read()[-3000:])
        raise SystemExit(f"failed: {' '.join(cmd)}")


def short(name):
    for k in ("blind_rotate_xcd_kernel", "blind_rotate_cluster_kernel", "blind_rotate_seq_kernel", "blind_rotate_large_kernel", "blind_rotate_wide_kernel",
              "blind_rotate_multibit_kernel", "blind_rotate_kernel", "keyswitch_mfma_kernel", "ks_decompose_kernel",
              "keyswitch_dot4_kernel", "lincomb_kernel"):
        if k in name:
            return k
    return name[:60]


def main():
    tag = sys.argv[1]
    p44 = "--p44" in sys.argv
    out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    os.makedirs(out, exist_ok=True)
    custom = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--prog=")), None)
    batch = int(next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--batch=")), 256))
    if custom:         # any other program of this repo, e.g. --prog="python3 scripts/param_sweep.py 256 8192 5_CARRY_1"
        prog = custom.split()
        stats_prog = prog
    elif p44:
        prog = ["python3", "scripts/p44_prof.py"]
        stats_prog = prog
    else:
        common = ["--no-cpu-baseline", "--no-sweep", "--no-strings", "--no-p44"]
        prog = ["python3", "bench.py", "--steps", "3", "--warmup", "1"] + common
        stats_prog = ["python3", "bench.py", "--steps", "40", "--warmup", "5"] + common
    # 1. kernel-trace + stats
    tdir = os.path.join(out, "trace_stats")
    shutil.rmtree(tdir, ignore_errors=True)
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tdir, "-o", "run", "--"] + stats_prog,
        os.path.join(out, f"{tag}_bench.json"))
    bj = os.path.join(out, f"{tag}_bench.json")
    lines = [l for l in open(bj) if l.startswith('{"metric"')]
    if lines:   # keep only bench.py's own line (rocprofv3 logs to the same stream)
        open(bj, "w").write(lines[-1])
    for f in glob.glob(tdir + "/**/*kernel_stats.csv", recursive=True)[:1]:
        shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats.csv"))
    # 2. PMC passes
    per_kernel = collections.defaultdict(dict)
    for name, ctrs in SETS.items():
        tdir = os.path.join(out, f"trace_{name}")
        shutil.rmtree(tdir, ignore_errors=True)
        run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", tdir, "-o", "run", "--"] + prog,
            os.path.join(out, f"pmc_{name}.log"))
        files = glob.glob(tdir + "/**/*counter_collection.csv", recursive=True)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        with open(os.path.join(out, f"{tag}_pmc_{name}.txt"), "w") as f:
            for k, d in agg.items():
                if any(s in k for s in SKIP):
                    continue
                f.write(k[:100] + "\n")
                for c, v in sorted(d.items()):
                    n = cnt[(k, c)]
                    f.write(f"   {c:28s} per-launch {v / n:18.1f}  (launches {n})\n")
                    per_kernel[short(k)][c] = v / n
                    per_kernel[short(k)]["launches_" + name] = n
    # 3. consolidated json
    rev = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, 'fhe-string-bounty_amd'); import fhestr; print(fhestr.kernel_revision())"],
                         cwd=ROOT, capture_output=True, text=True).stdout.strip()
    res = {"_comment": "per-launch averages from rocprofv3 --pmc, one pass per counter set (scripts/prof_round.py); "
                       "traffic_bytes_per_launch = 2 x FETCH_SIZE (KB, gfx950 correction for 16-B/lane streaming "
                       "loads, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE (KB); fabric-side bytes incl. Infinity-Cache hits",
           "kernel_revision": rev, "command": " ".join(prog), "p44_command": " ".join(prog) if p44 else ""}
    for k, d in per_kernel.items():
        e = dict(d)
        e["batch"] = batch
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            e["traffic_bytes_per_launch"] = int(2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in d:
            e["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1.0)
        res[k] = e
    with open(os.path.join(out, f"{tag}_counters.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read()[:1500])
    print(json.dumps({k: v for k, v in res.items() if k.startswith(("blind", "keyswitch"))}, indent=1)[:4000])


if __name__ == "__main__":
    main()
