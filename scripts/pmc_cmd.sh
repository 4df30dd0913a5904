#!/bin/bash
# usage: scripts/pmc_cmd.sh <tag> "<counters>" <python script> [args...] -- one PMC pass (kernel-trace only) of any script
set -e
TAG=$1; shift
CTRS=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/trace -o run -- python3 "$@" > $OUT/run.log 2> $OUT/run.err || { tail -20 $OUT/run.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "fillBuffer" in k or "at::native" in k or "copyBuffer" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print(f"   {c:28s} per-launch {v / n:16.1f}  (launches {n})")
PY
cat $OUT/run.log
