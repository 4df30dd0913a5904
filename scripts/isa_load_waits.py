#!/usr/bin/env python3
"""Static check of the blind-rotation kernels' main loops for serialised memory loads (no GPU needed).

Compiles csrc/engine.hip to gfx950 assembly and, for every blind_rotate_* kernel, finds its largest loop and counts global /
buffer loads, scratch (spill) loads, `s_waitcnt vmcnt` instructions, and how many of those waits drain a queue of at most two
loads -- the signature of "load, wait, use" chains that pay the L2 latency once per load.  Round 4 found an N = 1024 build whose
36 key loads per step had come out that way (2x slower); the largest loop of the multi-CU kernels is their per-LWE loop, whose
initialisation shows up here too.

    python3 scripts/isa_load_waits.py [extra hipcc flags ...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "engine.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                "-o", out, os.path.join(ROOT, "fhe-string-bounty_amd", "csrc", "engine.hip")] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
kern, cur = {}, None
for l in open(out):
    l = l.rstrip("\n")
    m = re.match(r"^(_ZN3fhe\w+):", l)
    if m:
        cur = m.group(1); kern[cur] = []
    elif cur is not None:
        kern[cur].append(l)
        if "s_endpgm" in l:
            cur = None
rows = []
for name, lines in kern.items():
    if "blind_rotate" not in name:
        continue
    lines = [l for l in lines if not l.strip().startswith(";")]
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[0]):
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    if not best:
        continue
    span, a, b = best
    loads = scratch = waits = lone = outstanding = 0
    for l in lines[a:b + 1]:
        t = l.strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith(("global_load", "buffer_load", "flat_load")):
            loads += 1; outstanding += 1
        elif op.startswith("scratch_load"):
            scratch += 1; outstanding += 1
        elif op == "s_waitcnt" and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1)); waits += 1
            if n == 0 and 0 < outstanding <= 2:
                lone += 1
            outstanding = min(outstanding, n)
    rows.append((lone, re.sub(r"^_ZN3fhe\d+", "", name)[:72], span, loads, scratch, waits))
print(f"{'kernel':72s} {'loop':>6s} {'loads':>6s} {'spill':>6s} {'waits':>6s} {'lone':>5s}")
for lone, name, span, loads, scratch, waits in sorted(rows, reverse=True):
    print(f"{name:72s} {span:6d} {loads:6d} {scratch:6d} {waits:6d} {lone:5d}")
