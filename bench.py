#!/usr/bin/env python3
"""bench.py -- PBS/sec of the batched keyswitch + programmable bootstrap on MI355X.

Workload (BASELINE.json configs[1]): B = 256 independent shortint LWE ciphertexts,
PARAM_MESSAGE_2_CARRY_2_KS_PBS, per-LWE lookup tables, one GPU.  A "step" is one pass of the hot
path (memset -> keyswitch kernel -> blind-rotate kernel) over that batch with inputs, keys and
tables already resident in HBM.  With N GPUs every rank runs its own 256-LWE batch on its own
GPU (independent units, replicated keys, no data-path collective): weak scaling, value = all
LWEs of all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: bench.py starts that torch.distributed.run itself, as a fresh child
                                       process before anything touches the GPU, relays rank 0's line and its exit code)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
# FP64 vector peak: 256 CUs x 4 SIMDs x 16 f64 FMA lanes/clk x 2 FLOP x 2.4 GHz = half the guide's
# 157.3 TF FP32 vector rate (one wave64 f64 instruction occupies its SIMD for >= 4 cycles)
F64_VALU_PEAK_TFLOPS = 78.6
COUNTERS = os.path.join(ROOT, "profiles", "r04_counters.json")   # written by scripts/prof_round.py (this kernel revision)
COUNTERS_P44 = os.path.join(ROOT, "profiles", "r04_p44_counters.json")   # the same for `scripts/p44_prof.py 256` (config 5)
SEED = 0x5EED0002


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps before the timed ones (topped up to 40 untimed steps, see prewarm_steps)")
    ap.add_argument("--batch", type=int, default=256, help="LWEs per step per GPU")
    ap.add_argument("--log2-points", type=int, default=0, help="blind-rotate variant (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strings", action="store_true", help="skip the FheString ms/op section")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep")
    ap.add_argument("--no-p44", action="store_true", help="skip the PARAM_MESSAGE_4_CARRY_4 (config 5) section")
    ap.add_argument("--serial", action="store_true", help="headline without the keyswitch / blind-rotation pipelining")
    ap.add_argument("--print-launch", action="store_true", help="print the torch.distributed.run command --gpus N > 1 would start, and exit")
    return ap.parse_args(argv)


def launch_command(argv, gpus, port, script=None):
    """The command `bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts: one rank per GPU under
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    rest = [a for a in argv if a != "--print-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + rest


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, gpus, script=None):
    """`python bench.py --gpus N` without a launcher: start the ranks as a FRESH child process tree (this process has
    touched neither torch.cuda nor libfhestr, and never will), relay what they print, return the launcher's exit code.
    A rank that fails makes torch.distributed.run exit non-zero, and that is what the driver sees."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = launch_command(argv, gpus, free_port(), script)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:          # rank 0's JSON line (and nothing else) arrives on stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def cpu_baseline(P, bsk, ksk, cts, lut_tables, lut_sel):
    """Time the CPU oracle (a C port of the reference algorithm) on the same batch, all host cores.
    The ONLY place bench.py touches oracle/ -- as a reported baseline, never as the product."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import oracle as O
    op = O.Params(P.n, P.k, P.N, P.pbs_base_log, P.pbs_level, P.ks_base_log, P.ks_level,
                  P.msg_mod, P.carry_mod, P.lwe_std, P.glwe_std, P.name)
    # All the host CPU this job may use (the reference's throughput bench spreads over every core,
    # benches/core_crypto/pbs_bench.rs:517-532).  nproc is not that number on a shared box: the GPU pool gives a one-GPU
    # job a 16-CPU share of a 256-thread host, and 256 threads on a 16-CPU quota run 2x SLOWER than 16 (measured:
    # 187 vs 346 PBS/s).  So: read the cgroup quota, time a short sample at the candidate thread counts and report the
    # fastest, with the whole scan next to it.
    nproc = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = nproc
    if os.environ.get("FHESTR_CPU_BASELINE_THREADS"):
        candidates = [max(1, int(os.environ["FHESTR_CPU_BASELINE_THREADS"]))]
    else:
        candidates = sorted({c for c in (16, 32, 64, 128, int(quota) if quota else 0, affinity, nproc) if 1 <= c <= nproc})
    try:   # rebuild the checker for this host's ISA (AVX-512 where present); falls back to the shipped build
        O.build(force=True, arch="native")
    except Exception:
        pass
    L = O.lib()
    fbsk = np.zeros(bsk.size, dtype=np.float64)
    L.orc_bsk_to_fourier(C.byref(op.c()), bsk, fbsk)
    luts = np.zeros((len(lut_tables), op.glwe_len), dtype=np.uint64)
    for i, t in enumerate(lut_tables):
        L.orc_fill_accumulator(C.byref(op.c()), np.ascontiguousarray(t, dtype=np.uint64), luts[i])
    n_batch = cts.shape[0]
    idx_batch = np.ascontiguousarray(np.asarray(lut_sel, dtype=np.uint32))
    scan = {}
    for c in candidates:       # short sample: one LWE per thread, at least 64
        m = min(n_batch, max(64, c))
        t0 = time.perf_counter()
        L.orc_ks_pbs_batch(C.byref(op.c()), ksk, fbsk.ctypes.data_as(C.c_void_p), None, 0, np.ascontiguousarray(cts[:m]),
                           idx_batch[:m].ctypes.data_as(C.c_void_p), luts, np.zeros_like(cts[:m]), m, c)
        scan[c] = m / (time.perf_counter() - t0)
    cores = max(scan, key=scan.get)
    tiles = max(1, -(-2 * cores // n_batch))
    sample = np.ascontiguousarray(np.tile(cts, (tiles, 1)))
    idx = np.ascontiguousarray(np.tile(idx_batch, tiles))
    big_out = np.zeros_like(sample)
    t0 = time.perf_counter()
    L.orc_ks_pbs_batch(C.byref(op.c()), ksk, fbsk.ctypes.data_as(C.c_void_p), None, 0, sample,
                       idx.ctypes.data_as(C.c_void_p), luts, big_out, sample.shape[0], cores)
    dt = time.perf_counter() - t0
    out = big_out[:n_batch]
    # also one thread alone (ms per KS+PBS), and which CPU this was
    one = np.zeros_like(cts[:8])
    t0 = time.perf_counter()
    L.orc_ks_pbs_batch(C.byref(op.c()), ksk, fbsk.ctypes.data_as(C.c_void_p), None, 0, np.ascontiguousarray(cts[:8]),
                       idx[:8].ctypes.data_as(C.c_void_p), luts, one, 8, 1)
    single_ms = (time.perf_counter() - t0) / 8 * 1e3
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": sample.shape[0] / dt, "unit": "PBS/s", "cores": cores, "kind": "port",
            "single_thread_ms_per_pbs": single_ms, "cpu_model": model, "nproc": nproc, "cpu_quota": quota, "affinity": affinity,
            "pbs_per_s_by_threads": {str(k): round(v, 1) for k, v in scan.items()},
            "sample": f"the same {n_batch}-LWE batch tiled x{tiles} ({sample.shape[0]} KS+PBS, about "
                      f"{sample.shape[0] * single_ms / 1e3:.0f} CPU-seconds), one LWE per task over {cores} host threads = the fastest "
                      f"of the thread counts scanned (pbs_per_s_by_threads; the job's CPU share on this box is what limits it, "
                      f"not the {nproc} hardware threads the host shows) (oracle/tfhe_oracle.c, gcc -O3 -march=native; its plain "
                      f"radix-4 FFT is ~2.7x slower per core than the reference's published 16.6 ms/PBS on a Xeon 8375C)",
            "seconds": dt}, out


def bench_strings(fhestr, eng, ck, P, rank, world, local_rank, reps=3, staged=False):
    """ms/op of FheString::eq (256 vs 256 chars, both encrypted) and ::contains (16-char encrypted
    pattern in a 256-char haystack); every level's KS+PBS batch is split over the ranks."""
    import torch
    from fhestr.distributed import GpuBackend, ShardedPlanRunner
    rng = np.random.default_rng(0x5EED0003)
    hay = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
    off = int(rng.integers(0, 240))
    pat = hay[off: off + 16]
    enc = lambda s, cap: ck.encrypt(fhestr.string_to_blocks(P, s, cap))
    differ = bytearray(hay)
    differ[int(rng.integers(0, 256))] ^= 1
    differ = bytes(differ)
    text = bytes(rng.choice(np.frombuffer(b"The quick brown fox jumps over the lazy dog and the cat", dtype=np.uint8), size=1000))
    text = text.replace(b"the ", b"xyz ") + b" the end of the line"
    bit = lambda want: (lambda d: int(d[0]) == want)
    index = lambda want: (lambda d: int(d[0]) == 1 and sum(int(v) * P.msg_mod ** i for i, v in enumerate(d[1:])) == want)
    chars = lambda want, cap: (lambda d: fhestr.blocks_to_string(P, d)[:cap].rstrip(b"\0") == want)
    # (op, a_cap, b_cap, clear pattern, encrypted inputs, check of the decrypted outputs, timed repetitions)
    cases = {
        # config 3: eq / ne on 256-char strings, pattern encrypted and clear, equal and differing in one position
        "eq_256_enc_enc": ("eq", 256, 256, None, np.concatenate([enc(hay, 256), enc(hay, 256)]), bit(1), reps),
        "eq_256_enc_enc_differ": ("eq", 256, 256, None, np.concatenate([enc(hay, 256), enc(differ, 256)]), bit(0), 1),
        "ne_256_enc_enc": ("ne", 256, 256, None, np.concatenate([enc(hay, 256), enc(differ, 256)]), bit(1), reps),
        "eq_256_enc_clear": ("eq_clear", 256, 0, hay, enc(hay, 256), bit(1), reps),
        "ne_256_enc_clear": ("ne_clear", 256, 0, differ, enc(hay, 256), bit(1), reps),
        # config 4: 16-char encrypted pattern in a 256-char haystack
        "contains_16_in_256_enc_enc": ("contains", 256, 16, None, np.concatenate([enc(hay, 256), enc(pat, 16)]), bit(1), reps),
        "contains_16_in_256_enc_enc_absent": ("contains", 256, 16, None,
                                              np.concatenate([enc(hay, 256), enc(b"0123456789ABCDEF", 16)]), bit(0), 1),
        "find_16_in_256_enc_enc": ("find", 256, 16, None, np.concatenate([enc(hay, 256), enc(pat, 16)]), index(hay.find(pat)), 2),
        # config 5 on this parameter set (N = 2048; the N = 32768 run is the "p44" section)
        "to_lower_1024": ("to_lower", 1024, 0, None, enc(text, 1024), chars(text.lower(), 1024), 2),
        "replace_clear_4_in_1024": ("replace_clear", 1024, 0, b"the THAT", enc(text, 1024),
                                    chars(text.replace(b"the ", b"THAT"), 1024), 2),
        # the same with an ENCRYPTED 4-character pattern and replacement: the occurrences need a scan over the offsets
        # (fhe_string.cpp: occurrences_scan -- blocked, 2 B + n / B lookup levels instead of n / 2)
        "replace_enc_4_in_1024": ("replace", 1024, 8, None, np.concatenate([enc(text, 1024), enc(b"the ", 4), enc(b"THAT", 4)]),
                                  chars(text.replace(b"the ", b"THAT"), 1024), 2),
    }
    out = {}
    dev = torch.device("cuda", local_rank)
    for name, (op, a_cap, b_cap, clear, inputs, check, n_rep) in cases.items():
        plan = fhestr.Plan.string_op(eng, op, a_cap, b_cap, clear=clear, world=world)
        runner = ShardedPlanRunner(plan, rank, world, GpuBackend(plan, dev, staged=staged))
        res = runner.run(inputs)   # warm-up + correctness
        ok = bool(check(ck.decrypt(res)))
        d_inputs = torch.from_numpy(inputs.view(np.int64)).to(dev)
        timings = {}
        # resident: inputs and outputs in HBM (the plan's outputs feed the caller's next operation; up to round 4's first bench
        # line this leg downloaded them, 11 of to_lower_1024's 36.7 ms); from_host: host arrays in, host array out
        for label, src in (("resident", d_inputs), ("from_host", inputs)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_rep):
                last = runner.run(src, device_outputs=(label == "resident"))
            torch.cuda.synchronize()
            timings[label] = (time.perf_counter() - t0) / n_rep * 1e3
            if label == "resident":
                ok = ok and bool(check(ck.decrypt(last.cpu().numpy().view(np.uint64))))
        ms = timings["resident"]
        info = plan.info()
        out[name] = {"ms_per_op": ms, "ms_per_op_inputs_from_host": timings["from_host"], "n_pbs": info["n_pbs"],
                     "levels": info["n_levels"], "correct": bool(ok), "pbs_per_s": info["n_pbs"] / (ms * 1e-3),
                     # what the ranks exchange: only ciphertexts another rank consumes
                     "sharded_over": world, "collectives_per_op": runner.collectives,
                     "gathered_lwes_per_rank": runner.gathered_lwes, "gathered_bytes_per_rank": runner.gathered_bytes,
                     "worst_pbs_input_noise": plan.noise_info()["max_pbs_input_noise"],
                     "noise_budget": plan.noise_info()["budget"]}
        plan.close()
    if world == 1:
        # latency knob, off by default and off for every figure above: replicas on the idle CUs during the small levels keep the
        # clock up for the next op's large first level (fhe_engine_set_keep_busy; results bit-identical, energy is the price)
        op, a_cap, b_cap, clear, inputs, check, n_rep = cases["eq_256_enc_enc"]
        plan = fhestr.Plan.string_op(eng, op, a_cap, b_cap, clear=clear, world=world)
        runner = ShardedPlanRunner(plan, rank, world, GpuBackend(plan, dev, staged=staged))
        d_inputs = torch.from_numpy(inputs.view(np.int64)).to(dev)
        eng.set_keep_busy(True)
        try:
            res = runner.run(d_inputs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                res = runner.run(d_inputs)
            torch.cuda.synchronize()
            out["eq_256_enc_enc_keep_busy"] = {"ms_per_op": (time.perf_counter() - t0) / 10 * 1e3,
                                               "ms_per_op_inputs_from_host": None, "n_pbs": plan.info()["n_pbs"],
                                               "levels": plan.info()["n_levels"], "correct": bool(check(ck.decrypt(res))),
                                               "note": "fhe_engine_set_keep_busy(1): opt-in, not the default figure"}
        finally:
            eng.set_keep_busy(False)
            plan.close()
    eng.set_stream(None)
    # ---- many strings per pass (fhe_plan_run_batch): level l of all instances is ONE launch.  A single eq pays four
    #      dependent single-PBS latencies on a nearly idle GPU; M of them share those four.  With several GPUs the
    #      instances shard over the ranks (every rank runs its own M here): no collective at all. ----
    try:
        import torch.distributed as dist
        batch_cases = [("eq_256_batch8", "eq", 256, 256, 8), ("eq_256_batch32", "eq", 256, 256, 32),
                       ("contains_16_in_256_batch8", "contains", 256, 16, 8)]
        for name, op, a_cap, b_cap, M in batch_cases:
            plan = fhestr.Plan.string_op(eng, op, a_cap, b_cap)
            rows, want = [], []
            for i in range(M):
                if op == "eq":
                    r = hay if i % 2 == 0 else bytes(bytearray(hay[:i]) + bytearray([hay[i] ^ 1]) + bytearray(hay[i + 1:]))
                    rows.append(np.concatenate([enc(r, a_cap), enc(hay, b_cap)]))
                    want.append(int(r == hay))
                else:
                    r = hay if i % 2 == 0 else hay[:off] + bytes([hay[off] ^ 1]) + hay[off + 1:]
                    rows.append(np.concatenate([enc(r, a_cap), enc(pat, b_cap)]))
                    want.append(int(pat in r))
            d_in = torch.from_numpy(np.stack(rows).view(np.int64)).to(dev)
            info = plan.info()
            d_out = torch.zeros((M, info["n_outputs"], P.big_size), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            n_rep = 3
            for it in range(1 + n_rep):
                if it == 1:
                    eng.synchronize()
                    if world > 1:
                        dist.barrier()
                    t0 = time.perf_counter()
                plan.run_batch_dev(d_in.data_ptr(), d_out.data_ptr(), M)
            eng.synchronize()
            dt = (time.perf_counter() - t0) / n_rep
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device="cpu" if staged else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            got = [int(v) for v in ck.decrypt(d_out.cpu().numpy().view(np.uint64)[:, 0])]
            out[name] = {"ms_per_op": dt * 1e3 / (M * world), "ms_per_pass": dt * 1e3, "instances_per_rank": M, "instances_sharded_over": world,
                         "collectives_per_op": 0, "n_pbs_per_op": info["n_pbs"], "levels": info["n_levels"],
                         "pbs_per_s": info["n_pbs"] * M * world / dt, "correct": got == want}
            plan.close()
            del d_in, d_out
    except Exception as e:      # secondary section
        out["batch_error"] = f"{type(e).__name__}: {e}"
    return out


def bench_multi_bit(fhestr, local_rank, B, steps):
    """Same 256-LWE step with the multi-bit PBS (PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS,
    lwe_multi_bit_programmable_bootstrapping.rs): different parameter sets of the reference, reported
    next to the headline, never as the headline.  Keys are generated on the device.  Also the small-batch
    latency (1 and 32 LWEs: the groups' GGSWs are prepared on the whole GPU first, fhe_engine_set_multibit_combine_max)
    and FheString::eq on 256-char strings, which is what the shorter serial chain is for."""
    import torch
    out = {}
    for P in (fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS, fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS):
        M = P.msg_mod * P.carry_mod
        ck = fhestr.ClientKey(P, SEED + 1)
        g, s = ck.secret_keys()
        eng = fhestr.Engine(P, local_rank)
        try:
            eng.generate_keys(g, s, SEED + 1)
            rng = np.random.default_rng(SEED + 1)
            table = rng.integers(0, M, size=M)
            lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
            rec = {"params": P.name, "grouping_factor": P.grouping}
            for nb in (B, 32, 1):
                msgs = rng.integers(0, M, size=nb)
                d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
                d_idx = torch.full((nb,), int(lut), dtype=torch.int32, device="cuda")
                d_out = torch.zeros_like(d_in)
                torch.cuda.synchronize()
                for _ in range(3):
                    eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), nb)
                eng.synchronize()
                eng.kernel_times(reset=True)
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), nb)
                eng.synchronize()
                dt = time.perf_counter() - t0
                ks_ms, br_ms, calls = eng.kernel_times(reset=True)
                ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
                r = {"batch": nb, "pbs_per_s": nb * steps / dt, "ms_per_step": dt / steps * 1e3,
                     "kernel_ms": {"keyswitch": ks_ms / max(calls, 1), "blind_rotate": br_ms / max(calls, 1)},
                     "verified_decrypt": ok}
                if nb == B:
                    rec.update(r)
                else:
                    rec[f"batch_{nb}"] = r
            ops = fhestr.FheStringOps(eng)
            text = bytes(rng.integers(0x61, 0x7B, size=256, dtype=np.uint8))
            ea, eb = (ck.encrypt(fhestr.string_to_blocks(P, text, 256)) for _ in range(2))
            bit = ops.eq(ea, eb)
            t0 = time.perf_counter()
            for _ in range(3):
                ops.eq(ea, eb)
            rec["fhestring_eq_256_ms"] = (time.perf_counter() - t0) / 3 * 1e3     # host buffers in and out
            rec["fhestring_eq_256_correct"] = int(ck.decrypt(np.asarray(bit).reshape(1, -1))[0]) == 1
            out[f"group_{P.grouping}"] = rec
        finally:
            eng.close()
    return out


def bench_n1024_k2(fhestr, local_rank):
    """PARAM_MESSAGE_2_CARRY_1_KS_PBS (N = 1024, k = 2: the other polynomial size BASELINE.json's north_star names): KS+PBS
    steps of 256 LWEs (one per CU), 512 (two per CU) and 2,048 (the dense kernel, four per CU), decrypt-checked."""
    import torch
    P = fhestr.PARAM_MESSAGE_2_CARRY_1_KS_PBS
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, SEED + 2)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, local_rank)
    try:
        eng.generate_keys(g, s, SEED + 2)
        rng = np.random.default_rng(SEED + 2)
        table = rng.integers(0, M, size=M)
        lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
        rec = {"params": P.name}
        for nb in (256, 512, 2048):
            msgs = rng.integers(0, M, size=nb)
            d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
            d_idx = torch.full((nb,), int(lut), dtype=torch.int32, device="cuda")
            d_out = torch.zeros_like(d_in)
            for it in range(8):
                if it == 3:
                    eng.synchronize()
                    eng.kernel_times(reset=True)
                    t0 = time.perf_counter()
                eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), nb)
            eng.synchronize()
            dt = (time.perf_counter() - t0) / 5
            ks_ms, br_ms, calls = eng.kernel_times(reset=True)
            ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
            rec[f"batch_{nb}"] = {"pbs_per_s": nb / dt, "ms_per_step": dt * 1e3, "ms_per_256_lwes": dt * 1e3 * 256 / nb,
                                  "kernel_ms": {"keyswitch": ks_ms / max(calls, 1), "blind_rotate": br_ms / max(calls, 1)},
                                  "verified_decrypt": ok}
        return rec
    finally:
        eng.close()


def flop_per_cmux_step(P):
    """Algorithmic f64 FLOP of one CMUX step (SURVEY 8(d) "secondary"; ggsw.rs:477-598): l(k+1) forward +
    (k+1) inverse size-N/2 complex FFTs at 5 n log2 n, 8 FLOP per complex multiply-accumulate of the
    l(k+1)^2 N/2 Fourier products, 6 FLOP per twist multiply.  P22: 262,144."""
    half, k1, L = P.N // 2, P.k + 1, P.pbs_level
    ffts = L * k1 + k1
    return ffts * 5 * half * (half.bit_length() - 1) + 8 * L * k1 * k1 * half + 6 * half * ffts


def rooflines(P, B, world, value, br_avg_ms, revision, log2_points, kernel="blind_rotate_kernel", concurrency=1):
    """The `roofline` object of the bench contract (HBM, SURVEY 8(d)'s per-LWE key-streaming model for the
    dominant kernel) and, next to it, what actually bounds that kernel: f64 VALU issue + LDS
    (`roofline`; SURVEY's model is `roofline_hbm_model`).  Counter-derived fields come from the committed rocprofv3 passes
    (profiles/r04_counters.json, one --pmc pass per counter set, scripts/prof_round.py) and are only
    attached when they were taken on this kernel revision, batch and variant."""
    br_bytes = P.bsk_len * 8 + P.glwe_len * 8 + P.small_size * 8 + P.big_size * 8   # BSK + LUT + LWE in/out
    achieved = concurrency * br_bytes * B / (br_avg_ms * 1e-3) / 1e9
    pbs_bytes = P.bsk_len * 8 + P.ksk_len * 8 + 2 * P.big_size * 8 + P.glwe_len * 8  # 109,559,824 (P22)
    compulsory = (P.bsk_len * 8 + P.ksk_len * 8) / B + 2 * P.big_size * 8 + P.small_size * 16
    ctr, src = None, None
    try:
        cj = json.load(open(COUNTERS))
        c = cj[kernel]
        if c["batch"] == B and log2_points == 0 and cj.get("kernel_revision") == revision:
            ctr, src = c, "profiles/r04_counters.json (static: rocprofv3 --pmc passes of this kernel revision, " + cj.get("command", "") + ")"
    except Exception:
        pass
    traffic = ctr["traffic_bytes_per_launch"] if ctr else None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "kernel": kernel, "avg_launch_ms": br_avg_ms,
            "algorithmic_bytes_per_lwe": br_bytes, "lwes_per_launch": B,
            "model": "per-LWE key streaming (SURVEY 8(d)): every LWE is charged the whole Fourier key",
            "note": "NOT the binding resource: the 48.6 MB key is shared by all workgroups and served from "
                    "L2 / Infinity Cache; see measured_hbm_* (fabric-side bytes) and `roofline`"}
    if traffic:
        roof["measured_hbm_gbs"] = traffic / (br_avg_ms * 1e-3) / 1e9
        roof["measured_hbm_frac"] = roof["measured_hbm_gbs"] / HBM_PEAK_GBS
        roof["l2_hit_rate"] = ctr.get("l2_hit_rate")
    flop_step = flop_per_cmux_step(P)
    n_steps = P.n if P.grouping <= 1 else P.n // P.grouping
    # `concurrency` launches of this kernel share the GPU (overlapped-batches mode: 2), each lasting br_avg_ms
    tflops = concurrency * B * n_steps * flop_step / (br_avg_ms * 1e-3) / 1e12
    comp = {"bound": "valu_f64", "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / F64_VALU_PEAK_TFLOPS, "flop_per_cmux_step": flop_step, "cmux_steps": n_steps,
            "kernel": kernel, "launches_sharing_the_gpu": concurrency,
            "model": "algorithmic f64 FLOP (5 n log2 n per FFT) / launch time vs the FP64 vector peak"}
    if ctr and "SQ_BUSY_CYCLES" in ctr:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; SQ_ACTIVE_INST_* / SQ_WAIT_* count
        # quad-cycles summed over all SIMDs (MI355X_MICROARCH.md, cycle constants)
        simd_cycles = ctr["SQ_BUSY_CYCLES"] / 32.0 * 256 * 4
        comp["valu_issue_frac"] = 4.0 * ctr["SQ_ACTIVE_INST_VALU"] / simd_cycles
        comp["lds_issue_frac"] = 4.0 * ctr["SQ_ACTIVE_INST_LDS"] / simd_cycles
        comp["lds_wait_frac_of_wave_cycles"] = ctr["SQ_WAIT_INST_LDS"] / ctr["SQ_WAVE_CYCLES"]
        comp["valu_insts_per_wave_step"] = ctr["SQ_INSTS_VALU"] / (ctr["SQ_WAVES"] * n_steps)
        comp["lds_insts_per_wave_step"] = ctr["SQ_INSTS_LDS"] / (ctr["SQ_WAVES"] * n_steps)
        comp["counters_source"] = src
        if concurrency > 1:
            # rocprofv3 --pmc serialises dispatches: the fractions above are those of ONE launch running alone (one wave per
            # SIMD for the two-LWEs-per-CU kernel).  Overlapped, `concurrency` launches put that many times the VALU-active
            # cycles into one launch duration; the SIMD clock under load is not read here, so both ends are given.
            active = 4.0 * ctr["SQ_ACTIVE_INST_VALU"] / (256 * 4)          # VALU-active cycles per SIMD and launch
            comp["valu_issue_frac_note"] = "counters of a launch running alone (PMC passes serialise dispatches)"
            comp["valu_busy_frac_overlapped_estimate"] = [concurrency * active / (br_avg_ms * 1e-3 * f) for f in (2.4e9, 2.1e9)]
    whole = {"bytes_per_pbs": pbs_bytes, "frac_of_peak": value * pbs_bytes / (world * HBM_PEAK_GBS * 1e9),
             "compulsory_bytes_per_pbs_at_this_batch": compulsory,
             "compulsory_frac_of_peak": value * compulsory / (world * HBM_PEAK_GBS * 1e9),
             "note": "SURVEY 8(d) charges BSK + KSK to every PBS; with a batch sharing the keys out of "
                     "L2 / Infinity Cache that model can exceed the HBM peak (frac_of_peak > 1 means "
                     "cache-served, not skipped work: verified_decrypt covers the timed output); the "
                     "batch-amortised compulsory traffic is the HBM-side floor"}
    # `roofline` = the resource that binds this kernel (f64 vector issue + the dependent LDS/barrier chain, DESIGN.md
    # section 3), with the measured HBM-side bytes as `traffic`; SURVEY 8(d)'s per-LWE key-streaming model is kept
    # as `roofline_hbm_model`, labelled as what it is
    comp["traffic"] = traffic
    comp["traffic_source"] = src
    comp["avg_launch_ms"] = br_avg_ms
    roof["bound"] = "hbm (SURVEY 8(d) model, not binding)"
    return {"roofline": comp, "roofline_hbm_model": roof, "whole_pbs_hbm_model": whole}


def bench_p44(fhestr, local_rank):
    """BASELINE.json config 5 on one GPU: PARAM_MESSAGE_4_CARRY_4_KS_PBS as the reference defines it
    (N = 32768, shortint/parameters/mod.rs:1063-1077; keys generated on the device): the 256-LWE KS+PBS
    step and FheString::to_lower / replace on a 1024-char string, decrypt-checked."""
    import torch
    P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
    M = P.msg_mod * P.carry_mod
    ck = fhestr.ClientKey(P, SEED + 5)
    g, sm = ck.secret_keys()
    eng = fhestr.Engine(P, local_rank)
    try:
        t0 = time.perf_counter()
        eng.generate_keys(g, sm, SEED + 5)
        keygen_s = time.perf_counter() - t0
        rng = np.random.default_rng(SEED + 5)
        table = rng.integers(0, M, size=M)
        lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
        B = 256
        msgs = rng.integers(0, M, size=B)
        d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
        d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)   # warm-up
        eng.synchronize()
        eng.kernel_times(reset=True)
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ks_ms, br_ms, calls = eng.kernel_times(reset=True)
        ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
        pbs_bytes = P.bsk_len * 8 + P.ksk_len * 8 + 2 * P.big_size * 8 + P.glwe_len * 8   # 3,919,314,960
        br_avg_ms = br_ms / max(calls, 1)
        clusters = eng.cluster_info()
        out = {"params": P.name, "batch": B, "device_keygen_s": keygen_s, "pbs_per_s": B / dt, "ms_per_step": dt * 1e3,
               "kernel_ms": {"keyswitch": ks_ms / max(calls, 1), "blind_rotate": br_avg_ms},
               "kernel": "blind_rotate_cluster_kernel" if clusters else "blind_rotate_large_kernel",
               "clusters_formed": clusters,
               "verified_decrypt": ok,
               "hbm_model": {"bytes_per_pbs": pbs_bytes, "frac_of_peak": B / dt * pbs_bytes / (HBM_PEAK_GBS * 1e9),
                             "note": "SURVEY 8(d) per-LWE key-streaming model; bound 2.04 k PBS/s per GPU"}}
        # small batches: one LWE is worked on by a cluster of CUs, so latency no longer equals the 256-LWE step
        lat = {}
        for nb in (1, 16, 32):     # 1 and 16: the whole-XCD kernel (two LWEs in flight per XCD); 32 and above: the 8-CU clusters
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), nb)
            eng.synchronize()
            eng.kernel_times(reset=True)
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), nb)
            eng.synchronize()
            k1, b1, c1 = eng.kernel_times(reset=True)
            lat[str(nb)] = {"keyswitch_ms": k1 / max(c1, 1), "blind_rotate_ms": b1 / max(c1, 1)}
        out["small_batch_kernel_ms"] = lat
        # roofline of the dominant kernel from the committed counters of this kernel revision: the bytes that crossed
        # the L2 <-> fabric boundary (FETCH_SIZE doubled for 16-byte-per-lane reads per MI355X_MICROARCH.md, WRITE_SIZE
        # as is, separate --pmc passes) against the HBM peak; algorithmic bytes = the cluster's exchange matrices
        # (4 MiB per LWE-step: 2 out + 2 back) + the 2 MiB GGSW once per XCD, step and round of clusters (256 LWEs on 32
        # clusters: 8 rounds)
        try:
            cj = json.load(open(COUNTERS_P44))
            c = cj.get(out["kernel"])
            if c and c.get("batch") == B and cj.get("kernel_revision") == fhestr.kernel_revision():
                steps = P.n
                rounds = -(-B // max(clusters, 1))
                algo = B * steps * 4 * (1 << 20) + steps * 8 * rounds * (P.bsk_len * 8 // P.n)
                out["roofline"] = {"bound": "hbm", "kernel": out["kernel"], "avg_launch_ms": br_avg_ms,
                                   "algorithmic_bytes_per_launch": algo,
                                   "achieved": algo / (br_avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": algo / (br_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": c["traffic_bytes_per_launch"], "l2_hit_rate": c.get("l2_hit_rate"),
                                   "traffic_gbs": c["traffic_bytes_per_launch"] / (br_avg_ms * 1e-3) / 1e9,
                                   "traffic_source": "profiles/r04_p44_counters.json (static rocprofv3 --pmc passes, " + cj.get("p44_command", "") + ")"}
        except Exception:
            pass
        # config 5: 1024-char string, to_lower and replace (4-char clear pattern), one call each.  268 MB go each way:
        # host buffers from fhe_host_alloc (page-locked), allocated before the clock starts; to_lower is also timed
        # with pageable numpy arrays, the way every other host figure of this file is taken
        words = [b"The ", b"quick ", b"BROWN ", b"fox ", b"Jumps ", b"over ", b"the ", b"LAZY ", b"dog. "]
        s = b"".join(words[int(i)] for i in rng.integers(0, len(words), size=400))[:1000]
        es_pageable = ck.encrypt(fhestr.string_to_blocks(P, s, 1024))
        es = fhestr.pinned_empty(es_pageable.shape)
        es[...] = es_pageable
        locked = {}

        def locked_out(shape):
            key = tuple(int(d) for d in shape)
            if key not in locked:
                locked[key] = fhestr.pinned_empty(key)
            return locked[key]

        locked_out(es.shape)
        e_from, e_to = ck.encrypt(fhestr.string_to_blocks(P, b"the ", 4)), ck.encrypt(fhestr.string_to_blocks(P, b"THAT", 4))
        ops = fhestr.FheStringOps(eng, out_alloc=locked_out)
        dec = lambda ct: fhestr.blocks_to_string(P, ck.decrypt(ct))
        strings = {}
        for name, fn, want, plan_args in (
                ("to_lower_1024", lambda: ops.to_lower(es), s.lower(), ("to_lower", 1024, 0, None)),
                ("replace_clear_4_in_1024", lambda: ops.replace(es, b"the ", b"THAT"), s.replace(b"the ", b"THAT"),
                 ("replace_clear", 1024, 0, b"the THAT")),
                # encrypted pattern and replacement: the blocked occurrence scan, blocks of 64 offsets on this parameter set
                ("replace_enc_4_in_1024", lambda: ops.replace(es, e_from, e_to), s.replace(b"the ", b"THAT"), ("replace", 1024, 8, None))):
            t0 = time.perf_counter()
            res = fn()
            ms = (time.perf_counter() - t0) * 1e3
            info = fhestr.Plan.string_op(eng, *plan_args).info()
            strings[name] = {"ms_per_op_inputs_from_host": ms, "host_buffers": "page-locked (fhe_host_alloc)", "n_pbs": info["n_pbs"],
                             "levels": info["n_levels"], "correct": bool(dec(res) == want), "pbs_per_s": info["n_pbs"] / (ms * 1e-3)}
        t0 = time.perf_counter()
        res = fhestr.FheStringOps(eng).to_lower(es_pageable)
        strings["to_lower_1024"]["ms_per_op_pageable_host_buffers"] = (time.perf_counter() - t0) * 1e3
        strings["to_lower_1024"]["correct"] = bool(strings["to_lower_1024"]["correct"] and dec(res) == s.lower())
        out["string_ops"] = strings
        return out
    finally:
        eng.close()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.print_launch:
        print(json.dumps(launch_command(sys.argv[1:], args.gpus, 0)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as `python bench.py --gpus N`: become the launcher (before anything touches the GPU)
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    import fhestr

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # FHESTR_BENCH_REHEARSAL=1: every rank on GPU 0 with a gloo group and host-staged gathers -- walks the N > 1
    # code path on a one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing and say so
    rehearsal = world > 1 and os.environ.get("FHESTR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if rehearsal else "cuda"

    P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
    B = args.batch
    M = P.msg_mod * P.carry_mod

    # ---- synthetic workload: keys from a seed, uniform messages, 16 random LUTs ----
    ck = fhestr.ClientKey(P, SEED)
    rng = np.random.default_rng(SEED + rank)
    tables = rng.integers(0, M, size=(16, M))
    msgs = rng.integers(0, M, size=B)
    sel = rng.integers(0, 16, size=B)
    cts = ck.encrypt(msgs)

    eng = fhestr.Engine(P, local_rank, args.log2_points)
    # server keys are generated on the device (bit-identical to fhe_client_gen_server_keys, see
    # tests/test_gpu_parity.py); the standard-domain copies are only exported where the CPU baseline runs
    need_host_keys = rank == 0 and world == 1 and not args.no_cpu_baseline
    glwe_sk, small_sk = ck.secret_keys()
    exported = eng.generate_keys(glwe_sk, small_sk, SEED, export=need_host_keys)
    bsk, ksk = exported if need_host_keys else (None, None)
    lut_ids = np.array([eng.generate_lookup_table(lambda x, t=t: int(t[x]))[0] for t in tables], dtype=np.uint32)
    idx = lut_ids[sel]

    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_idx = torch.from_numpy(idx.view(np.int32)).cuda()
    # independent batches write to their own buffers (a ring of four): calls that write the same buffer are ordered by
    # the engine, which would serialise the overlapped mode
    d_outs = [torch.zeros_like(d_in) for _ in range(4)]
    d_out = d_outs[0]
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_outs[step_no[0] & 3].data_ptr(), B)
        step_no[0] += 1

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # throughput mode: consecutive steps are independent batches, so the keyswitch of step k+1 may run in the shadow
    # of the blind rotation of step k (fhe_engine_set_pipeline); the serial figure is reported next to it
    # Headline = overlapped batches (mode 2: consecutive steps alternate between two streams on the two-LWEs-per-CU
    # kernel); mode 1 (keyswitch in the shadow of the previous blind rotation, the round-2 headline) and the serial
    # figure are reported next to it.
    eng.set_pipeline(0 if args.serial else 2)
    # Untimed, before the W warm-up steps the caller asked for: enough steps that the clock ramp that follows an idle GPU is over after
    # the idle seconds of key generation and encryption (profiles/r03_after_idle.txt: launches run 12-16 % slower for tens of ms) --
    # together with W at least 40 steps; reported as `prewarm_steps`.
    prewarm = max(0, 40 - args.warmup)
    for _ in range(prewarm + args.warmup):
        step()
    fence()
    eng.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ks_ms, br_ms, calls = eng.kernel_times(reset=True)
    eng.set_pipeline(False)
    got = d_out.cpu().numpy().view(np.uint64).copy()      # the TIMED loop's outputs, before anything else writes
    got_others = [d.cpu().numpy().view(np.uint64).copy() for d in d_outs[1:]]
    shadow = None
    if not args.serial:        # mode 1: only the keyswitch overlaps (bit-identical to serial calls)
        eng.set_pipeline(1)
        n_sh = min(args.steps, 20)
        d_out_sh = torch.zeros_like(d_in)
        for _ in range(3):
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out_sh.data_ptr(), B)
        eng.synchronize()
        eng.kernel_times(reset=True)
        t1 = time.perf_counter()
        for _ in range(n_sh):
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out_sh.data_ptr(), B)
        eng.synchronize()
        sh_dt = (time.perf_counter() - t1) / n_sh
        sh_ks, sh_br, sh_calls = eng.kernel_times(reset=True)
        eng.set_pipeline(0)
        shadow = {"ms_per_step": sh_dt * 1e3, "pbs_per_s_per_gpu": B / sh_dt,
                  "verified_decrypt": bool(np.array_equal(ck.decrypt(d_out_sh.cpu().numpy().view(np.uint64)), tables[sel, msgs])),
                  "kernel_ms": {"keyswitch": sh_ks / max(sh_calls, 1), "blind_rotate": sh_br / max(sh_calls, 1)}}
    serial = None
    if not args.serial:        # the same steps one after the other (what a single dependent chain of calls gets)
        n_serial = min(args.steps, 10)
        d_out_serial = torch.zeros_like(d_in)

        def serial_step():
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out_serial.data_ptr(), B)

        serial_step()
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_serial):
            serial_step()
        eng.synchronize()
        serial_dt = (time.perf_counter() - t1) / n_serial
        s_ks, s_br, s_calls = eng.kernel_times(reset=True)
        serial_ok = bool(np.array_equal(ck.decrypt(d_out_serial.cpu().numpy().view(np.uint64)), tables[sel, msgs]))
        serial = {"ms_per_step": serial_dt * 1e3, "pbs_per_s_per_gpu": B / serial_dt, "verified_decrypt": serial_ok,
                  "kernel_ms": {"keyswitch": s_ks / max(s_calls, 1), "blind_rotate": s_br / max(s_calls, 1)}}
    per_rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        every = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(every, t)
        per_rank_elapsed = [float(v) for v in every.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # ---- correctness gate on the timed (pipelined unless --serial) output: decrypt == LUT(message) for every LWE ----
    dec = ck.decrypt(got)
    verified = bool(np.array_equal(dec, tables[sel, msgs])) and all(bool(np.array_equal(ck.decrypt(g), tables[sel, msgs])) for g in got_others)
    if world > 1:
        v = torch.tensor([1 if verified else 0], device=coll_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())

    # ---- SURVEY 8(d) config 2, second variant: one shared identity table for the whole batch ----
    identity = None
    if world == 1:
        ident_id, _ = eng.generate_lookup_table(lambda x: x)
        d_same = torch.full((B,), int(ident_id), dtype=torch.int32, device="cuda")
        d_id_out = torch.zeros_like(d_in)
        eng.set_pipeline(not args.serial)
        for _ in range(3):
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_same.data_ptr(), d_id_out.data_ptr(), B)
        eng.synchronize()
        n_id = min(args.steps, 20)
        t1 = time.perf_counter()
        for _ in range(n_id):
            eng.apply_lookup_table_dev(d_in.data_ptr(), d_same.data_ptr(), d_id_out.data_ptr(), B)
        eng.synchronize()
        id_dt = (time.perf_counter() - t1) / n_id
        eng.set_pipeline(False)
        eng.kernel_times(reset=True)
        id_ok = bool(np.array_equal(ck.decrypt(d_id_out.cpu().numpy().view(np.uint64)), msgs))
        identity = {"pbs_per_s": B / id_dt, "ms_per_step": id_dt * 1e3, "verified_decrypt": id_ok}

    # ---- batch-size sweep like the reference's throughput bench (benches/core_crypto/pbs_bench.rs:430-549) ----
    sweep = None
    if world == 1 and not args.no_sweep:
        sweep = {}
        for nb in (1, 16, 32, 64, 128, 256, 512, 1024, 4096):
            reps_in = (nb + B - 1) // B
            big_in = d_in.repeat(reps_in, 1)[:nb].contiguous()
            big_idx = d_idx.repeat(reps_in)[:nb].contiguous()
            big_out = torch.empty_like(big_in)
            torch.cuda.synchronize()
            # two untimed launches first: a launch that follows the nearly idle GPU of the small batches runs 12-16 % slower
            # for tens of ms (profiles/r03_after_idle.txt)
            for it in range(5):
                if it == 2:
                    eng.synchronize()
                    t1 = time.perf_counter()
                eng.apply_lookup_table_dev(big_in.data_ptr(), big_idx.data_ptr(), big_out.data_ptr(), nb)
            eng.synchronize()
            sweep[str(nb)] = round(nb * 3 / (time.perf_counter() - t1), 1)
        eng.kernel_times(reset=True)

    if rank == 0:
        total_pbs = B * world * args.steps
        value = total_pbs / elapsed
        br_avg_ms = br_ms / max(calls, 1)
        rec = {
            "metric": "PBS/sec (whole node)", "value": value, "unit": "PBS/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batched KS+PBS: 256 independent shortint LWE ciphertexts, "
                                   "PARAM_MESSAGE_2_CARRY_2_KS_PBS, 16 random LUTs, 1xMI355X per rank",
                       "batch_per_gpu": B, "params": P.name, "parallelism": f"replicated keys x{world}"},
            "kernel_ms": {"keyswitch": ks_ms / max(calls, 1), "blind_rotate": br_avg_ms, "launches": calls,
                          "kernel_revision": fhestr.kernel_revision()},
            "pipelined": (None if args.serial else
                          "overlapped batches (fhe_engine_set_pipeline(2)): consecutive steps alternate between two streams on the "
                          "two-LWEs-per-CU kernel, two batches share the GPU; kernel_ms are the durations of kernels that overlap, "
                          "ms_per_step is wall time / steps"),
            "pipelined_keyswitch_only": shadow,
            "serial": serial,
            # every rank's own rate over the same timed steps: value = the sum of the batches / the slowest rank's time
            "per_rank_pbs_per_s": [B * args.steps / e for e in per_rank_elapsed],
            "verified_decrypt": verified,
            **({"rehearsal": "all ranks share GPU 0 (gloo, host-staged gathers): code-path check only, not a measurement"}
               if rehearsal else {}),
            "string_ops": None,
            "shared_identity_lut": identity,
            "batch_sweep_pbs_per_s": sweep,
        }
        overlapped = not args.serial
        rec.update(rooflines(P, B, world, value, br_avg_ms, fhestr.kernel_revision(), args.log2_points,
                             kernel="blind_rotate_wide_kernel" if overlapped else "blind_rotate_kernel", concurrency=2 if overlapped else 1))
        if serial:   # the one-LWE-per-CU kernel of the serial comparison steps, nothing sharing the CUs with it
            alone = rooflines(P, B, world, value, serial["kernel_ms"]["blind_rotate"], fhestr.kernel_revision(), args.log2_points)
            rec["roofline"]["kernel_alone"] = {"avg_launch_ms": serial["kernel_ms"]["blind_rotate"], "frac": alone["roofline"]["frac"]}
            rec["roofline_hbm_model"]["kernel_alone"] = {"avg_launch_ms": serial["kernel_ms"]["blind_rotate"],
                                                         "frac": alone["roofline_hbm_model"]["frac"]}
            rec["roofline"]["note"] = ("overlapped batches: two launches of the two-LWEs-per-CU kernel share the GPU, avg_launch_ms is one launch's "
                                       "duration; kernel_alone = the one-LWE-per-CU kernel of the serial steps")


    # ---- FheString ms/op (BASELINE.json configs 3 and 4): level batches sharded over the ranks, one
    #      RCCL all-gather per level; single GPU = same code with world 1.  A watchdog makes sure the
    #      headline line is printed even if a collective of this secondary section were to hang. ----
    if not args.no_strings:
        import threading

        def bail():
            if rank == 0:
                rec["string_ops"] = {"error": "timeout: FheString section did not finish in 180 s"}
                print(json.dumps(rec), flush=True)
            os._exit(3)   # a wedged kernel / collective must reach the driver as a failure

        dog = threading.Timer(180.0, bail)
        dog.daemon = True
        dog.start()
        try:
            string_ops = bench_strings(fhestr, eng, ck, P, rank, world, local_rank, staged=rehearsal)
        except Exception as e:   # never let the secondary section take the headline number down
            string_ops = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            rec["string_ops"] = string_ops
            # first-class: FheString ms/op with every level's jobs sharded over the `world` ranks
            for key, field in (("eq_256_enc_enc", "fhestring_eq_256_ms"), ("contains_16_in_256_enc_enc", "fhestring_contains_16_in_256_ms"),
                               ("eq_256_enc_enc_keep_busy", "fhestring_eq_256_keep_busy_ms"),
                               ("eq_256_batch8", "fhestring_eq_256_batch8_ms_per_op"), ("eq_256_batch32", "fhestring_eq_256_batch32_ms_per_op"),
                               ("contains_16_in_256_batch8", "fhestring_contains_16_in_256_batch8_ms_per_op")):
                if isinstance(string_ops.get(key), dict):
                    rec[field] = string_ops[key]["ms_per_op"]

    if rank == 0 and world == 1 and not args.no_strings:
        try:
            rec["multi_bit_pbs"] = bench_multi_bit(fhestr, local_rank, B, args.steps)
        except Exception as e:   # secondary section
            rec["multi_bit_pbs"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and not args.no_strings:
        try:
            rec["n1024_k2"] = bench_n1024_k2(fhestr, local_rank)
        except Exception as e:   # secondary section
            rec["n1024_k2"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and not args.no_p44:
        try:
            rec["p44"] = bench_p44(fhestr, local_rank)
        except Exception as e:   # secondary section
            rec["p44"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                cb, cpu_out = cpu_baseline(P, bsk, ksk, cts, tables, sel)
                cb["decrypt_matches_gpu"] = bool(np.array_equal(ck.decrypt(cpu_out), dec))
                rec["cpu_baseline"] = cb
            except Exception as e:  # the baseline is informative; never let it kill the GPU number
                rec["cpu_baseline"] = {"value": None, "unit": "PBS/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        else:
            rec["cpu_baseline"] = None
            rec["cpu_baseline_reason"] = ("--no-cpu-baseline" if args.no_cpu_baseline else
                                          "timed on rank 0 at N = 1 only (the contract); see the N = 1 line of the same tree")
        print(json.dumps(rec), flush=True)
    if world > 1:
        # the headline line is out; never let the teardown of a wedged collective keep the job alive
        import threading
        end = threading.Timer(60.0, lambda: os._exit(4))   # teardown wedged: fail loudly (the line is already out)
        end.daemon = True
        end.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
        end.cancel()
    if not verified:
        raise SystemExit("decrypt check failed")


if __name__ == "__main__":
    main()
