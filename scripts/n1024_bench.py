#!/usr/bin/env python3
"""The N = 1024 (k = 2) family -- PARAM_MESSAGE_2_CARRY_1_KS_PBS, named by north_star next to N = 2048 -- by batch size and
throughput mode (GPU box): serial launches of 256 ... 4096 LWEs (one LWE per CU: the 384-thread kernel, three polynomials
on six waves; more: the 128-thread kernel whose threads carry all three polynomials, several workgroups per CU), and 256-LWE
calls overlapped on 2 / 3 / 4 streams (fhe_engine_set_pipeline(2), FHESTR_OVERLAP_STREAMS).  Decrypt-checked.

    python3 scripts/n1024_bench.py [name-part]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
import fhestr  # noqa: E402
import torch  # noqa: E402

TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_parameter_sets.json")))
name = next(k for k in sorted(TABLE) if TABLE[k]["polynomial_size"] == 1024 and (sys.argv[1] if len(sys.argv) > 1 else "MESSAGE_2_CARRY_1_KS_PBS") in k)
r = TABLE[name]
P = fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"], r["ks_base_log"], r["ks_level"],
                  r["message_modulus"], r["carry_modulus"], r["lwe_modular_std_dev"], r["glwe_modular_std_dev"], name)
M = P.msg_mod * P.carry_mod
ck = fhestr.ClientKey(P, 3)
g, s = ck.secret_keys()
streams = int(os.environ.get("FHESTR_OVERLAP_STREAMS", "2"))
eng = fhestr.Engine(P, 0)
eng.generate_keys(g, s, 3)
rng = np.random.default_rng(1)
table = rng.integers(0, M, size=M)
lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
print(name, f"n = {P.n}, k = {P.k}, N = {P.N}; overlap streams = {streams}")
for B in (256, 512, 1024, 2048, 4096):
    msgs = rng.integers(0, M, size=B)
    d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
    d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")
    d_out = torch.zeros_like(d_in)
    for it in range(6):
        if it == 2:
            eng.synchronize()
            t0 = time.perf_counter()
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 4
    ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
    print(f"  serial B = {B:5d}: {dt * 1e3:7.3f} ms per launch -> {B / dt:9.0f} PBS/s, correct {ok}", flush=True)
B = 256
msgs = rng.integers(0, M, size=B)
d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda")
outs = [torch.zeros_like(d_in) for _ in range(8)]
for mode in (0, 1, 2):
    eng.set_pipeline(mode)
    for it in range(48):
        if it == 8:
            eng.synchronize()
            t0 = time.perf_counter()
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), outs[it % 8].data_ptr(), B)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 40
    eng.set_pipeline(0)
    ok = all(bool(np.array_equal(ck.decrypt(o.cpu().numpy().view(np.uint64)), table[msgs])) for o in outs)
    print(f"  256-LWE calls back to back, pipeline mode {mode}: {dt * 1e3:7.3f} ms per call -> {B / dt:9.0f} PBS/s, correct {ok}", flush=True)
eng.close()
