"""Print the fields of a bench.py JSON line that the round's documents quote.   python3 scripts/show_bench.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "PBS/s; ms/step", round(d["ms_per_step"], 3), "; serial", round(d["serial"]["pbs_per_s_per_gpu"]),
      "; ks-shadow", round(d["pipelined_keyswitch_only"]["pbs_per_s_per_gpu"]))
print("sweep", d.get("batch_sweep_pbs_per_s"))
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "frac", "avg_launch_ms", "kernel")})
for k, v in (d.get("string_ops") or {}).items():
    if isinstance(v, dict):
        host = v.get('ms_per_op_inputs_from_host')
        if "instances_per_rank" in v:      # many instances per pass (fhe_plan_run_batch)
            print(f"  {k}: {v['ms_per_op']:.2f} ms per op ({v['instances_per_rank']} per pass, {v['ms_per_pass']:.1f} ms per pass), "
                  f"{v['pbs_per_s']:.0f} PBS/s, correct {v['correct']}")
            continue
        print(f"  {k}: {v['ms_per_op']:.2f} ms resident, " + (f"{host:.2f} ms from host, " if host else "") + f"{v['n_pbs']} PBS, correct {v['correct']}")
p = d.get("p44") or {}
if p:
    print("p44", round(p["pbs_per_s"], 1), "PBS/s;", p["kernel_ms"], p.get("small_batch_kernel_ms"))
    print("p44 roofline", p.get("roofline", {}).get("frac"))
    for k, v in p.get("string_ops", {}).items():
        print("  p44", k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
mb = d.get("multi_bit_pbs") or {}
for g, v in mb.items():
    print(" ", g, round(v["pbs_per_s"]), "PBS/s; 1 LWE", round(v["batch_1"]["ms_per_step"], 3), "ms; eq", round(v["fhestring_eq_256_ms"], 2), "ms")
for k, v in (d.get("n1024_k2") or {}).items():
    if isinstance(v, dict):
        print("  n1024_k2", k, round(v["pbs_per_s"]), "PBS/s;", round(v["ms_per_step"], 3), "ms per step;", round(v["ms_per_256_lwes"], 3), "ms per 256 LWEs; correct", v["verified_decrypt"])
c = d.get("cpu_baseline") or {}
print("cpu", c.get("value"), c.get("cores"), c.get("pbs_per_s_by_threads"))
