"""Is the slow launch after a nearly idle GPU (profiles/r03_after_idle.txt) a clock ramp?  Samples the shader clock
(rocm-smi / amdsmi through torch.cuda.clock_rate, whichever answers) from a host thread every few ms while the GPU runs
256-LWE launches, then 1-LWE launches, then one 2048-LWE launch.  (GPU box)"""
import os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr, torch


_smi = None


def read_clock():
    """(shader, memory, fabric) clocks in MHz through amdsmi when it answers, else the shader clock alone."""
    global _smi
    try:
        if _smi is None:
            import amdsmi
            amdsmi.amdsmi_init()
            _smi = (amdsmi, amdsmi.amdsmi_get_processor_handles()[0])
        a, h = _smi
        out = []
        for t in (a.AmdSmiClkType.GFX, a.AmdSmiClkType.MEM, a.AmdSmiClkType.DF, a.AmdSmiClkType.SOC):
            try:
                info = a.amdsmi_get_clock_info(h, t)
                out.append(float(info.get("clk", info.get("cur_clk", -1))))
            except Exception:
                out.append(-1.0)
        return tuple(out)
    except Exception:
        _smi = False
    try:
        return (float(torch.cuda.clock_rate()),)
    except Exception:
        pass
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        import json
        d = json.loads(out)
        card = next(iter(d.values()))
        for k, v in card.items():
            if "sclk" in k.lower():
                return float(str(v).strip("()Mhz ").replace("Mhz", ""))
    except Exception as e:
        return (-1.0,)
    return (-1.0,)


P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 5); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 5)
lut = eng.generate_lookup_table(lambda x: x)[0]
B = 2048
d_in = torch.from_numpy(ck.encrypt(np.arange(B) % 16).view(np.int64)).cuda()
d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
run = lambda n: eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), n)
print("clock source answers:", read_clock(), flush=True)
samples, stop, phase = [], False, ["start"]


def sampler():
    t0 = time.perf_counter()
    while not stop:
        samples.append((time.perf_counter() - t0, phase[0], read_clock()))
        time.sleep(0.002)


th = threading.Thread(target=sampler); th.start()
for name, sizes in (("busy: 40 x 256 LWEs", [256] * 40), ("nearly idle: 12 x 1 LWE", [1] * 12), ("one 2048-LWE launch", [2048]),
                    ("busy again: 20 x 256", [256] * 20)):
    phase[0] = name
    t0 = time.perf_counter()
    for n in sizes: run(n)
    eng.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
stop = True; th.join()
for name in ("busy: 40 x 256 LWEs", "nearly idle: 12 x 1 LWE", "one 2048-LWE launch", "busy again: 20 x 256"):
    v = [c if isinstance(c, tuple) else (c,) for _, p, c in samples if p == name]
    if v:
        cols = list(zip(*v))
        print(f"{name}: {len(v)} samples; per clock (gfx, mem, fabric, soc) min/mean/max MHz: " +
              "  ".join(f"{min(c):.0f}/{sum(c) / len(c):.0f}/{max(c):.0f}" for c in cols) + f"; first three {v[:3]}")
