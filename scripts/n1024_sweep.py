import json, os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
import fhestr, torch
TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_parameter_sets.json")))
name = "PARAM_MESSAGE_2_CARRY_1_KS_PBS"; r = TABLE[name]
P = fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"], r["ks_base_log"], r["ks_level"],
                  r["message_modulus"], r["carry_modulus"], r["lwe_modular_std_dev"], r["glwe_modular_std_dev"], name)
M = P.msg_mod * P.carry_mod
ck = fhestr.ClientKey(P, 3); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 3)
rng = np.random.default_rng(1); table = rng.integers(0, M, size=M)
lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
print("FHESTR_DENSE_PER_CU", os.environ.get("FHESTR_DENSE_PER_CU"), "kernel revision", fhestr.kernel_revision() if hasattr(fhestr, "kernel_revision") else "")
for B in (256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192):
    msgs = rng.integers(0, M, size=B)
    d_in = torch.from_numpy(ck.encrypt(msgs).view(np.int64)).cuda()
    d_idx = torch.full((B,), int(lut), dtype=torch.int32, device="cuda"); d_out = torch.zeros_like(d_in)
    for it in range(5):
        if it == 2: eng.synchronize(); t0 = time.perf_counter()
        eng.apply_lookup_table_dev(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), B)
    eng.synchronize(); dt = (time.perf_counter() - t0) / 3
    ok = bool(np.array_equal(ck.decrypt(d_out.cpu().numpy().view(np.uint64)), table[msgs]))
    print(f"  B = {B:5d}: {dt*1e3:7.3f} ms -> {B/dt:9.0f} PBS/s, correct {ok}", flush=True)
