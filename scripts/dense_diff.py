"""Diagnostic (GPU box): how far apart are the outputs of the three N = 1024 (k = 2) kernels on the same ciphertexts?
Every word differs (a decomposition digit moved by a rounding adds a key row), the phases differ by one noise sample."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fhe-string-bounty_amd"))
import fhestr
TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_parameter_sets.json")))
name = "PARAM_MESSAGE_2_CARRY_1_KS_PBS"; r = TABLE[name]
P = fhestr.Params(r["lwe_dimension"], r["glwe_dimension"], r["polynomial_size"], r["pbs_base_log"], r["pbs_level"], r["ks_base_log"], r["ks_level"],
                  r["message_modulus"], r["carry_modulus"], r["lwe_modular_std_dev"], r["glwe_modular_std_dev"], name)
M = P.msg_mod * P.carry_mod
ck = fhestr.ClientKey(P, 3); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 3)
rng = np.random.default_rng(1); table = rng.integers(0, M, size=M)
lut, _ = eng.generate_lookup_table(lambda x: int(table[x]))
B = 800
msgs = rng.integers(0, M, size=B); cts = ck.encrypt(msgs); idx = np.full(B, lut, dtype=np.uint32)
a = eng.apply_lookup_table(cts, idx)
a2 = eng.apply_lookup_table(cts, idx)
b = np.concatenate([eng.apply_lookup_table(cts[:400], idx[:400]), eng.apply_lookup_table(cts[400:], idx[400:])])
b2 = np.concatenate([eng.apply_lookup_table(cts[:400], idx[:400]), eng.apply_lookup_table(cts[400:], idx[400:])])
c = np.concatenate([eng.apply_lookup_table(cts[i:i + 200], idx[i:i + 200]) for i in range(0, B, 200)])
def dist(x, y):
    d = (x - y).view(np.int64).astype(np.float64); return np.abs(d)
print("dense run to run identical:", np.array_equal(a, a2), "; two-per-CU run to run identical:", np.array_equal(b, b2))
for label, x, y in (("dense vs two-per-CU", a, b), ("dense vs one-per-CU", a, c), ("two-per-CU vs one-per-CU", b, c)):
    d = dist(x, y)
    print(label, ": max 2^%.1f, median 2^%.1f, words beyond 2^40: %d of %d, rows affected %d" %
          (np.log2(d.max() + 1), np.log2(np.median(d) + 1), int((d > 2.0**40).sum()), d.size, int((d > 2.0**40).any(axis=1).sum())))
    rows = np.flatnonzero((d > 2.0**40).any(axis=1))
    print("   rows:", rows[:20], "cols of first:", np.flatnonzero(d[rows[0]] > 2.0**40)[:10] if len(rows) else None)
big_sel = np.flatnonzero(g == 1)
ph = lambda x: x[:, -1] - x[:, big_sel].sum(axis=1, dtype=np.uint64)
for label, x, y in (("dense vs two-per-CU", a, b), ("dense vs one-per-CU", a, c), ("two-per-CU vs one-per-CU", b, c)):
    d = dist(ph(x), ph(y)); print(label, "phase distance max 2^%.1f median 2^%.1f" % (np.log2(d.max() + 1), np.log2(np.median(d) + 1)))
for label, x in (("dense", a), ("two-per-CU", b), ("one-per-CU", c)):
    ph = x[:, -1] - x[:, big_sel].sum(axis=1, dtype=np.uint64)
    want = (table[msgs].astype(np.uint64) << np.uint64(64 - 1 - int(np.log2(M))))
    e = (ph - want).view(np.int64).astype(np.float64)
    print(label, "phase error std 2^%.1f max 2^%.1f" % (np.log2(e.std()), np.log2(np.abs(e).max())), "decrypt", np.array_equal(ck.decrypt(x), table[msgs]))
