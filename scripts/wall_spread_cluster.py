"""Do the clusters of one XCD stay in step?  (-DFHESTR_WALL build: FHESTR_LIB=build/wall/libfhestr.so python3 scripts/wall_spread_cluster.py)
PARAM_MESSAGE_4_CARRY_4, 256 LWEs on the cluster kernel: the 100 MHz clock when each cluster's leader starts CMUX steps
0, n/8, ... 7n/8 of its first LWE, printed per XCD (clusters are numbered XCD-major) relative to the XCD's earliest."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr
P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ck = fhestr.ClientKey(P, 7); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 7)
M = P.msg_mod * P.carry_mod
lut, _ = eng.generate_lookup_table(lambda x: x)
msgs = np.arange(B) % M
cts = ck.encrypt(msgs)
eng.set_cluster_mode(1)
for _ in range(2):
    out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
print("correct:", np.array_equal(ck.decrypt(out), msgs), "kernel ms", eng.last_kernel_ms(), "clusters", eng.cluster_info())
nc = eng.cluster_info()
buf = np.zeros(8 * nc, dtype=np.uint64)
L = fhestr.lib(); L.fhe_debug_read_wall.argtypes = [C.c_void_p, C.c_size_t]
assert L.fhe_debug_read_wall(buf.ctypes.data_as(C.c_void_p), 8 * nc) == 0
t = buf.reshape(nc, 8).astype(np.int64)
step_us = (t[:, 7] - t[:, 0]).mean() / 100.0 / (7 * (P.n // 8))
print(f"mean step time {step_us:.2f} us")
per = nc // 8
for x in range(8):
    rows = t[x * per:(x + 1) * per]
    rel = (rows - rows.min(axis=0)) / 100.0
    print(f"XCD-major group {x}: lag behind the group's first cluster (us) at steps k*n/8:")
    for c in range(per):
        print("   cluster", x * per + c, " ".join(f"{v:7.1f}" for v in rel[c]))
