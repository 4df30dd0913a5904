"""Where a CMUX step of blind_rotate_xcd_kernel (csrc/pbs_xcd_kernels.hip.h) spends its cycles, per wave role, and which
compute unit hosts which clusters (diagnostic -DFHESTR_STAMPS build):

    make -C fhe-string-bounty_amd stamps
    FHESTR_LIB=build/stamps/libfhestr_stamps.so python3 scripts/stamp_xcd.py [B]      (default 16 = two clusters per XCD)

The stamps' scheduling fences forbid overlaps the real kernel has: shares, not absolute times."""
import collections
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

SEGS = ["phase 1: gather, decompose, digits -> partner (owner) / wait for them (partner)", "phase 1: twist, column transform, T stores",
        "key rows requested + hand-over 1", "phase 2: T row loads + forward row transform", "hand-over 2",
        "phase 3: inverse columns, accumulate, publish (owner)", "hand-over 3", "phase 2: multiply, reduce, LDS hand-off, barrier",
        "phase 2: inverse row transform + stores (even waves)", "(placement)"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
ck = fhestr.ClientKey(P, 7)
g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0)
eng.generate_keys(g, s, 7)
eng.set_cluster_mode(1)
lut, _ = eng.generate_lookup_table(lambda x: x)
rng = np.random.default_rng(0)
msgs = rng.integers(0, P.msg_mod * P.carry_mod, size=B)
cts = ck.encrypt(msgs)
out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
print("B =", B, "correct:", np.array_equal(ck.decrypt(out), msgs), "kernel ms", eng.last_kernel_ms())
NB = 512
n = NB * 8 * len(SEGS)
buf = np.zeros(n, dtype=np.uint64)
L = fhestr.lib()
L.fhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.fhe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
raw = buf.reshape(NB, 8, len(SEGS))[:, :4, :]
live = raw[:, 0, :9].sum(axis=1) > 0
raw = raw[live]
print(f"{raw.shape[0]} workgroups stamped")
st = raw[:, :, :9].astype(np.float64) / P.n
tot = st.sum(axis=2).mean()
print(f"sum of segments: {tot:.0f} cycles per step and wave = {tot / 2.4e3:.2f} us at 2.4 GHz")
for i, name in enumerate(SEGS[:9]):
    print(f"  {i}: {st[:, :, i].mean():8.1f} {100 * st[:, :, i].mean() / tot:5.1f} %   owner waves {st[:, 1::2, i].mean():8.1f}   partner waves {st[:, 0::2, i].mean():8.1f}   {name}")
# placement: clusters per compute unit
place = raw[:, 0, 9]
cu_of = collections.defaultdict(list)
for v in place:
    v = int(v)
    cluster, member, xcc, hw = v >> 48, (v >> 40) & 0xFF, (v >> 32) & 7, v & 0xFFFFFFFF
    cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)       # XCC, SE_ID, SH_ID, CU_ID
    cu_of[cu].append(cluster)
hist = collections.Counter()
for cu, cl in cu_of.items():
    hist[(len(cl), len(set(cl)))] += 1
print("compute units by (workgroups hosted, distinct clusters among them):", dict(hist), "over", len(cu_of), "CUs")
