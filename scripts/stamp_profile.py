"""Where one CMUX step of blind_rotate_kernel spends its cycles (diagnostic -DFHESTR_STAMPS build):

    hipcc ... -DFHESTR_STAMPS -o build/ab/libfhestr_stamps.so ...    ("stamps")
    FHESTR_LIB=build/ab/libfhestr_stamps.so python3 scripts/stamp_profile.py

Prints, per segment, the average cycles per step and wave and its share.  The stamps' fences forbid the
overlaps the real kernel has, so only the shares mean something."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

SEGS = ["gather+decompose (incl. key-load issue)", "cvt+twist+fwd stage1 (2 passes, swap, 8 wr)", "fwd stage2 (8 rd, 2 passes, swap)",
        "fwd stage3 (twiddle, 8 wr)", "barrier 1", "fwd tail x2 (16 rd) + key wait + MAC", "inv head (pass, 8 wr)", "barrier 2",
        "inv tail (2 round trips, 2 swaps, 4 passes)", "untwist+round+acc+8 wr+barrier 3"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
if len(sys.argv) > 2 and sys.argv[2] == "p44":      # the large-N kernel: phases of one CMUX step
    P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
    SEGS = ["step head", "phase 1: decompose + column transforms -> tmp", "barrier 1", "phase 2: rows, multiply-accumulate, inverse rows -> tmp2",
            "barrier 2", "phase 3: inverse columns, accumulate", "barrier 3", "-", "-", "-"]
CLUSTER = len(sys.argv) > 2 and sys.argv[2] == "p44c"
if CLUSTER:                                          # the cluster kernel (several CUs per LWE): same phases, cluster hand-overs
    P = fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS
    SEGS = ["-", "phase 1: gather, decompose, column transforms -> T", "hand-over 1", "phase 2: rows, multiply-accumulate, inverse rows",
            "hand-over 2", "phase 3: inverse columns, accumulate, publish", "hand-over 3", "-", "-", "-"]
ck = fhestr.ClientKey(P, 7)
g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0)
eng.generate_keys(g, s, 7)
if CLUSTER:
    eng.set_cluster_mode(1)
lut, _ = eng.generate_lookup_table(lambda x: x)
rng = np.random.default_rng(0)
msgs = rng.integers(0, P.msg_mod * P.carry_mod, size=B)
cts = ck.encrypt(msgs)
for _ in range(1 if P.N > 4096 else 3):
    out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
print("correct:", np.array_equal(ck.decrypt(out), msgs), "kernel ms", eng.last_kernel_ms())
NB = 256 if CLUSTER else B          # the cluster kernel stamps per workgroup of its grid (first LWE of every cluster)
n = NB * 8 * len(SEGS)
buf = np.zeros(n, dtype=np.uint64)
L = fhestr.lib()
L.fhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.fhe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
st = buf.reshape(NB, 8, len(SEGS)).astype(np.float64) / P.n
if CLUSTER:
    st = st[st.sum(axis=(1, 2)) > 0]
    print(f"{st.shape[0]} workgroups stamped")     # cycles per step (s_memtime: shader clock... 100 MHz ticks?)
tot = st.sum(axis=2).mean()
print(f"sum of segments: {tot:.0f} ticks per step and wave")
per_wg = st.sum(axis=2).mean(axis=1)          # ticks per step, per workgroup
q = np.percentile(per_wg, [0, 5, 50, 95, 100])
print(f"per workgroup (ticks per step): min {q[0]:.0f}, 5 % {q[1]:.0f}, median {q[2]:.0f}, 95 % {q[3]:.0f}, max {q[4]:.0f}  (max / median {q[4] / q[2]:.3f})")
for i, name in enumerate(SEGS):
    print(f"  {i}: {st[:, :, i].mean():8.1f}  {100 * st[:, :, i].mean() / tot:5.1f} %   (waves 0-3 {st[:, :4, i].mean():7.1f}, waves 4-7 {st[:, 4:, i].mean():7.1f})  {name}")
