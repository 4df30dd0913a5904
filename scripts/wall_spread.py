"""Do the workgroups of one 256-LWE blind-rotation launch take the same wall TIME?  (-DFHESTR_WALL build:
FHESTR_LIB=build/ab/libfhestr_wall.so python3 scripts/wall_spread.py [B])  Prints the spread per XCD on the 100 MHz clock."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 7); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 7)
lut, _ = eng.generate_lookup_table(lambda x: x)
cts = ck.encrypt(np.arange(B) % 16)
for _ in range(4):
    out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
print("correct:", np.array_equal(ck.decrypt(out), np.arange(B) % 16), "kernel ms", eng.last_kernel_ms())
buf = np.zeros(2 * B, dtype=np.uint64)
L = fhestr.lib(); L.fhe_debug_read_wall.argtypes = [C.c_void_p, C.c_size_t]
assert L.fhe_debug_read_wall(buf.ctypes.data_as(C.c_void_p), 2 * B) == 0
wall = buf[0::2].astype(np.float64) / 100.0      # microseconds
xcc = buf[1::2].astype(int)
print(f"all workgroups: min {wall.min():.1f} us, median {np.median(wall):.1f}, max {wall.max():.1f}  (max / median {wall.max() / np.median(wall):.3f})")
for x in range(8):
    w = wall[xcc == x]
    if w.size:
        print(f"  XCD {x}: {w.size:3d} workgroups, min {w.min():.1f}, mean {w.mean():.1f}, max {w.max():.1f} us")
