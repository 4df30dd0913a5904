"""-DFHESTR_WALL build: wall time and start time of every workgroup of one two-LWEs-per-CU launch (512 / 4096 LWEs)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe-string-bounty_amd"))
import fhestr
P = fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS
ck = fhestr.ClientKey(P, 7); g, s = ck.secret_keys()
eng = fhestr.Engine(P, 0); eng.generate_keys(g, s, 7)
lut, _ = eng.generate_lookup_table(lambda x: x)
L = fhestr.lib(); L.fhe_debug_read_wall.argtypes = [C.c_void_p, C.c_size_t]
for B in (512, 1024, 4096):
    cts = ck.encrypt(np.arange(B) % 16)
    for _ in range(3):
        out = eng.apply_lookup_table(cts, np.full(B, lut, dtype=np.uint32))
    buf = np.zeros(2 * B, dtype=np.uint64)
    assert L.fhe_debug_read_wall(buf.ctypes.data_as(C.c_void_p), 2 * B) == 0
    wall = buf[0::2].astype(np.float64) / 100.0
    start = buf[1::2].astype(np.float64) / 100.0
    start -= start.min()
    end = start + wall
    print(f"B={B}: kernel {eng.last_kernel_ms()[1]:.3f} ms; workgroup wall time min {wall.min():.0f} / median {np.median(wall):.0f} / max {wall.max():.0f} us; "
          f"last start {start.max():.0f} us, last end {end.max():.0f} us")
    for r in range(0, B, 512):
        w = wall[r:r + 512]
        print(f"   workgroups {r}..{r + 511}: median wall {np.median(w):.0f} us, start {start[r:r + 512].min():.0f}..{start[r:r + 512].max():.0f}")
