import sys, numpy as np, os, ctypes
sys.path.insert(0, "fhe-string-bounty_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fhestr, oracle as O
import torch
hip = ctypes.CDLL("libamdhip64.so")
def cnt(tag):
    c = ctypes.c_int(-1); rc = hip.hipGetDeviceCount(ctypes.byref(c)); print(tag, "hipGetDeviceCount rc", rc, "count", c.value, flush=True)
mode = sys.argv[1]
P = fhestr.Params(16, 1, 256, 10, 2, 4, 4, 4, 4, 1e-12, 1e-15, "toy")
cnt("before")
e = fhestr.Engine(P, 0)
cnt("after create")
if mode == "close":
    e.close(); cnt("after close")
print("torch count", torch._C._cuda_getDeviceCount(), flush=True)
