import sys, numpy as np, os
sys.path.insert(0, "fhe-string-bounty_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fhestr, oracle as O
from conftest import keyset, gpu_engine
import torch
def chk(tag):
    try:
        print(tag, "hip device_count", torch._C._cuda_getDeviceCount(), flush=True)
    except Exception as e:
        print(tag, "ERR", e, flush=True)
chk("start")
ks = keyset(O.TOY_K1); eng = gpu_engine(ks); chk("after toy engine")
p22 = keyset(O.PARAM_MESSAGE_2_CARRY_2_KS_PBS)
for sel in (0, 2, 3, 4, 18, 19):
    e = gpu_engine(p22, sel)
    e.generate_lookup_table(lambda x: x)
    cts = p22.ck.encrypt_many(range(4))
    e.apply_lookup_table(cts)
    chk(f"after p22 variant {sel}")
g = np.load("tests/golden/toy_k1.npz")
from test_golden import _load
g, p = _load("tests/golden/toy_k1.npz")
from conftest import to_fhestr_params
e2 = fhestr.Engine(to_fhestr_params(p), 0); e2.load_keys(g["bsk"], g["ksk"]); e2.close(); chk("after close")
torch.zeros(1, device="cuda"); print("torch cuda ok")
