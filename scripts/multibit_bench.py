"""PBS/s of the multi-bit PBS (PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS) next to the classic one
(PARAM_MESSAGE_2_CARRY_2_KS_PBS): device-generated keys, decrypt-checked.

    python3 scripts/multibit_bench.py [B ...]      (default 1 64 256 1024)

FHESTR_MULTIBIT_COMBINE_MAX=0 forces the fused kernel for every batch size (default: batches <= 64 prepare the
groups' GGSWs on the whole GPU first); MULTIBIT_ONLY=1 skips the classic parameter set."""
import os, sys, time
import numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

BATCHES = [int(a) for a in sys.argv[1:]] or [1, 64, 256, 1024]
SETS = (fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS,
        fhestr.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS)
if os.environ.get("OTHER_SHAPES"):      # the N = 512 (k = 3) and N = 8192 (two levels) sets: two-kernel path
    SETS = (fhestr.PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_2_KS_PBS, fhestr.PARAM_MULTI_BIT_MESSAGE_1_CARRY_1_GROUP_3_KS_PBS,
            fhestr.PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_2_KS_PBS, fhestr.PARAM_MULTI_BIT_MESSAGE_3_CARRY_3_GROUP_3_KS_PBS)
for P in SETS[1 if os.environ.get("MULTIBIT_ONLY") else 0:]:
    ck = fhestr.ClientKey(P, 0x5EED0002)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    eng.generate_keys(g, s, 0x5EED0002)
    M = P.msg_mod * P.carry_mod
    lut, _ = eng.generate_lookup_table(lambda x: (x * x + 1) % M)
    rng = np.random.default_rng(3)
    for B in BATCHES:
        msgs = rng.integers(0, M, size=B)
        cts = ck.encrypt(msgs)
        idx = np.full(B, lut, dtype=np.uint32)
        out = eng.apply_lookup_table(cts, idx)
        ok = np.array_equal(ck.decrypt(out), (msgs * msgs + 1) % M)
        eng.apply_lookup_table(cts, idx)
        ks, br = eng.last_kernel_ms()
        print(f"{P.name} B={B}: keyswitch {ks:.3f} ms, blind rotation {br:.3f} ms -> {B / ((ks + br) * 1e-3):.0f} PBS/s (kernels), correct {ok}", flush=True)
    eng.close()
