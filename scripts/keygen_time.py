"""Time device-side server-key generation (fhe_engine_generate_keys) against the CPU client's."""
import sys, time
import numpy as np
sys.path.insert(0, "fhe-string-bounty_amd")
import fhestr

for P in (fhestr.PARAM_MESSAGE_2_CARRY_2_KS_PBS, fhestr.PARAM_MESSAGE_4_CARRY_4_KS_PBS):
    ck = fhestr.ClientKey(P, 0x5EED0002)
    g, s = ck.secret_keys()
    eng = fhestr.Engine(P, 0)
    t = time.time(); eng.generate_keys(g, s, 0x5EED0002); dt_gpu = time.time() - t
    lut, _ = eng.generate_lookup_table(lambda x: (x + 1) % (P.msg_mod * P.carry_mod))
    msgs = np.arange(8) % (P.msg_mod * P.carry_mod)
    out = eng.apply_lookup_table(ck.encrypt(msgs), np.full(8, lut, dtype=np.uint32))
    ok = np.array_equal(ck.decrypt(out), (msgs + 1) % (P.msg_mod * P.carry_mod))
    t = time.time(); ck.gen_server_keys(); dt_cpu = time.time() - t
    print(f"{P.name}: device keygen+install {dt_gpu*1e3:.0f} ms, CPU client keygen {dt_cpu:.1f} s (host threads), "
          f"keys {(P.bsk_len + P.ksk_len) * 8 / 2**20:.0f} MiB, PBS with device keys correct: {ok}", flush=True)
    eng.close()
