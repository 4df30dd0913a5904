#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ with the CPU oracle.

The reference (Rust) cannot run in this image and holds no ciphertext fixtures of its own
(SURVEY.md F3, F7), so these vectors are produced by the oracle -- which is itself pinned to the
reference's doctest known-answer vectors (tests/test_oracle_kat.py) -- from fixed seeds.  They pin
today's behaviour of both the oracle and the HIP path: integer-only stages bit-for-bit, PBS at
decrypt / phase level.   Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def make(params, seed, path):
    ck = O.ClientKey(params, seed)
    sk = O.ServerKey(ck)
    M = params.msg_mod * params.carry_mod
    rng = np.random.default_rng(seed)
    # keyswitch: random big LWEs (incl. edge rows) -> small LWEs, bit-exact
    ks_in = rng.integers(0, 2**64, size=(9, params.big_size), dtype=np.uint64)
    ks_in[0] = 0
    ks_in[1, :-1] = 0
    ks_in[2] = 2**64 - 1
    ks_out = np.stack([sk.keyswitch(c) for c in ks_in])
    # LUTs
    table = np.array([(5 * x + 3) % M for x in range(M)], dtype=np.uint64)
    lut, degree = sk.generate_lookup_table(lambda x: int(table[x]))
    # zero-mask PBS (integer-only path), bit-exact
    zm_in = np.zeros((6, params.small_size), dtype=np.uint64)
    zm_in[:, -1] = np.array([0, 2**64 - 1, 2**63, 1 << 40, 0x0123456789ABCDEF, 0xFEDCBA9876543210], dtype=np.uint64)
    zm_out = np.stack([sk.pbs(s, lut) for s in zm_in])
    # full KS+PBS: inputs, expected clear results, phases of the oracle's f64 and exact-integer outputs
    msgs = np.arange(M, dtype=np.uint64)
    pbs_in = ck.encrypt_many(msgs, O.Rng(seed, 77))
    out_fft = sk.apply_lookup_table_batch(pbs_in, lut)
    out_exact = sk.apply_lookup_table_batch(pbs_in, lut, exact=True)
    phases_fft = np.array([ck.decrypt_plaintext(c) for c in out_fft], dtype=np.uint64)
    phases_exact = np.array([ck.decrypt_plaintext(c) for c in out_exact], dtype=np.uint64)
    np.savez_compressed(
        path, seed=np.uint64(seed), params=np.array([params.n, params.k, params.N, params.pbs_base_log,
                                                     params.pbs_level, params.ks_base_log, params.ks_level,
                                                     params.msg_mod, params.carry_mod], dtype=np.uint32),
        stds=np.array([params.lwe_std, params.glwe_std]), glwe_sk=ck.glwe_sk, small_sk=ck.small_sk,
        bsk=sk.bsk, ksk=sk.ksk, ks_in=ks_in, ks_out=ks_out, table=table, lut=lut, degree=np.uint64(degree),
        zm_in=zm_in, zm_out=zm_out, msgs=msgs, pbs_in=pbs_in, expected=table[msgs],
        phases_fft=phases_fft, phases_exact=phases_exact)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    make(O.TOY_K1, 0x601D0001, os.path.join(HERE, "toy_k1.npz"))
    make(O.TOY_K2, 0x601D0002, os.path.join(HERE, "toy_k2.npz"))
