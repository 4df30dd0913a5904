"""tfhe-rs wire format: LweCiphertext / LweKeyswitchKey / standard LweBootstrapKey /
shortint::Ciphertext as serde + bincode 1.x write them (csrc/wire_format.cpp cites the struct definitions).

The reference ships no serialized fixture and cannot run here (no Rust toolchain): PARITY UNPINNED at the byte
level.  What is pinned: the layout against bincode's rules on a hand-assembled example (fixed-width little
endian integers, usize = u64, u128 = 16 bytes, Vec / String = u64 length + elements, struct fields in
declaration order, unit enum variant = u32 index), round trips, and the reader's refusals (the reference's
own: wrong scalar width ciphertext_modulus.rs:74-80, version / type-name mismatch and size limit
safe_deserialization.rs:62-99, dimensions against the parameter set conformance.rs)."""
import struct

import numpy as np
import pytest

import oracle as O
from conftest import keyset, to_fhestr_params


def _w():
    from fhestr import wire
    return wire


def test_lwe_ciphertext_layout_by_hand():
    w = _w()
    ct = np.array([1, 0x0123456789ABCDEF, 2**64 - 1], dtype=np.uint64)
    data = w.write_lwe_ciphertext(ct)
    want = struct.pack("<Q", 3) + struct.pack("<QQQ", 1, 0x0123456789ABCDEF, 2**64 - 1)   # data: Vec<u64>
    want += (0).to_bytes(16, "little")                                                      # modulus: u128, 0 = native
    want += struct.pack("<Q", 64)                                                           # scalar_bits: usize
    assert data == want
    back, used = w.read_lwe_ciphertext(data)
    assert np.array_equal(back, ct) and used == len(data)


def test_shortint_ciphertext_layout_and_safe_framing():
    w = _w()
    ct = np.arange(5, dtype=np.uint64)
    meta = w.ShortintMeta(degree=3, noise_level=1, message_modulus=4, carry_modulus=4, pbs_order=0)
    plain = w.write_shortint_ciphertext(ct, meta)
    body = struct.pack("<Q", 5) + ct.tobytes() + (0).to_bytes(16, "little") + struct.pack("<Q", 64)
    body += struct.pack("<QQQQ", 3, 1, 4, 4) + struct.pack("<I", 0)      # degree, noise_level, moduli, PBSOrder variant
    assert plain == body
    safe = w.write_shortint_ciphertext(ct, meta, safe=True)
    header = struct.pack("<Q", 3) + b"0.1" + struct.pack("<Q", 20) + b"shortint::Ciphertext"
    assert safe == header + body
    for data, is_safe in ((plain, False), (safe, True)):
        back, m, used = w.read_shortint_ciphertext(data, safe=is_safe)
        assert np.array_equal(back, ct) and m == meta and used == len(data)
    small = w.write_shortint_ciphertext(ct, w.ShortintMeta(2, 1, 4, 4, 1), safe=True)
    assert w.read_shortint_ciphertext(small, safe=True)[1].pbs_order == 1


def test_reader_refusals():
    import fhestr
    w = _w()
    ct = np.arange(4, dtype=np.uint64)
    data = w.write_lwe_ciphertext(ct)
    for cut in (0, 7, 8, 20, len(data) - 1):
        with pytest.raises(fhestr.FheError, match="truncated"):
            w.read_lwe_ciphertext(data[:cut])
    wrong_bits = data[:-8] + struct.pack("<Q", 32)
    with pytest.raises(fhestr.FheError, match="64 bits"):
        w.read_lwe_ciphertext(wrong_bits)
    custom_modulus = data[:-24] + (2**63).to_bytes(16, "little") + data[-8:]
    with pytest.raises(fhestr.FheError, match="native modulus"):
        w.read_lwe_ciphertext(custom_modulus)
    with pytest.raises(fhestr.FheError, match="longer than the destination"):
        w.read_lwe_ciphertext(data, max_words=3)
    huge = struct.pack("<Q", 2**61) + data[8:]            # a length field that would overflow a naive size computation
    with pytest.raises(fhestr.FheError, match="longer than the destination|truncated"):
        w.read_lwe_ciphertext(huge)
    meta = w.ShortintMeta(3)
    safe = w.write_shortint_ciphertext(ct, meta, safe=True)
    with pytest.raises(fhestr.FheError, match="version"):
        w.read_shortint_ciphertext(safe.replace(b"0.1", b"0.2", 1), safe=True)
    with pytest.raises(fhestr.FheError, match="expected type"):
        w.read_shortint_ciphertext(safe.replace(b"shortint::Ciphertext", b"shortint::Ciphertexu"), safe=True)
    with pytest.raises(fhestr.FheError, match="size limit"):
        w.read_shortint_ciphertext(safe, safe=True, size_limit=16)
    bad_variant = safe[:-4] + struct.pack("<I", 7)
    with pytest.raises(fhestr.FheError, match="PBSOrder"):
        w.read_shortint_ciphertext(bad_variant, safe=True)


def test_server_keys_round_trip_and_conformance(toy_k1):
    """KSK / standard BSK of a key set: byte form -> back, bit for bit; a key for another parameter set is refused."""
    import dataclasses
    import fhestr
    w = _w()
    P = to_fhestr_params(toy_k1.params)
    ksk_bytes = w.write_keyswitch_key(P, toy_k1.sk.ksk)
    bsk_bytes = w.write_bootstrap_key(P, toy_k1.sk.bsk)
    assert len(ksk_bytes) == 8 + P.ksk_len * 8 + 3 * 8 + 24
    assert len(bsk_bytes) == 8 + P.bsk_len * 8 + 4 * 8 + 24
    # trailing fields: decomp_base_log, decomp_level_count, output_lwe_size, then the modulus
    assert struct.unpack("<QQQ", ksk_bytes[8 + P.ksk_len * 8: 8 + P.ksk_len * 8 + 24]) == (P.ks_base_log, P.ks_level, P.n + 1)
    # ggsw list: glwe_size, polynomial_size, decomp_base_log, decomp_level_count
    assert struct.unpack("<QQQQ", bsk_bytes[8 + P.bsk_len * 8: 8 + P.bsk_len * 8 + 32]) == (P.k + 1, P.N, P.pbs_base_log, P.pbs_level)
    assert np.array_equal(w.read_keyswitch_key(P, ksk_bytes), toy_k1.sk.ksk.ravel())
    assert np.array_equal(w.read_bootstrap_key(P, bsk_bytes), toy_k1.sk.bsk.ravel())
    other = dataclasses.replace(P, ks_base_log=P.ks_base_log + 1)
    with pytest.raises(fhestr.FheError, match="does not match the parameter set"):
        w.read_keyswitch_key(other, ksk_bytes)
    other = dataclasses.replace(P, pbs_base_log=P.pbs_base_log - 1)
    with pytest.raises(fhestr.FheError, match="does not match the parameter set"):
        w.read_bootstrap_key(other, bsk_bytes)
    smaller = dataclasses.replace(P, n=P.n - 1)
    with pytest.raises(fhestr.FheError):
        w.read_keyswitch_key(smaller, ksk_bytes)


@pytest.mark.gpu
def test_wire_format_end_to_end_on_the_gpu(toy_k1):
    """A tfhe-rs-shaped exchange: server keys and a ciphertext arrive as bytes, the engine evaluates, the
    result leaves as a shortint::Ciphertext with the reference's metadata (degree = LUT degree, noise NOMINAL)."""
    import fhestr
    w = _w()
    P = to_fhestr_params(toy_k1.params)
    eng = fhestr.Engine(P, 0)
    try:
        eng.load_keys(w.read_bootstrap_key(P, w.write_bootstrap_key(P, toy_k1.sk.bsk)),
                      w.read_keyswitch_key(P, w.write_keyswitch_key(P, toy_k1.sk.ksk)))
        M = P.msg_mod * P.carry_mod
        lut_id, degree = eng.generate_lookup_table(lambda x: (3 * x + 1) % M)
        wire_in = w.write_shortint_ciphertext(toy_k1.ck.encrypt(5), w.ShortintMeta(3, 1, P.msg_mod, P.carry_mod, 0), safe=True)
        ct, meta, _ = w.read_shortint_ciphertext(wire_in, safe=True, size_limit=1 << 20)
        assert meta.pbs_order == 0 and meta.message_modulus == P.msg_mod
        out = eng.apply_lookup_table(ct[None, :], np.array([lut_id], dtype=np.uint32))[0]
        wire_out = w.write_shortint_ciphertext(out, w.ShortintMeta(degree, 1, P.msg_mod, P.carry_mod, 0), safe=True)
        back, m2, _ = w.read_shortint_ciphertext(wire_out, safe=True)
        assert toy_k1.ck.decrypt_message_and_carry(back) == (3 * 5 + 1) % M and m2.degree == degree
    finally:
        eng.close()
