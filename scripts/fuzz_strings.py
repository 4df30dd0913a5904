"""Randomised differential test of every FheString operation against Python bytes semantics on the GPU
(toy parameters: a PBS costs ~0.1 ms).  Usage: python scripts/fuzz_strings.py [cases] [seed]
FUZZ_PARAMS=n32768 runs it on 4-bit blocks (msg_mod = carry_mod = 16 on N = 32768, toy n): the whole-character and
pattern-class circuits of PARAM_MESSAGE_4_CARRY_4."""
import os
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "fhe-string-bounty_amd")
import oracle as O
import fhestr

N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 150
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = O.TOY_N32768 if os.environ.get("FUZZ_PARAMS") == "n32768" else O.TOY_K1
ck = O.ClientKey(p, 0xF022)
sk = O.ServerKey(ck, fourier=False)
P = fhestr.Params(p.n, p.k, p.N, p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level, p.msg_mod, p.carry_mod,
                  p.lwe_std, p.glwe_std, p.name)
eng = fhestr.Engine(P, 0)
eng.load_keys(sk.bsk, sk.ksk)
ops = fhestr.FheStringOps(eng)
rng = np.random.default_rng(SEED)
enc = lambda s, cap: ck.encrypt_many(fhestr.string_to_blocks(P, s, cap))
dec = lambda ct: ck.decrypt_many(np.asarray(ct).reshape(-1, p.big_size))
dec_str = lambda ct: fhestr.blocks_to_string(P, dec(ct))
ALPHA = b"abAB \t\nxyz.Zq"
WS = b" \t\n\x0b\x0c\r"


def rand_str(max_len):
    n = int(rng.integers(0, max_len + 1))
    return bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), size=n))


def digits_to_int(d):
    return sum(int(v) * (p.msg_mod ** i) for i, v in enumerate(d))


fails = 0
def check(name, got, want, ctx):
    global fails
    if got != want:
        fails += 1
        print(f"MISMATCH {name}: got {got!r} want {want!r} ctx {ctx!r}", flush=True)


for case in range(N_CASES):
    # 20 / 24: more than msg*carry = 16 candidate offsets, so reductions and prefix scans meet runs of exactly 16 bits
    cap = int(rng.choice([2, 3, 4, 6, 8, 20] if p.msg_mod == 4 else [3, 4, 6, 8, 12, 24]))
    a = rand_str(cap)
    # patterns: often substrings of a so that positive cases are frequent
    if len(a) and rng.random() < 0.6:
        i = int(rng.integers(0, len(a))); j = int(rng.integers(i, len(a) + 1))
        b = a[i:j]
    else:
        b = rand_str(min(cap, 4))
    b_cap = max(1, int(rng.choice([len(b), min(cap, max(len(b), 1) + 1)])))
    if b_cap < len(b): b_cap = len(b)
    ea, eb = enc(a, cap), enc(b, b_cap)
    op = str(rng.choice(["eq", "ne", "lt", "le", "gt", "ge", "eqic", "starts", "ends", "contains", "find", "rfind",
                         "upper", "lower", "trim_start", "trim_end", "strip", "replace", "len", "is_empty",
                         "strip_prefix", "strip_suffix", "concat", "repeat", "replace_general"]))
    clear = bool(rng.random() < 0.5)
    rhs = b if clear else eb
    ctx = (op, a, b, cap, b_cap, clear)
    if op in ("eq", "ne", "lt", "le", "gt", "ge"):
        if not clear and b_cap != cap:      # compare equal capacities or clear
            eb2 = enc(b[:cap], cap); b2 = b[:cap]; rhs = eb2
        else:
            b2 = b
        want = {"eq": a == b2, "ne": a != b2, "lt": a < b2, "le": a <= b2, "gt": a > b2, "ge": a >= b2}[op]
        check(op, int(dec(getattr(ops, op)(ea, rhs))[0]), int(want), ctx)
    elif op == "eqic":
        check(op, int(dec(ops.eq_ignore_case(ea, rhs))[0]), int(a.lower() == b.lower()), ctx)
    elif op == "starts":
        check(op, int(dec(ops.starts_with(ea, rhs))[0]), int(a.startswith(b)), ctx)
    elif op == "ends":
        check(op, int(dec(ops.ends_with(ea, rhs))[0]), int(a.endswith(b)), ctx)
    elif op == "contains":
        check(op, int(dec(ops.contains(ea, rhs))[0]), int(b in a), ctx)
    elif op in ("find", "rfind"):
        r = dec(getattr(ops, op)(ea, rhs))
        idx = a.find(b) if op == "find" else a.rfind(b)
        check(op + ".found", int(r[0]), int(idx >= 0), ctx)
        if idx >= 0:
            check(op + ".index", digits_to_int(r[1:]), idx, ctx)
    elif op == "upper":
        check(op, dec_str(ops.to_upper(ea)), a.upper(), ctx)
    elif op == "lower":
        check(op, dec_str(ops.to_lower(ea)), a.lower(), ctx)
    elif op == "trim_start":
        check(op, dec_str(ops.trim_start(ea)), a.lstrip(WS), ctx)
    elif op == "trim_end":
        check(op, dec_str(ops.trim_end(ea)), a.rstrip(WS), ctx)
    elif op == "strip":
        check(op, dec_str(ops.strip(ea)), a.strip(WS), ctx)
    elif op == "replace":
        if len(b) == 0:
            continue
        to = bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), size=len(b)))
        if clear:
            got = dec_str(ops.replace(ea, b, to))
        else:
            got = dec_str(ops.replace(ea, enc(b, len(b)), enc(to, len(b))))
        check(op, got, a.replace(b, to), ctx + (to,))
    elif op == "replace_general":
        # any lengths; clear operands (incl. the empty pattern) or encrypted zero padded ones
        to = bytes(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), size=int(rng.integers(0, 4))))
        if not clear and len(b) == 0:
            continue        # an encrypted empty pattern replaces nothing by definition (fhestr.h)
        want = a.replace(b, to)
        out_cap = max(1, len(want) + int(rng.integers(0, 2)))
        if clear:
            got = dec_str(ops.replace(ea, b, to, out_cap=out_cap))
        else:
            got = dec_str(ops.replace(ea, eb, enc(to, max(1, len(to) + int(rng.integers(0, 2)))), out_cap=out_cap))
        check(op, got, want, ctx + (to, out_cap))
    elif op == "len":
        check(op, digits_to_int(dec(ops.len(ea))), len(a), ctx)
    elif op == "is_empty":
        check(op, int(dec(ops.is_empty(ea))[0]), int(len(a) == 0), ctx)
    elif op == "concat":
        check(op, dec_str(ops.concat(ea, rhs)), a + b, ctx)
    elif op == "repeat":
        n = int(rng.integers(1, 4))
        check(op, dec_str(ops.repeat(ea, n)), a * n, ctx + (n,))
    elif op in ("strip_prefix", "strip_suffix"):
        bit, out = getattr(ops, op)(ea, rhs)          # clear or encrypted (padded) pattern
        had = a.startswith(b) if op == "strip_prefix" else a.endswith(b)
        want = (a[len(b):] if op == "strip_prefix" else a[:len(a) - len(b)]) if had else a
        check(op + ".bit", int(dec(bit)[0]), int(had), ctx)
        check(op + ".str", dec_str(out), want, ctx)
print(f"fuzz: {N_CASES} cases, {fails} mismatches")
sys.exit(1 if fails else 0)
