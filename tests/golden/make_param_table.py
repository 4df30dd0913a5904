#!/usr/bin/env python3
"""Reads the parameter CONSTANTS of the reference's shortint layer (numbers only) into JSON fixtures:

    python3 tests/golden/make_param_table.py        (needs /root/reference; run in the authoring container)

Sources: tfhe/src/shortint/parameters/mod.rs (ClassicPBSParameters) and multi_bit.rs (MultiBitPBSParameters)
-> tests/golden/reference_parameter_sets.json; parameters_compact_pk.rs (the "experimental" compact-public-key sets,
ClassicPBSParameters with other keyswitch decompositions, up to 22 levels) -> reference_parameter_sets_compact_pk.json.
One record per `pub const`, with the line it starts on."""
import json
import os
import re

ROOT = "/root/reference/tfhe/src/shortint/parameters"
FIELDS = {"lwe_dimension": int, "glwe_dimension": int, "polynomial_size": int, "lwe_modular_std_dev": float,
          "glwe_modular_std_dev": float, "pbs_base_log": int, "pbs_level": int, "ks_base_log": int, "ks_level": int,
          "message_modulus": int, "carry_modulus": int, "grouping_factor": int}


def read(sources):
    out = {}
    for fname, kind in sources:
        text = open(os.path.join(ROOT, fname)).read()
        for m in re.finditer(r"pub const (\w+): %s =\s*%s \{(.*?)\n\s*\};" % (kind, kind), text, re.S):
            name, body = m.group(1), m.group(2)
            rec = {"source": f"tfhe/src/shortint/parameters/{fname}:{text[:m.start()].count(chr(10)) + 1}"}
            for field, conv in FIELDS.items():
                f = re.search(r"\b%s: \w+\(([-+.eE0-9]+)\)" % field, body)
                if f:
                    rec[field] = conv(f.group(1))
            k = re.search(r"encryption_key_choice: EncryptionKeyChoice::(\w+)", body)
            rec["encryption_key_choice"] = k.group(1)
            if "lwe_dimension" in rec:       # aliases (`= OTHER_CONST;`) have no body and are skipped by the pattern
                out[name] = rec
    return out


for target, sources in (("reference_parameter_sets.json", (("mod.rs", "ClassicPBSParameters"), ("multi_bit.rs", "MultiBitPBSParameters"))),
                        ("reference_parameter_sets_compact_pk.json", (("parameters_compact_pk.rs", "ClassicPBSParameters"),))):
    out = read(sources)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), target)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(len(out), "parameter sets ->", path)
