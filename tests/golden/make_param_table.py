#!/usr/bin/env python3
"""Reads the parameter CONSTANTS of the reference's shortint layer (numbers only) into a JSON fixture:

    python3 tests/golden/make_param_table.py        (needs /root/reference; run in the authoring container)

Sources: tfhe/src/shortint/parameters/mod.rs (ClassicPBSParameters) and multi_bit.rs (MultiBitPBSParameters).
Output: tests/golden/reference_parameter_sets.json, one record per `pub const`, with the line it starts on."""
import json
import os
import re

ROOT = "/root/reference/tfhe/src/shortint/parameters"
FIELDS = {"lwe_dimension": int, "glwe_dimension": int, "polynomial_size": int, "lwe_modular_std_dev": float,
          "glwe_modular_std_dev": float, "pbs_base_log": int, "pbs_level": int, "ks_base_log": int, "ks_level": int,
          "message_modulus": int, "carry_modulus": int, "grouping_factor": int}
out = {}
for fname, kind in (("mod.rs", "ClassicPBSParameters"), ("multi_bit.rs", "MultiBitPBSParameters")):
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
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_parameter_sets.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(len(out), "parameter sets ->", path)
