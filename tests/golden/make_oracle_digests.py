#!/usr/bin/env python3
"""Writes tests/golden/oracle_digests.json: SHA-256 digests of the key material and of a few
gate outputs produced by the CPU oracle from fixed seeds.  Keys are 30 MB + 83 MB, so the
fixture holds seeds and digests; both the oracle and the product regenerate the keys."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    seed = 0x5EBA2
    ks = O.KeySet(O.params("P128"), seed)
    r = O.Rng(4242)
    cts = ks.encrypt(r, [0, 1, 1, 0, 1, 1])
    out = {
        "params": "P128", "key_seed": seed, "encrypt_seed": 4242,
        "lwe_key": sha(ks.lwe_key()), "tlwe_key": sha(ks.tlwe_key()), "bk": sha(ks.bk()), "ksk": sha(ks.ksk()),
        "encryptions_011011": sha(cts),
        "gates": {
            "AND_1_2": sha(ks.gate("AND", cts[1], cts[2])),
            "XOR_0_1": sha(ks.gate("XOR", cts[0], cts[1])),
            "OR_0_3": sha(ks.gate("OR", cts[0], cts[3])),
            "XNOR_4_5": sha(ks.gate("XNOR", cts[4], cts[5])),
            "MUX_1_0_2": sha(ks.mux(cts[1], cts[0], cts[2])),
        },
        "blind_rotate_acc_AND_1_2": sha(ks.blind_rotate(*(lambda b: (b[:-1], b[-1]))(ks.modswitch_ct(ks.prelude("AND", cts[1], cts[2]))))),
    }
    small = O.KeySet(O.custom_params(n=16, N=64, l=3, Bgbit=7), 99)
    out["small_n16_N64_seed99"] = {"bk": sha(small.bk()), "ksk": sha(small.ksk())}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
