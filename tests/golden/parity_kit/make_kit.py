#!/usr/bin/env python3
"""Writes the parity kit: raw-word fixtures of a tiny parameter tuple with which someone who HAS upstream
tfhe/tfhe can pin this repo's oracle (and therefore its GPU path) at ciphertext level -- the one thing nothing in
this environment can do (the reference links an absent, un-versioned libtfhe: /root/reference/CMakeLists.txt:9-15;
it includes tfhe_io.h, /root/reference/include/Math.h:5, but holds no vector).

    python tests/golden/parity_kit/make_kit.py            # (re)writes the files below from the CPU oracle

Parameter tuple: TFHE's default 128-bit set with the LWE dimension cut to n = 4 (N = 1024, k = 1, l = 3,
Bg = 2^7, key switch t = 8 x 2 bit, same noise parameters): the arithmetic of every step is the full-size one,
the keys are 0.7 MB instead of 110 MB.  Files (int32 little endian, raw words, no header):

    lwe_key.i32    n words 0/1                                   upstream: key->lwe_key->key[i]
    tlwe_key.i32   k N words 0/1                                 key->tgsw_key->tlwe_key.key[p].coefs[j]
    bk.i32         [i < n][row < (k+1) l][poly <= k][coef < N]   cloud->bk->bk[i].all_sample[row].a[poly].coefsT[coef]
                                                                 (poly k is the body, upstream's ->b)
    ksk.i32        [i < k N][j < t][v = 1 .. base-1][n + 1]      cloud->bk->ks->ks[i][j][v] (.a[0..n-1], .b); v = 0 is all zero
    inputs.i32     [8][n + 1]   encryptions of 0,1,1,0,1,1,0,0   sample->a[0..n-1], sample->b
    expected.i32   [9][n + 1]   outputs, in the order of kit.json "cases"
    extracted.i32  [k N + 1]    case 0 before its key switch (tfhe_bootstrap_woKS output): localises a mismatch

kit.json holds the tuple, the seeds, the case list and SHA-256 of every file.  check_against_upstream.cpp is the
program to run on the machine that has upstream.  tests/test_host_cpu.py checks the committed files against the oracle,
tests/test_gpu_kernels.py::test_parity_kit_words_are_what_the_gpu_computes against the GPU."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

KEY_SEED, ENC_SEED = 0x9A17, 0x1D
N_LWE = 4
BITS = [0, 1, 1, 0, 1, 1, 0, 0]
# (gate, operand indices into inputs)
CASES = [("AND", 1, 2), ("AND", 0, 1), ("XOR", 1, 2), ("XOR", 0, 3), ("OR", 0, 3), ("XNOR", 1, 3), ("NAND", 4, 5),
         ("MUX", 1, 0, 2), ("MUX", 3, 0, 2)]


def build():
    O.build()
    p = O.custom_params(n=N_LWE, N=1024, l=3, Bgbit=7)
    ks = O.KeySet(p, KEY_SEED)
    base = 1 << p.ks_basebit
    ksk = ks.ksk().reshape(p.k * p.N, p.ks_t, base, p.n + 1)
    assert not ksk[:, :, 0, :].any()                                   # the digit-0 rows are never read and are zero
    cts = ks.encrypt(O.Rng(ENC_SEED), BITS)
    outs, plain = [], []
    for c in CASES:
        if c[0] == "MUX":
            outs.append(ks.mux(cts[c[1]], cts[c[2]], cts[c[3]], use_ntt=False))      # schoolbook: exact by construction
            plain.append(BITS[c[2]] if BITS[c[1]] else BITS[c[3]])
        else:
            outs.append(ks.gate(c[0], cts[c[1]], cts[c[2]], use_ntt=False))
            a, b = BITS[c[1]], BITS[c[2]]
            plain.append({"AND": a & b, "XOR": a ^ b, "OR": a | b, "XNOR": 1 - (a ^ b), "NAND": 1 - (a & b)}[c[0]])
        assert (outs[-1] == (ks.mux(cts[c[1]], cts[c[2]], cts[c[3]], use_ntt=2) if c[0] == "MUX"
                             else ks.gate(c[0], cts[c[1]], cts[c[2]], use_ntt=2))).all()
    assert list(ks.decrypt(np.stack(outs))) == plain
    extracted = ks.bootstrap_woks(ks.prelude(CASES[0][0], cts[CASES[0][1]], cts[CASES[0][2]]), use_ntt=False)
    files = {     # copies: the key arrays are views of memory the KeySet frees when it goes out of scope
        "lwe_key.i32": ks.lwe_key().copy(), "tlwe_key.i32": ks.tlwe_key().copy(), "bk.i32": ks.bk().copy(),
        "ksk.i32": np.ascontiguousarray(ksk[:, :, 1:, :]).copy(), "inputs.i32": cts, "expected.i32": np.stack(outs),
        "extracted.i32": extracted,
    }
    meta = {
        "what": "parity kit: raw-word fixtures for checking this repo's exact-integer TFHE gate bootstrapping against "
                "upstream tfhe/tfhe's exact (non-FFT) bootstrap; see make_kit.py and check_against_upstream.cpp",
        "params": {"n": p.n, "N": p.N, "k": p.k, "l": p.l, "Bgbit": p.Bgbit, "ks_t": p.ks_t, "ks_basebit": p.ks_basebit,
                   "ks_stdev": p.ks_stdev, "bk_stdev": p.bk_stdev, "max_stdev": p.max_stdev},
        "key_seed": KEY_SEED, "encrypt_seed": ENC_SEED, "input_bits": BITS,
        "cases": [list(c) for c in CASES], "expected_bits": [int(b) for b in plain],
        "mu": 1 << 29,
        "sha256": {k: hashlib.sha256(np.ascontiguousarray(v, dtype=np.int32).tobytes()).hexdigest() for k, v in files.items()},
    }
    return files, meta


def main():
    files, meta = build()
    for name, arr in files.items():
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(np.array(arr, dtype="<i4").tobytes())
    with open(os.path.join(HERE, "kit.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta["sha256"], indent=1))
    print("bytes:", {k: int(np.asarray(v).size * 4) for k, v in files.items()})


if __name__ == "__main__":
    main()
