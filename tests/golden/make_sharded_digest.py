#!/usr/bin/env python3
"""Writes tests/golden/sharded_match_digest.json: SHA-256 of every ciphertext that crosses the
exchange step of the SLOT-SHARDED match (peba1_amd/dist.py) and of its 24 output
ciphertexts, evaluated on the CPU ORACLE through this repo's circuit library
(oracle/liboracle_boots.so): 3 slots over 2 logical ranks (ragged: 2 + 1), i.e.

    rank r:  partial_r = peba1_partial_distance(slots of r)          (Math.cpp:351-360 per shard)
    rank 0:  result_b  = peba1_combine_and_compare(partial_0, partial_1, bound)   (Math.cpp:384)

The tree of partial sums is a different gate DAG from the reference's left-to-right ripple, so
its ciphertexts are pinned here against the oracle evaluating the SAME DAG; the GPU test
tests/test_gpu_sharded.py regenerates keys and inputs with the product, runs the sharded path
(logical ranks on one device) and must reproduce every digest bit for bit.
Takes ~15 CPU-minutes (one gate at a time, single thread)."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from peba1_amd.dist import shard_slots  # noqa: E402  (pure-Python slot partition, no library needed)

KEY_SEED, ENC_SEED = 0x5EBA2, 4242
TEMPLATE, PROBE, BOUND, BITS, WORLD = [37, 200, 91], [40, 190, 92], 100, 8, 2


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    B = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_boots.so"))
    V = C.c_void_p
    B.orc_keygen.restype = V
    B.orc_keygen.argtypes = [C.POINTER(O.OrcParams), C.c_uint64]
    B.orc_boots_bind.argtypes = [V, C.c_uint64]
    B.orc_boots_params.restype = V
    B.orc_boots_cloud.restype = V
    B.orc_boots_gate_count.restype = C.c_longlong
    B.new_gate_bootstrapping_ciphertext_array.restype = V
    B.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, V]
    B.bootsSymEncrypt.argtypes = [V, C.c_int32, V]
    B.bootsSymDecrypt.argtypes = [V, V]
    B.orc_boots_export.argtypes = [V, C.c_int32, V]
    B.peba1_partial_distance.argtypes = [V, V, V, C.c_int, C.c_int, V]
    B.peba1_combine_and_compare.argtypes = [V, V, C.c_int, V, V]
    p = O.params("P128")
    ks = B.orc_keygen(C.byref(p), KEY_SEED)
    B.orc_boots_bind(ks, ENC_SEED)
    params, cloud = B.orc_boots_params(), B.orc_boots_cloud()
    SZ = 24

    def enc(v, bits):
        a = B.new_gate_bootstrapping_ciphertext_array(bits, params)
        for i in range(bits):
            B.bootsSymEncrypt(a + i * SZ, (v >> i) & 1, None)
        return a

    def words_of(arr, count):
        w = np.zeros((count, p.n + 1), dtype=np.int32)
        B.orc_boots_export(arr, count, w.ctypes.data_as(V))
        return w

    # encryption order is part of the fixture: per slot template then probe, then the bound
    T, S = [], []
    for t, s in zip(TEMPLATE, PROBE):
        T.append(enc(t, BITS))
        S.append(enc(s, BITS))
    bound = enc(BOUND, 3 * BITS)
    t0 = time.time()
    partials, digests, values = [], [], []
    for r in range(WORLD):
        lo, hi = shard_slots(len(TEMPLATE), WORLD, r)
        part = B.new_gate_bootstrapping_ciphertext_array(24, params)
        B.peba1_partial_distance(part, (V * (hi - lo))(*S[lo:hi]), (V * (hi - lo))(*T[lo:hi]), hi - lo, BITS, cloud)
        w = words_of(part, 24)
        digests.append(hashlib.sha256(w.tobytes()).hexdigest())
        values.append(sum(B.bootsSymDecrypt(part + i * SZ, None) << i for i in range(24)))
        assert values[-1] == sum((a - b) ** 2 for a, b in zip(PROBE[lo:hi], TEMPLATE[lo:hi]))
        partials.append(part)
        print("rank", r, "partial", values[-1], digests[-1], round(time.time() - t0, 1), "s", flush=True)
    rb = B.new_gate_bootstrapping_ciphertext_array(24, params)
    B.peba1_combine_and_compare(rb, (V * WORLD)(*partials), WORLD, bound, cloud)
    words = words_of(rb, 24)
    bit = B.bootsSymDecrypt(rb, None)
    d = sum((a - b) ** 2 for a, b in zip(PROBE, TEMPLATE))
    assert bit == (1 if d > BOUND else 0)
    out = {"params": "P128", "key_seed": KEY_SEED, "encrypt_seed": ENC_SEED, "template": TEMPLATE, "probe": PROBE,
           "bound": BOUND, "bits": BITS, "world": WORLD, "blind_rotates": int(B.orc_boots_gate_count()),
           "match_bit": int(bit), "partial_values": values, "partial_sha256": digests,
           "result_b_sha256": hashlib.sha256(words.tobytes()).hexdigest(),
           "result_b0_sha256": hashlib.sha256(words[0].tobytes()).hexdigest(),
           "oracle_seconds": round(time.time() - t0, 1),
           "circuit": "peba1_partial_distance x world -> peba1_combine_and_compare"}
    with open(os.path.join(ROOT, "tests", "golden", "sharded_match_digest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
