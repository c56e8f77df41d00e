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


def synthetic_bytes(n, seed):
    """Uniform bytes from a fixed LCG (so the GPU test regenerates them without numpy's generators); the probe's bytes
    avoid 0 (the reference's subtractor is wrong for a zero subtrahend)."""
    x, out = seed, []
    for _ in range(n):
        x = (x * 1103515245 + 12345) % (1 << 31)
        out.append((x >> 16) & 255)
    return out


def slots256():
    """--slots256: BASELINE configs[2] at its own size -- 256 slots x 8 bit, uniform bytes, sharded over 8 ranks of 32 slots.
    Writes tests/golden/sharded_match_256_digest.json: SHA-256 of each rank's 24-ciphertext partial sum (what crosses the
    exchange) and of the 24 outputs of rank 0's combine, in both forms (pairwise tree of the reference's ripple adders +
    its comparator; carry-save / prefix form).  432k blind rotations: the oracle's recording mode evaluates the DAG level
    by level on --threads host threads (about two hours on 7)."""
    threads = int(sys.argv[sys.argv.index("--threads") + 1]) if "--threads" in sys.argv else 7
    nslots, world, bits = 256, 8, 8
    template = synthetic_bytes(nslots, 0xC0F3)
    probe = [v or 1 for v in synthetic_bytes(nslots, 0xC0F4)]
    d = sum((a - b) ** 2 for a, b in zip(probe, template))
    bound_v = d - 1                                      # just under the distance: the comparator's whole chain matters
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    B = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_boots.so"))
    V = C.c_void_p
    B.orc_keygen.restype = V
    B.orc_keygen.argtypes = [C.POINTER(O.OrcParams), C.c_uint64]
    B.orc_boots_bind.argtypes = [V, C.c_uint64]
    B.orc_boots_params.restype = V
    B.orc_boots_cloud.restype = V
    B.orc_boots_gate_count.restype = C.c_longlong
    B.orc_boots_unique_rotations.restype = C.c_longlong
    B.new_gate_bootstrapping_ciphertext_array.restype = V
    B.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, V]
    B.bootsSymEncrypt.argtypes = [V, C.c_int32, V]
    B.bootsSymDecrypt.argtypes = [V, V]
    B.orc_boots_export.argtypes = [V, C.c_int32, V]
    B.peba1_partial_distance.argtypes = [V, V, V, C.c_int, C.c_int, V]
    B.peba1_combine_and_compare.argtypes = [V, V, C.c_int, V, V]
    B.peba1_combine_and_compare_fast.argtypes = [V, V, C.c_int, V, V]
    p = O.params("P128")
    ks = B.orc_keygen(C.byref(p), KEY_SEED)
    B.orc_boots_bind(ks, ENC_SEED + 256)
    B.orc_boots_set_recording(threads)
    params, cloud, SZ = B.orc_boots_params(), B.orc_boots_cloud(), 24

    def enc(v, nb):
        a = B.new_gate_bootstrapping_ciphertext_array(nb, params)
        for i in range(nb):
            B.bootsSymEncrypt(a + i * SZ, (v >> i) & 1, None)
        return a

    def words_of(arr, count):
        w = np.zeros((count, p.n + 1), dtype=np.int32)
        B.orc_boots_export(arr, count, w.ctypes.data_as(V))
        return w

    T, S = [], []
    for t, s in zip(template, probe):                    # encryption order: per slot template then probe, then the bound
        T.append(enc(t, bits))
        S.append(enc(s, bits))
    bound = enc(bound_v, 3 * bits)
    t0 = time.time()
    partials = []
    for r in range(world):                               # recorded only: all ranks' DAGs are evaluated together below
        lo, hi = shard_slots(nslots, world, r)
        part = B.new_gate_bootstrapping_ciphertext_array(24, params)
        B.peba1_partial_distance(part, (V * (hi - lo))(*S[lo:hi]), (V * (hi - lo))(*T[lo:hi]), hi - lo, bits, cloud)
        partials.append(part)
    rb = B.new_gate_bootstrapping_ciphertext_array(24, params)
    B.peba1_combine_and_compare(rb, (V * world)(*partials), world, bound, cloud)
    rbf = B.new_gate_bootstrapping_ciphertext_array(24, params)
    B.peba1_combine_and_compare_fast(rbf, (V * world)(*partials), world, bound, cloud)
    digests, values = [], []
    for r, part in enumerate(partials):
        w = words_of(part, 24)                           # the first export evaluates the whole recording
        digests.append(hashlib.sha256(w.tobytes()).hexdigest())
        values.append(sum(B.bootsSymDecrypt(part + i * SZ, None) << i for i in range(24)))
        lo, hi = shard_slots(nslots, world, r)
        assert values[-1] == sum((a - b) ** 2 for a, b in zip(probe[lo:hi], template[lo:hi])), (r, values[-1])
        print("rank", r, "partial", values[-1], digests[-1], round(time.time() - t0, 1), "s", flush=True)
    words, words_fast = words_of(rb, 24), words_of(rbf, 24)
    bit, bit_fast = B.bootsSymDecrypt(rb, None), B.bootsSymDecrypt(rbf, None)
    assert bit == bit_fast == 1
    out = {"params": "P128", "key_seed": KEY_SEED, "encrypt_seed": ENC_SEED + 256, "nslots": nslots, "bits": bits, "world": world,
           "template_lcg_seed": 0xC0F3, "probe_lcg_seed": 0xC0F4, "distance": d, "bound": bound_v, "match_bit": int(bit),
           "blind_rotates_recorded": int(B.orc_boots_gate_count()), "blind_rotates_evaluated": int(B.orc_boots_unique_rotations()),
           "partial_values": values, "partial_sha256": digests,
           "result_b_sha256": hashlib.sha256(words.tobytes()).hexdigest(),
           "result_b_fast_sha256": hashlib.sha256(words_fast.tobytes()).hexdigest(),
           "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": threads,
           "circuit": "peba1_partial_distance x 8 -> peba1_combine_and_compare | peba1_combine_and_compare_fast"}
    with open(os.path.join(ROOT, "tests", "golden", "sharded_match_256_digest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


def latency_form():
    """--latency-form: the sharded match with BOTH phases depth-optimised (PEBA1_DIST_FAST_PARTIAL | PEBA1_DIST_FAST_COMBINE:
    peba1_euclidean_distance_fast per rank -> gather -> peba1_combine_and_compare_fast), 5 slots over 2 logical ranks
    (3 + 2), on both sides of the threshold.  Writes tests/golden/sharded_match_latency_form_digest.json."""
    threads = int(sys.argv[sys.argv.index("--threads") + 1]) if "--threads" in sys.argv else 4
    template, probe, bits, world = [37, 200, 91, 5, 255], [40, 190, 92, 250, 1], 8, 2
    d = sum((a - b) ** 2 for a, b in zip(probe, template))
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
    B.peba1_euclidean_distance_fast.argtypes = [V, V, V, C.c_int, C.c_int, V]
    B.peba1_combine_and_compare_fast.argtypes = [V, V, C.c_int, V, V]
    p = O.params("P128")
    ks = B.orc_keygen(C.byref(p), KEY_SEED)
    B.orc_boots_bind(ks, ENC_SEED + 77)
    B.orc_boots_set_recording(threads)
    params, cloud, SZ = B.orc_boots_params(), B.orc_boots_cloud(), 24

    def enc(v, nb):
        a = B.new_gate_bootstrapping_ciphertext_array(nb, params)
        for i in range(nb):
            B.bootsSymEncrypt(a + i * SZ, (v >> i) & 1, None)
        return a

    def words_of(arr, count):
        w = np.zeros((count, p.n + 1), dtype=np.int32)
        B.orc_boots_export(arr, count, w.ctypes.data_as(V))
        return w

    T, S = [], []
    for t, s in zip(template, probe):                    # encryption order: per slot template then probe, then the two bounds
        T.append(enc(t, bits))
        S.append(enc(s, bits))
    bounds = [enc(d - 1, 3 * bits), enc(d, 3 * bits)]
    t0 = time.time()
    partials = []
    for r in range(world):
        lo, hi = shard_slots(len(template), world, r)
        part = B.new_gate_bootstrapping_ciphertext_array(24, params)
        B.peba1_euclidean_distance_fast(part, (V * (hi - lo))(*S[lo:hi]), (V * (hi - lo))(*T[lo:hi]), hi - lo, bits, cloud)
        partials.append(part)
    results = []
    for b in bounds:
        rb = B.new_gate_bootstrapping_ciphertext_array(24, params)
        B.peba1_combine_and_compare_fast(rb, (V * world)(*partials), world, b, cloud)
        results.append(rb)
    digests, values = [], []
    for r, part in enumerate(partials):
        w = words_of(part, 24)
        digests.append(hashlib.sha256(w.tobytes()).hexdigest())
        values.append(sum(B.bootsSymDecrypt(part + i * SZ, None) << i for i in range(24)))
        lo, hi = shard_slots(len(template), world, r)
        assert values[-1] == sum((a - b) ** 2 for a, b in zip(probe[lo:hi], template[lo:hi])), (r, values[-1])
    bits_out = [int(B.bootsSymDecrypt(rb, None)) for rb in results]
    assert bits_out == [1, 0], bits_out
    out = {"params": "P128", "key_seed": KEY_SEED, "encrypt_seed": ENC_SEED + 77, "template": template, "probe": probe,
           "bits": bits, "world": world, "distance": d, "bounds": [d - 1, d], "match_bits": bits_out,
           "blind_rotates_recorded": int(B.orc_boots_gate_count()), "partial_values": values, "partial_sha256": digests,
           "result_b_sha256": [hashlib.sha256(words_of(rb, 24).tobytes()).hexdigest() for rb in results],
           "oracle_seconds": round(time.time() - t0, 1),
           "circuit": "peba1_euclidean_distance_fast x 2 -> peba1_combine_and_compare_fast (bounds d-1, d)"}
    with open(os.path.join(ROOT, "tests", "golden", "sharded_match_latency_form_digest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


def main():
    if "--slots256" in sys.argv[1:]:
        return slots256()
    if "--latency-form" in sys.argv[1:]:
        return latency_form()
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
