"""Repeat the full 128-slot match and check that every run produces the same 24 output
ciphertexts word for word (scheduling, priorities and atomics must never leak into results)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peba1_amd import api, circuits, lib  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
L.tfhe_hip_set_encrypt_seed(99)
base = [(37 * i + 11) % 255 for i in range(128)]
probe = circuits.EncryptedVector(pp, [(91 * i + 5) % 256 for i in range(128)], 8, ks).to_device()
tmpl = circuits.EncryptedVector(pp, base, 8, ks).to_device()
bound = circuits.encrypt_number(pp, 256, 24, ks)
api.set_deferred(True)
digests = set()
for r in range(runs):
    for fn, tag in ((circuits.function_f, "ref"), (circuits.function_f_fast, "fast")):
        rb = api.CiphertextArray(pp, 24)
        t0 = time.time()
        fn(rb, probe, tmpl, bound, 8, ks)
        api.flush()
        d = hashlib.sha256(rb.words().tobytes()).hexdigest()
        digests.add((tag, d))
        print(f"run {r} {tag}: {time.time() - t0:.3f} s  bit {int(rb.decrypt(ks)[0])}  sha256 {d[:16]}", flush=True)
assert len(digests) == 2, digests
print("OK: identical ciphertexts in every run")
