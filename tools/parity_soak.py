"""Random-input parity soak of the blind rotation: `count` uniformly random LWE inputs per parameter set, every
extracted sample (a signed permutation of the whole accumulator, so every accumulator word) compared word for word
with the CPU oracle -- whose exact product runs over one 64-bit prime, independent of the two 27-bit primes + signed
CRT of the HIP kernels -- in every form a default build can launch: the 4-wave kernel (wide launch, with and without
the 8-wave tail round), the 8-wave latency form (launches of 256), the split form.  The circuit digests check
hundreds of thousands of rotations, but through later gates' modulus switches, which hide low-bit differences;
this compares the raw words.  Then the same number of key switches of random extracted samples, in every form.

    python tools/parity_soak.py [--ks-only] [count_P128]     (P80: count/2, P2048: count/8; oracle on all host threads)
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O  # noqa: E402   (the checker; this is a diagnostic, not the product path)
from peba1_amd import api  # noqa: E402

ks_only = "--ks-only" in sys.argv[1:]                  # skip the rotations (the slow part of the oracle side)
args = [a for a in sys.argv[1:] if a != "--ks-only"]
count = int(args[0]) if args else 8192
threads = min(32, os.cpu_count() or 8)
total = 0
total_ks = 0
for pname, make, cnt in (("P128", lambda: api.ParameterSet(128), count), ("P80", lambda: api.ParameterSet(80), count // 2),
                         ("P2048", lambda: api.ParameterSet(p2048=True), count // 8)):
    pp = make()
    seed = 0x50AC + pp.n
    ks = api.SecretKeySet(pp, seed, device=True)
    oks = O.KeySet(O.params(pname), seed)
    rng = np.random.default_rng(pp.N + cnt)
    lin = rng.integers(-2**31, 2**31, (cnt, pp.words), dtype=np.int64).astype(np.int32)
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:                       # ctypes releases the GIL; the oracle is re-entrant
        want = None if ks_only else np.stack(list(ex.map(oks.bootstrap_woks, lin)))
    t_cpu = time.time() - t0
    forms = [("default (4-wave, tail round on the 8-wave form)" if pp.N == 1024 else "default (split form)", {}, cnt),
             ("one launch per level (br_tail8 = 0)", {"br_tail8": 0}, cnt),
             ("launches of 256" + (" (8-wave form)" if pp.N == 1024 else ""), {}, 256),
             ("4-wave form, no digit table", {"br_variant": 0, "br_digit_table": 0, "br8_max_rotations": 0}, cnt),
             ("split form", {"br_variant": 2}, cnt)]
    if ks_only:
        forms = []
    try:
        for label, tunings, chunk in forms:
            for k, v in tunings.items():
                api.set_tuning(k, v)
            t0 = time.time()
            got = np.concatenate([api.kernel_bootstrap_woks(ks, lin[i:i + chunk]) for i in range(0, cnt, chunk)])
            bad = int((got != want).any(axis=1).sum())
            print(f"{pname}: {cnt} random rotations, {label}: {bad} differ from the oracle ({time.time() - t0:.2f} s GPU side)", flush=True)
            assert bad == 0, (pname, label, np.argwhere(got != want)[:4])
            total += cnt
            for k in tunings:
                api.set_tuning(k, {"br_tail8": 1, "br_variant": -1, "br_digit_table": 1, "br8_max_rotations": 1 << 30}[k])
    finally:
        for k, v in (("br_tail8", 1), ("br_variant", -1), ("br_digit_table", 1), ("br8_max_rotations", 1 << 30)):
            api.set_tuning(k, v)
    if not ks_only:
        print(f"{pname}: oracle {t_cpu:.1f} s on {threads} threads ({cnt / t_cpu:.0f} rotations/s)", flush=True)
    # key switch of the same number of uniformly random extracted samples: index form (tiles of 16, 24, 32), LDS-strip
    # form, per-gate form
    u = rng.integers(-2**31, 2**31, (cnt, pp.k * pp.N + 1), dtype=np.int64).astype(np.int32)
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        want_ks = np.stack(list(ex.map(oks.keyswitch, u)))
    t_cpu = time.time() - t0
    try:
        for tile, index in ((16, 1), (24, 1), (32, 1), (16, 0), (0, 1)):
            api.set_tuning("ks_tile", tile)
            api.set_tuning("ks_index", index)
            got = api.kernel_keyswitch(ks, u)
            bad = int((got != want_ks).any(axis=1).sum())
            print(f"{pname}: {cnt} random key switches, tile {tile}, {'index' if index else 'LDS-strip'} form: {bad} differ from the oracle", flush=True)
            assert bad == 0, (pname, tile, index)
            total_ks += cnt
    finally:
        api.set_tuning("ks_tile", 16)
        api.set_tuning("ks_index", 1)
    print(f"{pname}: oracle key switches {t_cpu:.1f} s on {threads} threads", flush=True)
    ks.close()
print(f"OK: {total} rotation results and {total_ks} key-switch results equal the oracle's, word for word")
