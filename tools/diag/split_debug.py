"""where does the split negacyclic product differ from the unsplit one?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from peba1_amd import api

for pp in (api.ParameterSet(p2048=True), api.ParameterSet(128)):
    ks = api.SecretKeySet(pp, 0x51, device=True)
    N = pp.N
    rng = np.random.default_rng(11)
    ip = np.zeros((8, N), dtype=np.int32)
    tp = np.zeros((8, N), dtype=np.int32)
    tp[:, 0] = 1                       # times 1: res = ip
    ip[0, 0] = 1
    ip[1, 1] = 1
    ip[2, N // 2] = 1
    ip[3, N - 1] = 1
    ip[4, :] = rng.integers(-64, 64, N)
    ip[5, :] = rng.integers(-64, 64, N); tp[5, :] = rng.integers(-2**31, 2**31, N)
    ip[6, 64] = 1
    ip[7, 3] = 5; tp[7, 0] = 0; tp[7, 1] = 1
    api.set_tuning("br_variant", 2)
    got = api.kernel_negacyclic(ks, ip, tp)
    api.set_tuning("br_variant", -1)
    plain = api.kernel_negacyclic(ks, ip, tp)
    print("N", N)
    for c in range(8):
        bad = np.nonzero(got[c] != plain[c])[0]
        print(" row", c, "mismatches", len(bad), "first", bad[:8], "got", got[c][bad[:4]], "want", plain[c][bad[:4]])
        if c < 4 or c == 6:
            print("    nonzero got", np.nonzero(got[c])[0][:8], got[c][np.nonzero(got[c])[0][:8]])
    ks.close()

print("== the test's inputs")
for pp in (api.ParameterSet(128), api.ParameterSet(p2048=True)):
    ks = api.SecretKeySet(pp, 0x51, device=True)
    rng = np.random.default_rng(11)
    count, N = 10, pp.N
    ip = rng.integers(-2**31, 2**31, (count, N), dtype=np.int64).astype(np.int32)
    tp = rng.integers(-2**31, 2**31, (count, N), dtype=np.int64).astype(np.int32)
    ip[0:6] = rng.integers(-64, 64, (6, N))
    ip[0, :] = -64; tp[0, :] = -2**31
    ip[1, :] = 63; tp[1, :] = 2**31 - 1
    ip[2, :] = 0
    ip[3, :] = 0; ip[3, N - 1] = 1
    ip[4, :] = 0; ip[4, N // 2] = 1
    ip[6, :] = 2**31 - 1; ip[7, :] = -2**31
    api.set_tuning("br_variant", 2)
    got = api.kernel_negacyclic(ks, ip, tp)
    api.set_tuning("br_variant", -1)
    plain = api.kernel_negacyclic(ks, ip, tp)
    plain2 = api.kernel_negacyclic(ks, ip, tp)
    print("N", N, "unsplit repeatable", (plain == plain2).all())
    for c in range(count):
        bad = np.nonzero(got[c] != plain[c])[0]
        print(" row", c, "mismatches", len(bad), "first", bad[:8], "got", got[c][bad[:3]], "want", plain[c][bad[:3]])
    ks.close()
