"""Cross-lane (v_permlane32_swap / v_permlane16_swap / DPP) against LDS for the first transpose of the forward NTT
(VERDICT r2 item 8): negacyclic test kernel, N = 1024.  Checks that both forms give the same products, then launches
the timing forms (the forward transform repeated 64 times per workgroup) so that rocprofv3 --kernel-trace --stats of
this command compares negacyclic_kernel<10, false, 64> (LDS) with negacyclic_kernel<10, true, 64> (cross-lane)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peba1_amd import api  # noqa: E402


def main():
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, 0x5EBA2)
    rng = np.random.default_rng(2)
    count = 16384
    ip = rng.integers(-64, 64, (count, pp.N), dtype=np.int64).astype(np.int32)
    tp = rng.integers(-2**31, 2**31, (count, pp.N), dtype=np.int64).astype(np.int32)
    api.set_tuning("br_variant", 0)
    ref = api.kernel_negacyclic(ks, ip, tp)
    api.set_tuning("br_variant", 3)
    xl = api.kernel_negacyclic(ks, ip, tp)
    assert (ref == xl).all(), "cross-lane transpose changes the product"
    print("cross-lane form == LDS form on", count, "products", flush=True)
    for rep in range(3):
        for v, name in ((4, "lds x64"), (5, "cross-lane x64")):
            api.set_tuning("br_variant", v)
            t = time.perf_counter()
            api.kernel_negacyclic(ks, ip, tp)
            print(f"{name}: {1e3 * (time.perf_counter() - t):.1f} ms wall (incl. transfers)", flush=True)
    api.set_tuning("br_variant", -1)
    ks.close()


main()
