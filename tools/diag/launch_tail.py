"""Why does a 512-wide blind-rotate launch (one round of two workgroups per CU) take longer than an eighth of a
4,096-wide one?  Per-workgroup stamps of one launch, in shader cycles (s_memtime) and in constant 100 MHz time
(s_memrealtime), beside the launch's event time: how long the workgroups run, at what clock, how far apart they
start and how long after the last one ends the launch is over."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peba1_amd import api, lib  # noqa: E402

L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
q = lambda a, p: float(np.percentile(a, p))
for width in (1, 256, 512, 512, 768, 1024, 2048, 4096, 512):
    t = np.zeros(4 * width, dtype=np.uint64)
    ms = C.c_double(0)
    assert L.tfhe_hip_test_wg_times(ks.cloud, width, t.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(ms)) == 0
    t = t.reshape(width, 4)
    cyc = ((t[:, 1] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64) - (t[:, 0] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64))
    r0, r1 = t[:, 2].astype(np.int64), t[:, 3].astype(np.int64)
    run_ms = (r1 - r0) / 1e5
    first, last_start, last_end = r0.min(), r0.max(), r1.max()
    print(f"width {width:5d}: launch (events) {ms.value:7.3f} ms | first start -> last end {(last_end - first) / 1e5:7.3f} ms | "
          f"starts spread over {(last_start - first) / 1e5:6.3f} ms | workgroup run p50 {q(run_ms, 50):6.3f} max {run_ms.max():6.3f} ms = "
          f"{q(cyc, 50) / 1e6:6.3f}M cycles -> {q(cyc / run_ms, 50) / 1e6:5.3f} GHz", flush=True)
