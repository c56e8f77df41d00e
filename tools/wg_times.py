"""Where does a blind-rotate launch spend its time?  Per-workgroup start and end stamps
(s_memrealtime, constant 100 MHz) of one launch: dispatch ramp, spread of the run times, tail.
The launch is the second of two back to back, so it runs at a half-warm shader clock
(tools/diag/launch_clock.py); the cycle stamps next to the time stamps give that clock."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peba1_amd import api, lib  # noqa: E402

L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
CLK = 100.0       # s_memrealtime ticks per microsecond
for width in (256, 512, 1024, 2048):
    for rep in range(2):
        t = np.zeros(4 * width, dtype=np.uint64)
        assert L.tfhe_hip_test_wg_times(ks.cloud, width, t.ctypes.data_as(C.POINTER(C.c_uint64)), None) == 0
    t = t.reshape(width, 4)
    cu = (t[:, 0] >> np.uint64(48)).astype(np.int64)
    xcc = cu >> 8
    start = t[:, 2].astype(np.int64)
    end = t[:, 3].astype(np.int64)
    cycles = (t[:, 1] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64) - (t[:, 0] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
    dur = (end - start) / CLK / 1e3                              # ms
    rel = np.zeros(width)
    for x in np.unique(xcc):                                      # start relative to the XCD's first workgroup
        m = xcc == x
        rel[m] = (start[m] - start[m].min()) / CLK / 1e3
    fin = rel + dur
    q = lambda a, p: float(np.percentile(a, p))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"width {width:5d}: clock {float(np.median(cycles / ((end - start) / CLK))) / 1e3:5.3f} GHz | CUs used {len(per_cu)}, workgroups per CU min {per_cu.min()} max {per_cu.max()} | "
          f"start p50 {q(rel, 50):6.3f} p90 {q(rel, 90):6.3f} max {rel.max():6.3f} ms | run min {dur.min():5.2f} p10 {q(dur, 10):5.2f} "
          f"p50 {q(dur, 50):5.2f} p90 {q(dur, 90):5.2f} max {dur.max():5.2f} ms | finish max {fin.max():6.3f} ms", flush=True)
    if width == 512:
        early = rel < 0.5
        print("   512: started within 0.5 ms:", int(early.sum()), " run time histogram (ms):",
              np.histogram(dur, bins=[4.0, 4.4, 4.8, 5.2, 5.6, 6.0, 6.4, 6.8])[0].tolist())
        occ = per_cu[np.unique(cu, return_inverse=True)[1]]
        for k in sorted(set(occ)):
            print(f"   workgroups on a CU holding {k}: n {int((occ == k).sum())} run p50 {q(dur[occ == k], 50):5.2f} max {dur[occ == k].max():5.2f}")
