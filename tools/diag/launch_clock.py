"""At what shader clock does a blind-rotate launch run, and what does that depend on?  Stamped 512-wide launches
after different histories (TFHE_HIP_PROBE_WARM), and the clock of the workgroups of one 4,096-wide launch by the
time they started."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peba1_amd import api, lib  # noqa: E402

L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)


def probe(width, warm=None):
    if warm is None:
        os.environ.pop("TFHE_HIP_PROBE_WARM", None)
    else:
        os.environ["TFHE_HIP_PROBE_WARM"] = warm
    t = np.zeros(4 * width, dtype=np.uint64)
    ms = C.c_double(0)
    assert L.tfhe_hip_test_wg_times(ks.cloud, width, t.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(ms)) == 0
    t = t.reshape(width, 4)
    cyc = ((t[:, 1] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64) - (t[:, 0] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64))
    r0, r1 = t[:, 2].astype(np.int64), t[:, 3].astype(np.int64)
    return ms.value, cyc, r0, r1


for warm in (None, "0:512", "8:512", "32:512", "8:512:1", "32:512:1", "1:4096"):
    for rep in range(2):
        ms, cyc, r0, r1 = probe(512 if warm != "1:4096" else 4096, warm)
        if warm == "1:4096":
            break
        ghz = cyc / ((r1 - r0) / 1e5) / 1e6
        print(f"512-wide after warm-up {str(warm):9s}: launch {ms:6.3f} ms  clock p50 {np.median(ghz):5.3f} GHz  cycles p50 {np.median(cyc) / 1e6:6.3f}M", flush=True)

ms, cyc, r0, r1 = probe(4096, None)
ghz = cyc / ((r1 - r0) / 1e5) / 1e6
order = np.argsort(r0)
print(f"4096-wide: launch {ms:6.3f} ms; clock of the workgroups by start time (eighths):",
      " ".join(f"{np.median(ghz[order[k * 512:(k + 1) * 512]]):5.3f}" for k in range(8)), flush=True)
# within a workgroup's life the clock is an average; finer: clock of 512-wide launches issued one after another
os.environ["TFHE_HIP_PROBE_WARM"] = "0:512"
for k in range(6):
    ms, cyc, r0, r1 = probe(512, "0:512")
    ghz = cyc / ((r1 - r0) / 1e5) / 1e6
    print(f"512-wide, alone, repeat {k}: launch {ms:6.3f} ms  clock {np.median(ghz):5.3f} GHz", flush=True)
