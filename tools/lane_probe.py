"""How much would overlapping level-synchronous chains gain?  Runs `levels` rounds of (blind
rotate + key switch) over `width` random gates as 1, 2, 4 independent chains on as many HIP
streams (each chain: width/lanes gates per round, strictly ordered) and prints gates/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peba1_amd import api, lib  # noqa: E402

L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
levels = 24
for width in (512, 1024):
    for lanes in (1, 2, 4, 8):
        L.tfhe_hip_test_lane_probe(ks.cloud, lanes, 2, width)            # warm
        ms = L.tfhe_hip_test_lane_probe(ks.cloud, lanes, levels, width)
        print(f"width {width:5d} lanes {lanes}: {ms / levels:7.3f} ms/round  {width * levels / ms * 1e3:9.0f} gates/s", flush=True)
