"""Blind-rotate launch time vs number of rotations, for both kernel forms (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peba1_amd import api, lib
L = lib.load()
api.set_deferred(False)      # these timings bracket gate_batch itself: run it to completion
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
rng = np.random.default_rng(0)
G = 2048
a = api.CiphertextArray(pp, G).encrypt(rng.integers(0, 2, G), ks)
b = api.CiphertextArray(pp, G).encrypt(rng.integers(0, 2, G), ks)
wa, wb = a.words(), b.words()
L.tfhe_hip_set_kernel_timing(1)
for form, br4 in (("4-wave", 1 << 20), ("2-wave", 0)):
    api.set_tuning("br4_max_rotations", br4)
    for g in (64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 2048):
        A = api.CiphertextArray(pp, g).set_words(wa[:g]); B = api.CiphertextArray(pp, g).set_words(wb[:g]); R = api.CiphertextArray(pp, g)
        api.gate_batch("AND", R, A, B, ks)
        api.reset_stats()
        api.gate_batch("AND", R, A, B, ks)
        s = api.stats()
        print(f"{form} G={g:5d} br {s['ms_blind_rotate']:7.2f} ms  {s['ms_blind_rotate']*1e3/g:6.2f} us/gate  ks {s['ms_keyswitch']:5.2f}", flush=True)
