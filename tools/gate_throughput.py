"""Microbenchmark: G independent bootsAND on fresh encryptions, per batch size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from peba1_amd import api, lib

def main():
    L = lib.load()
    api.set_deferred(False)      # these timings bracket gate_batch itself: run it to completion
    args = sys.argv[1:]
    p2048, p80 = "--p2048" in args, "--p80" in args
    args = [a for a in args if a not in ("--p2048", "--p80")]
    pp = api.ParameterSet(p2048=True) if p2048 else api.ParameterSet(80 if p80 else 128)
    print(f"params n={pp.n} N={pp.N} l={pp.l} Bgbit={pp.Bgbit}", flush=True)
    t = time.time(); ks = api.SecretKeySet(pp, 0x5EBA2); print("keygen+upload s", time.time() - t, flush=True)
    rng = np.random.default_rng(0)
    sizes = [int(s) for s in (args or ["1", "16", "256", "1024", "4096"])]
    G = max(sizes)
    a = api.CiphertextArray(pp, G).encrypt(rng.integers(0, 2, G), ks)
    b = api.CiphertextArray(pp, G).encrypt(rng.integers(0, 2, G), ks)
    # move inputs to the device once
    a.set_words(a.words()); b.set_words(b.words())
    L.tfhe_hip_set_kernel_timing(1)
    for g in sizes:
        A = api.CiphertextArray(pp, g); B = api.CiphertextArray(pp, g); R = api.CiphertextArray(pp, g)
        A.set_words(a.words()[:g]); B.set_words(b.words()[:g])
        api.gate_batch("AND", R, A, B, ks)   # warm
        api.reset_stats()
        t0 = time.time()
        api.gate_batch("AND", R, A, B, ks)
        dt = time.time() - t0
        s = api.stats()
        print(f"G={g:5d} wall {dt*1e3:9.2f} ms  br {s['ms_blind_rotate']:9.2f} ms  ks {s['ms_keyswitch']:8.2f} ms  "
              f"gates/s {g/dt:10.0f}  (br-only {g/(s['ms_blind_rotate']*1e-3):10.0f})", flush=True)

main()
