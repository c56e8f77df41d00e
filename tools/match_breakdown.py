"""Where one Function_f match spends its wall time: host recording, scheduling + descriptor
build, device kernels (events), and the rest (launch gaps, syncs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peba1_amd import api, circuits, lib
L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
base = [(37 * i + 11) % 255 for i in range(128)]
probe = circuits.EncryptedVector(pp, [v + 1 for v in base], 8, ks).to_device()
tmpl = circuits.EncryptedVector(pp, base, 8, ks).to_device()
bound = circuits.encrypt_number(pp, 256, 24, ks); bound.set_words(bound.words())
api.set_deferred(True)
for timing in (0, 1, 1):
    L.tfhe_hip_set_kernel_timing(timing)
    api.reset_stats()
    t0 = time.perf_counter()
    rb = api.CiphertextArray(pp, 24)
    circuits.function_f(rb, probe, tmpl, bound, 8, ks)
    t1 = time.perf_counter()
    api.flush()
    t2 = time.perf_counter()
    s = api.stats()
    print(f"kernel_timing={timing}: record {1e3*(t1-t0):.1f} ms, flush {1e3*(t2-t1):.1f} ms (engine wall {s['ms_flush_wall']:.1f}, "
          f"BR events {s['ms_blind_rotate']:.1f}, KS events {s['ms_keyswitch']:.1f}) -> host scheduling+plan "
          f"{1e3*(t2-t1)-s['ms_flush_wall']:.1f} ms, in-engine non-kernel {s['ms_flush_wall']-s['ms_blind_rotate']-s['ms_keyswitch']:.1f} ms", flush=True)
