import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from peba1_amd import api, circuits, lib
L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tmpl = [(37*i+11) % 255 for i in range(nslots)]
gen = [t+1 for t in tmpl]
imp = [(91*i+5) % 256 for i in range(nslots)]
T = circuits.EncryptedVector(pp, tmpl, 8, ks).to_device()
G = circuits.EncryptedVector(pp, gen, 8, ks).to_device()
I = circuits.EncryptedVector(pp, imp, 8, ks).to_device()
bound = circuits.encrypt_number(pp, 256, 24, ks)
L.tfhe_hip_set_kernel_timing(1)
api.set_deferred(True)
for name, S, want_d in (("genuine", G, sum((a-b)**2 for a,b in zip(gen,tmpl))), ("impostor", I, sum((a-b)**2 for a,b in zip(imp,tmpl)))):
    rb = api.CiphertextArray(pp, 24)
    api.reset_stats()
    t0 = time.time()
    circuits.function_f(rb, S, T, bound, 8, ks)
    t1 = time.time()
    lv = api.flush()
    t2 = time.time()
    s = api.stats()
    bit = rb.decrypt(ks)[0]
    print(f"{name}: record {t1-t0:.3f}s flush {t2-t1:.3f}s levels {lv} br {s['blind_rotates']} ks {s['keyswitches']} "
          f"ms_br {s['ms_blind_rotate']:.1f} ms_ks {s['ms_keyswitch']:.1f} bit {bit} want {(1 if want_d > 256 else 0)} "
          f"rate {s['blind_rotates']/(t2-t0):.0f} br/s", flush=True)
