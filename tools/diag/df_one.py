import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from peba1_amd import api, lib
L = lib.load(); pp = api.ParameterSet(128); ks = api.SecretKeySet(pp, 0x5EBA2)
a = api.CiphertextArray(pp, 4).encrypt([0,0,1,1], ks); b = api.CiphertextArray(pp, 4).encrypt([0,1,0,1], ks)
r = api.CiphertextArray(pp, 4)
print("launch", flush=True)
api.gate_batch("AND", r, a, b, ks)
print("done", list(r.decrypt(ks)), flush=True)
