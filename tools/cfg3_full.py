"""BASELINE configs[3] at its full size on ONE GPU: a probe against 1,024 enrolled 128-slot templates, 128 per rank, the
8 ranks' shares run one after the other as logical ranks (each through libpeba1-dist's peba1_identify, 8 matches per
pipelined flush, the reference's gate sequence, library defaults), every match bit checked against the plaintext rule and
the genuine template the only hit among the 1,024.  Per-rank times are what each of 8 GPUs would take; their maximum is
the projected wall time of the sharded batch (a projection from logical ranks, not a multi-GPU measurement).

    python tools/cfg3_full.py <first rank> <last rank + 1> [matches per rank, default 128]
(gpurun calls are limited to 20 minutes: ranks 0-3 and 4-7 in two calls)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peba1_amd import api, circuits, identify, lib  # noqa: E402


def main():
    r0, r1 = int(sys.argv[1]), int(sys.argv[2])
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    world, nslots, bitsize = 8, 128, 8
    L = lib.load()
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, 0x5EBA2, device=True)
    base = [((37 * i + 11) % 255) or 1 for i in range(nslots)]
    probe_v = [v + 1 for v in base]
    genuine = 5 * M + 77 % M                      # one enrolled template matches: rank 5
    L.tfhe_hip_set_encrypt_seed(31337)
    probe = circuits.EncryptedVector(pp, probe_v, bitsize, ks).to_device()
    bound = circuits.encrypt_number(pp, 256, 3 * bitsize, ks)
    bound.set_words(bound.words())
    out = []
    for rank in range(r0, r1):
        tv = [identify.synthetic_template(base, rank * M + m + 1) for m in range(M)]
        if rank * M <= genuine < (rank + 1) * M:
            tv[genuine - rank * M] = base
        templates = [circuits.EncryptedVector(pp, t, bitsize, ks).to_device() for t in tv]
        api.reset_stats()
        t0 = time.time()
        bits = identify.identify(pp, ks, probe, templates, bound, bitsize, group=8)
        api.wait()
        dt = time.time() - t0
        st = api.stats()
        got = [int(b) for b in bits.decrypt(ks)]
        want = [1 if sum((a - b) ** 2 for a, b in zip(probe_v, t)) > 256 else 0 for t in tv]
        assert got == want, f"rank {rank}: match bits differ from the plaintext rule"
        hits = [rank * M + m for m, b in enumerate(got) if b == 0]
        rec = {"rank": rank, "matches": M, "seconds": dt, "ms_per_match": dt / M * 1e3, "executed_rotations": int(st["blind_rotates"]),
               "recorded_rotations": M * 215544, "gates_shared": int(st["reused_gates"]), "gates_dropped_as_dead": int(st["dead_gates"]),
               "flushes": int(st["flushes"]), "hits": hits}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del templates, bits
    print("CFG3-PART-OK ranks", r0, r1 - 1, "world", world, flush=True)


main()
