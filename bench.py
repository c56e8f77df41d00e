#!/usr/bin/env python3
"""bench.py -- bootstrapped gates/s of the PEBA1 match path on MI355X.

A "step" is ONE encrypted match: Function_f (squared-Euclidean distance of a
128-slot x 8-bit probe against a template, then the threshold comparator;
/root/reference/src/Math.cpp:379-387) = 215,544 blind rotations + 215,496 key
switches, TFHE default 128-bit parameters (n=630, N=1024, k=1, l=3, Bg=2^7).  Inputs
(probe, template, threshold ciphertexts) and the evaluation keys are resident in HBM
before the timed region.  With --gpus N every rank runs its own independent match
against its own template (1-to-N identification, weak scaling); the only exchange is
one RCCL gather of the N match-bit ciphertexts to rank 0 per step.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` counts blind rotations (a MUX is two).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes(pp):
    """bytes per blind rotate / key switch / ciphertext, no cross-gate reuse,
    bootstrapping key counted at 8 B per coefficient (the density this engine stores: two
    32-bit residues)."""
    kpl = (pp.k + 1) * pp.l
    a_br = pp.n * kpl * (pp.k + 1) * pp.N * 8
    a_ks = pp.N * pp.k * pp.ks_t * (1.0 - 2.0 ** (-pp.ks_basebit)) * (pp.n + 1) * 4
    ct = (pp.n + 1) * 4
    return a_br, a_ks, ct


def cpu_baseline(seed):
    """The oracle (exact-integer C port, oracle/) timed on this box's host cores on a bounded
    sample of the same work: independent bootsAND gates, all cores."""
    from oracle import pyoracle as O
    # the GPU box gives one GPU's share of the host (16 cores), whatever cpu_count() says
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    count = 12 * cores                                     # ~20 core-seconds of exact gates
    oks = O.KeySet(O.params("P128"), seed)
    r = O.Rng(77)
    import numpy as np
    a = oks.encrypt(r, np.arange(count) & 1)
    b = oks.encrypt(r, (np.arange(count) >> 1) & 1)
    oks.gate_batch("AND", a[:cores], b[:cores], nthreads=cores)      # warm caches / tables
    t0 = time.perf_counter()
    out = oks.gate_batch("AND", a, b, nthreads=cores)
    dt = time.perf_counter() - t0
    assert list(oks.decrypt(out)) == [int(x & y) for x, y in zip(np.arange(count) & 1, (np.arange(count) >> 1) & 1)]
    t1 = time.perf_counter()
    oks.gate_batch("AND", a[:2], b[:2], nthreads=1)
    single = 2.0 / (time.perf_counter() - t1)
    # beside it: the same gates through an fp64 FFT product, the way upstream TFHE multiplies
    # (oracle mode 3: approximate, NOT the oracle; a cost-faithful stand-in for upstream's CPU path)
    nf = 3 * count
    af, bf = np.tile(a, (3, 1)), np.tile(b, (3, 1))
    oks.gate_batch("AND", af[:cores], bf[:cores], nthreads=cores, use_ntt=3)
    t2 = time.perf_counter()
    outf = oks.gate_batch("AND", af, bf, nthreads=cores, use_ntt=3)
    dtf = time.perf_counter() - t2
    assert list(oks.decrypt(outf[:count])) == list(oks.decrypt(out))
    t3 = time.perf_counter()
    oks.gate_batch("AND", a[:4], b[:4], nthreads=1, use_ntt=3)
    single_f = 4.0 / (time.perf_counter() - t3)
    return {"value": count / dt, "unit": "bootstrapped gates/s", "cores": cores, "kind": "port",
            "sample": f"{count} independent bootsAND (P128) on {cores} threads, oracle exact-integer two-prime NTT "
                      f"(scalar C); 1 thread: {single:.2f} gates/s",
            "fft_standin": {"value": nf / dtf, "unit": "bootstrapped gates/s", "cores": cores,
                            "single_thread": single_f,
                            "note": "same gates with an fp64-FFT negacyclic product (plain radix-2 C, not "
                                    "spqlios): stands in for upstream TFHE's CPU path, which is absent here; "
                                    "approximate arithmetic, not the parity oracle"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slots", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-extra", type=int, default=4,
                    help="after the timed single-match steps, also run this many matches recorded together "
                         "(1-to-N identification, BASELINE configs[3]) and report their rate; 0 = skip")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (exercises the N>1 code path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from peba1_amd import api, circuits, lib
    L = lib.load()
    L.tfhe_hip_set_device(local_rank)
    seed = 0x5EBA2
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, seed, device=True)           # same key on every rank (replicated evaluation keys)
    nslots, bitsize = args.slots, 8
    # synthetic inputs of SURVEY.md 8(d): probe = genuine sample; rank r matches it against template r
    base = [(37 * i + 11) % 255 for i in range(nslots)]
    probe_vals = [v + 1 for v in base]
    tmpl_vals = base if rank == 0 else [(v + 29 * rank + 3 * i) % 256 for i, v in enumerate(base)]
    dist2 = sum((a - b) ** 2 for a, b in zip(probe_vals, tmpl_vals))
    threshold = 256
    L.tfhe_hip_set_encrypt_seed(1000 + rank)
    probe = circuits.EncryptedVector(pp, probe_vals, bitsize, ks).to_device()
    tmpl = circuits.EncryptedVector(pp, tmpl_vals, bitsize, ks).to_device()
    bound = circuits.encrypt_number(pp, threshold, 3 * bitsize, ks)
    bound.set_words(bound.words())
    L.tfhe_hip_set_kernel_timing(1)
    api.set_deferred(True)
    # The headline executes every gate the circuit records: the library's sharing of identical
    # pending gates (tuning "reuse_gates", on by default) is switched off for the timed steps
    # and reported separately below, so that `value` counts 215,544 blind rotations per match.
    api.set_tuning("reuse_gates", 0)

    gather_buf = None
    if use_dist:
        mine = torch.empty(pp.words, dtype=torch.int32, device="cuda")
        gather_buf = [torch.empty(pp.words, dtype=torch.int32, device="cuda") for _ in range(world)] if rank == 0 else None

    def one_match():
        rb = api.CiphertextArray(pp, 3 * bitsize)
        circuits.function_f(rb, probe, tmpl, bound, bitsize, ks)   # records ~350k API calls
        api.flush()                                                # levelised batched execution
        if use_dist:    # the exchange step: match-bit ciphertexts to rank 0 over RCCL
            L.tfhe_hip_export_samples_device(rb.ptr, 1, pp.ptr, mine.data_ptr())
            dist.gather(mine, gather_buf, dst=0)
        return rb

    def sync():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    last = None
    for _ in range(args.warmup):
        last = one_match()
    sync()
    api.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_match()
    sync()
    elapsed = time.perf_counter() - t0
    st = api.stats()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # correctness of what was timed: the decrypted match bit is (distance > threshold)
    bit = int(last.decrypt(ks)[0])
    assert bit == (1 if dist2 > threshold else 0), f"rank {rank}: match bit {bit}, distance {dist2}"
    ok_all = True
    if use_dist and rank == 0:
        tmp = api.CiphertextArray(pp, 1)
        for r in range(world):
            L.tfhe_hip_import_samples_device(tmp.ptr, 1, pp.ptr, gather_buf[r].data_ptr())
            ok_all &= int(tmp.decrypt(ks)[0]) in (0, 1)
        # rank 0's own entry must be its own match bit, bit for bit
        L.tfhe_hip_import_samples_device(tmp.ptr, 1, pp.ptr, gather_buf[0].data_ptr())
        assert (tmp.words()[0] == last.words()[0]).all() and ok_all, "gathered match-bit ciphertext differs"

    if rank == 0:
        a_br, a_ks, ct = algorithmic_bytes(pp)
        rot_per_match = st["blind_rotates"] / max(1, args.steps)
        value = world * st["blind_rotates"] / elapsed
        br_gbps = st["blind_rotates"] * a_br / (st["ms_blind_rotate"] * 1e-3) / 1e9 if st["ms_blind_rotate"] else 0.0
        ks_gbps = st["keyswitches"] * a_ks / (st["ms_keyswitch"] * 1e-3) / 1e9 if st["ms_keyswitch"] else 0.0
        # HBM-side bytes per blind-rotate launch from the PMC passes (tools/pmc_summary.py); the
        # counters cannot be read inside this process, so the committed summary of the same
        # command is quoted, in GB like `achieved` is in GB/s
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_blind_rotate.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "bootstrapped gates/sec (blind rotations/s) over one PEBA1 match, and end-to-end match ms",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"Function_f: {nslots} slots x {bitsize} bit template match, TFHE P128 "
                                   f"(n={pp.n}, N={pp.N}, k={pp.k}, l={pp.l}, Bg=2^{pp.Bgbit}), "
                                   f"{int(rot_per_match)} blind rotations per match (every recorded gate executed), "
                                   f"bit-exact vs CPU oracle",
                       "parallelism": f"1 match per GPU x {world}", "levels_per_match": int(st["levels"] / max(1, args.steps)),
                       "gate_sharing": "off"},
            "match_ms": elapsed * 1e3 / max(1, args.steps),
            "roofline": {"bound": "hbm", "kernel": "blind_rotate4_kernel", "achieved": br_gbps, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": br_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "launches": int(st["br_launches"]),
                         "avg_launch_ms": st["ms_blind_rotate"] / max(1, st["br_launches"]),
                         "algorithmic_bytes_per_blind_rotate": a_br,
                         "keyswitch_GBps": ks_gbps, "algorithmic_bytes_per_keyswitch": a_ks},
        }
        if world == 1 and args.batched_extra > 0:
            # extra: the same match with the library default, identical pending gates evaluated once
            api.set_tuning("reuse_gates", 1)
            api.reset_stats()
            tg = time.perf_counter()
            rbg = api.CiphertextArray(pp, 3 * bitsize)
            circuits.function_f(rbg, probe, tmpl, bound, bitsize, ks)
            api.flush()
            tg = time.perf_counter() - tg
            sg = api.stats()
            assert (rbg.words() == last.words()).all()          # the same ciphertexts, word for word
            out["match_with_gate_sharing"] = {"match_ms": tg * 1e3, "blind_rotates": int(sg["blind_rotates"]),
                                              "gates_shared": int(sg["reused_gates"])}
        if world == 1 and args.batched_extra > 1:
            # extra, not the headline: one probe against B different templates recorded together
            # (1-to-N identification) fills the narrow levels of the DAG
            B = args.batched_extra
            others = [circuits.EncryptedVector(pp, [(v + 29 * k + 3 * i) % 256 for i, v in enumerate(base)],
                                               bitsize, ks).to_device() for k in range(1, B)]
            api.reset_stats()
            tb = time.perf_counter()
            outs = []
            for t in [tmpl] + others:
                rb = api.CiphertextArray(pp, 3 * bitsize)
                circuits.function_f(rb, probe, t, bound, bitsize, ks)
                outs.append(rb)
            api.flush()
            tb = time.perf_counter() - tb
            sb = api.stats()
            assert int(outs[0].decrypt(ks)[0]) == bit and all(int(o.decrypt(ks)[0]) == 1 for o in outs[1:])
            out["batched_matches"] = {"matches": B, "gates_per_s": sb["blind_rotates"] / tb, "seconds": tb,
                                      "levels": int(sb["levels"]), "gates_shared": int(sb["reused_gates"])}
        if world == 1 and args.batched_extra > 0:
            # extra: BASELINE.json's literal wording, a 128-BIT template under Hamming distance +
            # threshold (peba1_hamming_match; not in the reference)
            import random
            rnd = random.Random(7)
            ta, tb = rnd.getrandbits(128), rnd.getrandbits(128)
            w = circuits.hamming_count_bits(128)
            A = circuits.encrypt_number(pp, ta, 128, ks); A.set_words(A.words())
            Bv = circuits.encrypt_number(pp, tb, 128, ks); Bv.set_words(Bv.words())
            hb = circuits.encrypt_number(pp, 40, w, ks)
            api.reset_stats()
            th = time.perf_counter()
            rbh = api.CiphertextArray(pp, w)
            circuits.hamming_match(rbh, A, Bv, 128, hb, ks)
            api.flush()
            th = time.perf_counter() - th
            sh = api.stats()
            assert int(rbh.decrypt(ks)[0]) == (1 if bin(ta ^ tb).count("1") > 40 else 0)
            out["hamming128_match"] = {"match_ms": th * 1e3, "blind_rotates": int(sh["blind_rotates"]),
                                       "levels": int(sh["levels"]), "gates_per_s": sh["blind_rotates"] / th}
        if world == 1 and args.batched_extra > 0:
            # extra: the same match through the optimised DAG (peba1_function_f_fast; not the
            # reference's gate sequence) -- same match bit, fewer and shallower gates
            api.reset_stats()
            tf = time.perf_counter()
            rbf = api.CiphertextArray(pp, 3 * bitsize)
            circuits.function_f_fast(rbf, probe, tmpl, bound, bitsize, ks)
            api.flush()
            tf = time.perf_counter() - tf
            sf = api.stats()
            assert int(rbf.decrypt(ks)[0]) == bit
            out["optimised_dag_match"] = {"match_ms": tf * 1e3, "blind_rotates": int(sf["blind_rotates"]),
                                          "levels": int(sf["levels"]), "gates_per_s": sf["blind_rotates"] / tf}
        if world == 1 and not args.no_cpu_baseline:
            api.set_deferred(False)
            out["cpu_baseline"] = cpu_baseline(seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ks.close()


if __name__ == "__main__":
    main()
