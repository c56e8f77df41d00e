#!/usr/bin/env python3
"""bench.py -- bootstrapped gates/s of the PEBA1 match path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input.  Three workloads
(`--mode`), all on the reference's own gate sequence (libpeba1-circuits == /root/reference/src/Math.cpp
gate for gate), TFHE default 128-bit parameters (n=630, N=1024, k=1, l=3, Bg=2^7):

  match     ONE encrypted match per GPU: Function_f (squared-Euclidean distance of a 128-slot x 8-bit
            probe against a template, then the threshold comparator; Math.cpp:379-387) = 215,544 blind
            rotations + 215,496 key switches.  BASELINE.json configs[1], the headline at --gpus 1.
  sharded   ONE match whose slots are partitioned over the ranks (north_star's split, BASELINE
            configs[2]): rank r evaluates slots/N of the reference's slot loop (Math.cpp:351-360),
            ONE RCCL gather moves each rank's 24-ciphertext partial sum to rank 0, which runs the
            adder tree and the comparator.  Total work is fixed as N grows: "scaling": "strong".
            With --gpus 1, --logical-ranks N runs the same phases for N logical ranks on the one
            device (256 slots by default, as configs[2] names).
  identify  1-to-N identification (BASELINE configs[3]): every rank matches the probe against
            --matches templates of its own (128 per GPU in configs[3]), recorded --group at a time;
            no data-path collective, "scaling": "weak".

`--mode auto` (default) = match at --gpus 1, sharded at --gpus N > 1, so that the driver's
`bench.py --gpus N` measures the split north_star names: ONE curve, "scaling": "strong" -- its N = 1
point is one rank holding every slot, which is the reference's unsplit Function_f (the line says so);
`weak_scaling` (independent matches per GPU) rides beside it at every N.

Inputs (probe, template, threshold ciphertexts) and the evaluation keys are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.  `value` counts executed blind
rotations (a MUX is two) of ALL ranks per second of the slowest rank.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            starts its own N ranks (a child `torch.distributed.run`; the parent never
                                            touches the GPU) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      the same, launched by hand
Every line carries `roofline.valu_issue` (the roofline that binds, from profiles/isa_mix.json and
profiles/valu_issue_costs.json: `frac` at the kernel's own occupancy, `frac_vs_chip_peak` at the chip's best issue rates; at
N > 1 for the 8-wave kernel that dominates there); the N > 1 line also `dist` (who took part: RCCL version, every rank's PCI
bus id, the collectives run and their issue order, the transport and why).

This file: argument parsing, the timed region, the CPU baseline, the JSON line.  benchkit/: the roofline models and committed
counter summaries (roofline.py), the launcher, the transport negotiation and the evidence (launch.py), the untimed extra
workloads of the single-GPU line (extras.py) -- nothing in there runs inside the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# everything that is not the timed region or the CPU baseline lives in benchkit/ (roofline models and committed counter
# summaries; the N > 1 launcher, the transport negotiation, the evidence of who took part; the untimed extra workloads)
from benchkit.roofline import HBM_PEAK_GBPS, algorithmic_bytes, committed_counters, valu_issue_block   # noqa: E402


def host_cpu_width():
    """Threads this process can really run at once: the CPUs it may be scheduled on, capped by the container's CPU quota
    (cgroup v2 cpu.max) -- on the GPU box 256 CPUs are visible and the quota is 16 (one GPU's share of the host)."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    usable = visible if quota is None else max(1, min(visible, int(quota)))
    return usable, visible, quota


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seed):
    """The CPU path timed beside the GPU, on this box's host cores, on a bounded sample of the same work (independent
    bootsAND gates under P128, one GPU's share of the host).  The reference links an fp64-FFT flavour of libtfhe
    (/root/reference/CMakeLists.txt:9-15), which is absent here, so two things are timed and both are reported flat:

      value             oracle/fft_standin.c (use_ntt = 4): an AVX2 + FMA fp64 FFT of N/2 points with the key image in the
                        evaluation domain -- what upstream's fastest CPU flavour (spqlios-fma) costs.  A STAND-IN for the
                        absent TFHE, approximate like it, NOT the parity oracle (checked here at decrypt level only).
      exact_port_value  the parity oracle itself (exact two-prime NTT in scalar C): what every -m gpu test compares with;
                        3-10x slower than any real TFHE build, so never the comparator of a speed-up claim.
    """
    import numpy as np
    from oracle import pyoracle as O
    usable, visible, quota = host_cpu_width()
    cores = min(16, usable)                                 # the GPU box gives one GPU's share of the host: 16 cores
    oks = O.KeySet(O.params("P128"), seed)
    r = O.Rng(77)
    have_avx2 = bool(O.lib().orc_fft4_available())
    fmode = 4 if have_avx2 else 3
    count = 12 * cores                                      # exact port: ~12 core-seconds
    nf = 64 * cores                                         # stand-in: ~6-15 core-seconds
    bits_a, bits_b = np.arange(nf) & 1, (np.arange(nf) >> 1) & 1
    a, b = oks.encrypt(r, bits_a), oks.encrypt(r, bits_b)
    want = [int(x & y) for x, y in zip(bits_a, bits_b)]

    def timed(n, threads, mode):
        t = time.perf_counter()
        out = oks.gate_batch("AND", a[:n], b[:n], nthreads=threads, use_ntt=mode)
        return out, time.perf_counter() - t

    timed(cores, cores, 2)                                  # warm caches, tables, the key images
    timed(cores, cores, fmode)
    out_exact, dt_exact = timed(count, cores, 2)
    out_fft, dt_fft = timed(nf, cores, fmode)
    assert list(oks.decrypt(out_exact)) == want[:count]
    assert list(oks.decrypt(out_fft)) == want
    _, dt1_exact = timed(2, 1, 2)
    dt1_fft = min(timed(8, 1, fmode)[1], timed(8, 1, fmode)[1])
    wider = None
    if usable > cores:          # a host share wider than 16 CPUs: the same gates on every CPU the process can really use
        outw, dtw = timed(min(nf, 16 * usable), usable, fmode)
        wider = {"threads": usable, "value": len(outw) / dtw, "unit": "bootstrapped gates/s"}
    return {"value": nf / dt_fft, "unit": "bootstrapped gates/s", "cores": cores, "threads": cores,
            "kind": "port (fft-standin: fp64 FFT, AVX2+FMA)" if have_avx2 else "port (fft-standin: fp64 FFT, scalar)",
            "ms_per_gate_single_thread": dt1_fft / 8 * 1e3,
            "exact_port_value": count / dt_exact, "exact_port_ms_per_gate_single_thread": dt1_exact / 2 * 1e3,
            "cpu_model": cpu_model_name(), "host_cpus_visible": visible, "cgroup_cpu_quota": quota, "all_usable_cpus": wider,
            "sample": f"{nf} independent bootsAND (P128) on {cores} threads through oracle/fft_standin.c (use_ntt = {fmode}); "
                      f"the exact oracle port: {count} of the same gates on {cores} threads",
            "note": "value = a cost-faithful STAND-IN for the CPU path the reference links (an fp64-FFT libtfhe, absent from "
                    "/root/reference and this image): folded N/2-point complex FFT in AVX2 + FMA, evaluation-domain key image, "
                    "the structure of upstream's spqlios-fma flavour; approximate arithmetic, checked at decrypt level, NOT the "
                    "parity oracle.  exact_port_value = the parity oracle (exact two-prime NTT, scalar C)."}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["auto", "match", "sharded", "identify"], default="auto")
    ap.add_argument("--slots", type=int, default=0, help="slots per template (default 128; 256 for --mode sharded)")
    ap.add_argument("--logical-ranks", type=int, default=0,
                    help="--mode sharded at --gpus 1: run the phases of this many logical ranks on the one device")
    ap.add_argument("--fast-partial", action="store_true",
                    help="--mode sharded: every rank computes its slots' distance with the depth-optimised circuit "
                         "(PEBA1_DIST_FAST_PARTIAL; NOT the reference's gate sequence) -- the latency form of the sharded match")
    ap.add_argument("--ripple-combine", action="store_true",
                    help="--mode sharded: rank 0 adds the partial sums with the pairwise tree of the reference's ripple "
                         "adders and its bit-serial comparator (the DAG the golden digest pins) instead of the "
                         "carry-save / prefix form (peba1_combine_and_compare_fast)")
    ap.add_argument("--matches", type=int, default=8, help="--mode identify: matches per GPU and step (configs[3]: 128)")
    ap.add_argument("--group", type=int, default=4, help="--mode identify: matches recorded per flush")
    ap.add_argument("--weak-matches", type=int, default=8,
                    help="--gpus N > 1, --mode sharded: after the timed (strong-scaling) steps every rank also runs this many "
                         "independent matches of its own (--group per flush; BASELINE configs[3]) -- timed on its own, reported "
                         "as `weak_scaling` beside the strong curve; 0 = skip")
    ap.add_argument("--reference-combine-leg", type=int, default=1,
                    help="--gpus N > 1, --mode sharded without --ripple-combine: also time ONE match with the reference-order "
                         "combine on rank 0 (SURVEY 8e's ADDN tree + minimum), reported as `reference_order_combine`; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: no HIP events around the launches (the roofline block is then empty): what the events cost")
    ap.add_argument("--extras", type=int, default=1,
                    help="at --gpus 1 --mode match: also run the untimed extra workloads (gate sharing, batched "
                         "matches, 256-slot match, Hamming, optimised DAG); 0 = skip")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (exercises the N>1 code path)")
    ap.add_argument("--transport", choices=["auto", "rccl", "torch"], default="auto",
                    help="--backend nccl: auto (default) = libpeba1-dist's own RCCL communicator with its collectives on the "
                         "library's stream, tried once before the timed steps, else torch's communicator behind the library's "
                         "host transport; rccl / torch force one")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL, one GPU per rank; the default and what the driver runs) or gloo: the exchange "
                         "goes through host buffers and every rank uses GPU `LOCAL_RANK mod device count`, so the N>1 "
                         "logic can be rehearsed with several processes on ONE GPU (tests do; never a measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: THIS process becomes the launcher and nothing else -- it has not
        # loaded libtfhe-hip, torch.cuda or any HIP library and never will (a process that has touched the GPU must not
        # start or become another one on this pool), starts one fresh process per GPU and relays rank 0's line
        from benchkit.launch import self_launch       # (imports nothing that touches the GPU)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         f"(`python bench.py --gpus N` does it by itself)")
    mode = args.mode
    if mode == "auto":
        mode = "match" if world == 1 else "sharded"
    nslots = args.slots or (256 if mode == "sharded" and world == 1 else 128)
    logical = args.logical_ranks or (8 if mode == "sharded" and world == 1 else 0)
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist or mode == "sharded":
        import torch
    xdev = "cuda"                     # where ciphertexts sit for the exchange
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        # one node, rendezvous on the loopback address: RCCL's bootstrap (torch's communicator and libpeba1-dist's own) may
        # use the loopback interface too -- a container without another interface would otherwise find none
        if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if args.backend == "gloo":
            xdev = "cpu"
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from peba1_amd import api, circuits, identify, lib
    from peba1_amd import dist as pd
    from benchkit.extras import extras, weak_scaling_leg
    from benchkit.launch import dist_evidence, make_comm
    L = lib.load()
    L.tfhe_hip_set_device(local_rank)
    seed = 0x5EBA2
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, seed, device=True)           # same key on every rank (replicated evaluation keys)
    bitsize = 8
    # synthetic inputs of SURVEY.md 8(d): probe = genuine sample of template 0.  No probe byte is 0: the
    # reference's subtractor is wrong for a zero subtrahend
    base = [((37 * i + 11) % 255) or 1 for i in range(nslots)]
    probe_vals = [v + 1 for v in base]
    threshold = 256
    L.tfhe_hip_set_encrypt_seed(1000 + rank)
    probe = circuits.EncryptedVector(pp, probe_vals, bitsize, ks).to_device()
    bound = circuits.encrypt_number(pp, threshold, 3 * bitsize, ks)
    bound.set_words(bound.words())

    def plain_bit(tmpl_vals):
        return 1 if sum((a - b) ** 2 for a, b in zip(probe_vals, tmpl_vals)) > threshold else 0

    L.tfhe_hip_set_kernel_timing(0 if args.no_kernel_timing else 1)
    api.set_deferred(True)
    # The headline executes every gate the circuit records: the library's sharing of identical
    # pending gates (tuning "reuse_gates", on by default) is switched off for the timed steps
    # and reported separately, so that `value` counts 215,544 blind rotations per 128-slot match.
    api.set_tuning("reuse_gates", 0)
    api.set_tuning("eliminate_dead", 0)     # likewise: gates whose result nothing can observe are executed too

    checked = None          # what the in-run check of the timed work was
    comm, transport, transport_fallback = None, None, None
    if mode == "match":
        tmpl_vals = base if rank == 0 else identify.synthetic_template(base, rank)
        tmpl = circuits.EncryptedVector(pp, tmpl_vals, bitsize, ks).to_device()
        comm, transport, transport_fallback = make_comm(pd, api, dist, torch, pp, args, xdev, rank, world, local_rank) if use_dist else (None, None, None)
        all_bits = api.CiphertextArray(pp, world) if use_dist and rank == 0 else None

        def step():
            rb = api.CiphertextArray(pp, 3 * bitsize)
            circuits.function_f(rb, probe, tmpl, bound, bitsize, ks)   # records ~350k API calls
            # levelised batched execution, pipelined: the launches are enqueued and the next step's recording and
            # levelling overlap their execution (at most one flush in flight); the timed region ends with api.wait()
            api.flush_async()
            if use_dist:
                # the only exchange: every rank's match-bit ciphertext to rank 0 (libpeba1-dist: RCCL gather enqueued on
                # the library's own stream between the stream-ordered export and import; the host waits only for the
                # ranks' one-word status exchange, which runs on a stream of its own and does not wait for the gates in
                # flight -- the next step's recording overlaps them; no buffer of one step is touched by the next before
                # the stream has passed it)
                pd.gather_samples(comm, all_bits.ptr if rank == 0 else None, rb.ptr, 1, pp.ptr)
            return rb

        def check(last):
            bit = int(last.decrypt(ks)[0])
            assert bit == plain_bit(tmpl_vals), f"rank {rank}: match bit {bit}"
            if use_dist and rank == 0:
                assert (all_bits.words()[0] == last.words()[0]).all(), "gathered match-bit ciphertext differs"
            return "decrypted match bit of the last timed match == plaintext rule (distance > bound)"
        workload = (f"Function_f: {nslots} slots x {bitsize} bit template match per GPU, every recorded gate executed")
        parallelism, scaling = f"1 match per GPU x {world}", "weak"
        if args.mode == "auto":
            # the driver's default curve (`bench.py --gpus N`, mode auto) is the slot-sharded match: ONE match, its slots
            # over N ranks -- strong scaling.  This is its N = 1 point: one rank holds every slot, no exchange, and the
            # reference's unsplit Function_f is what runs.  (The N = 1 point of the OTHER curve -- independent matches per
            # GPU -- is `weak_scaling` in the same line.)
            scaling = "strong"
            parallelism = f"{nslots} slots / 1 GPU (the 1-rank point of the slot-sharded curve: no exchange)"
    elif mode == "sharded":
        tmpl_vals = base
        nranks = world if use_dist else max(1, logical)
        lo, hi = pd.shard_slots(nslots, world, rank) if use_dist else (0, nslots)
        tmpl = circuits.EncryptedVector(pp, tmpl_vals, bitsize, ks).to_device()
        S = [a.ptr for a in probe.slots]
        T = [a.ptr for a in tmpl.slots]
        comm, transport, transport_fallback = make_comm(pd, api, dist, torch, pp, args, xdev, rank, world, local_rank) if use_dist else (None, None, None)
        phase_ms = {"ranks": [], "combine": []}

        def step():
            if use_dist:
                # libpeba1-dist (C++): partial sum of this rank's slots, ONE gather of 24 ciphertexts per rank, rank 0 combines
                res = pd.sharded_match(dist, torch, L, circuits.load(), pp.ptr, ks.cloud, pp.words, S[lo:hi], T[lo:hi],
                                       bound.ptr, bitsize, device=xdev, fast_combine=not args.ripple_combine, comm=comm,
                                       fast_partial=args.fast_partial)
                api.flush()
                return res
            # logical ranks on the one device: the same C phases (peba1_sharded_partial_packed / _combine_packed), timed one by one
            parts, rank_ms = [], []
            for r in range(nranks):
                rlo, rhi = pd.shard_slots(nslots, nranks, r)
                tr = time.perf_counter()
                parts.append(pd.local_partial_packed(ks.cloud, pp.words, S[rlo:rhi], T[rlo:rhi], bitsize, fast=args.fast_partial))
                rank_ms.append((time.perf_counter() - tr) * 1e3)
            tr = time.perf_counter()
            res = pd.combine_packed(L, pp.ptr, ks.cloud, parts, bound.ptr, fast=not args.ripple_combine)
            api.flush()
            phase_ms["ranks"].append(rank_ms)
            phase_ms["combine"].append((time.perf_counter() - tr) * 1e3)
            return res

        def check(last):
            if rank != 0:
                return None
            import ctypes as C
            bit = L.bootsSymDecrypt(C.cast(last, lib.LS), ks.ptr)
            assert bit == plain_bit(tmpl_vals), f"sharded match bit {bit}"
            return "decrypted match bit of the last timed sharded match == plaintext rule (distance > bound)"
        workload = (f"slot-sharded Function_f: ONE {nslots} slots x {bitsize} bit match, slots partitioned over "
                    f"{nranks} {'ranks' if use_dist else 'logical ranks on one device'}, one gather of 24-ciphertext "
                    f"partial sums, {'ripple-adder tree + bit-serial comparator' if args.ripple_combine else 'carry-save compressor + prefix adder + prefix comparator'} on rank 0"
                    + ("; per-rank phase: depth-optimised distance circuit (NOT the reference's gate sequence)" if args.fast_partial else ""))
        parallelism = f"{nslots} slots / {nranks} {('GPUs' if args.backend == 'nccl' else 'processes (gloo rehearsal)') if use_dist else 'logical ranks (1 GPU)'}"
        scaling = "strong"
    else:   # identify
        M = args.matches
        tv = [identify.synthetic_template(base, rank * M + m + 1) for m in range(M)]
        genuine = (M // 2) if rank == 0 else -1
        if genuine >= 0:
            tv[genuine] = base
        templates = [circuits.EncryptedVector(pp, t, bitsize, ks).to_device() for t in tv]

        def step():
            return identify.identify(pp, ks, probe, templates, bound, bitsize, group=args.group)

        def check(last):
            got = [int(b) for b in last.decrypt(ks)]
            assert got == [plain_bit(t) for t in tv], f"rank {rank}: identification bits {got}"
            return f"all {M} decrypted match bits per GPU == plaintext rule; the genuine template is the only 0"
        workload = (f"1-to-N identification: probe against {M} independent {nslots} slots x {bitsize} bit templates per GPU "
                    f"(Function_f each, {args.group} recorded per flush)")
        parallelism, scaling = f"{M} matches per GPU x {world}", "weak"

    def sync():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    last = None
    for _ in range(args.warmup):
        last = step()
    api.wait()
    sync()
    api.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    api.wait()                      # every launch of the K steps has completed on the library's stream
    sync()
    elapsed = time.perf_counter() - t0
    st = api.stats()
    rotations_all = float(st["blind_rotates"])
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rotations_all], dtype=torch.float64, device=xdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rotations_all = float(rr.item())
    checked = check(last)

    # N > 1: the other way the path shards -- independent matches per GPU (1-to-N identification, BASELINE configs[3];
    # /root/reference/src/main.cpp:533-542 once per enrolled client), no data-path collective, only the match bits are
    # gathered.  Untimed by the contract (the timed steps above are the strong-scaling split north_star names), timed on
    # its own like the single-GPU extras, so that ONE line per N carries both curves.
    weak = None
    if use_dist and mode == "sharded" and args.weak_matches > 0:
        weak = weak_scaling_leg(api, circuits, identify, dist, torch, pp, ks, probe, bound, base, bitsize, rank, world,
                                args.weak_matches, args.group, xdev, plain_bit, comm)

    # N > 1, default combine (carry-save / prefix): beside it, ONE more match with rank 0 combining as SURVEY 8(e) writes it
    # (the pairwise tree of the reference's ripple adders ADDN(23) + its bit-serial comparator minimum(24), the DAG the
    # golden digests pin) -- timed on its own, so that the line says what the reference's own combine costs at this N
    ref_combine = None
    if use_dist and mode == "sharded" and not args.ripple_combine and args.reference_combine_leg:
        sync()
        api.reset_stats()
        tr = time.perf_counter()
        res_ref = pd.sharded_match(dist, torch, L, circuits.load(), pp.ptr, ks.cloud, pp.words, S[lo:hi], T[lo:hi], bound.ptr,
                                   bitsize, device=xdev, fast_combine=False, comm=comm, fast_partial=args.fast_partial)
        api.flush()
        api.wait()
        sync()
        tr = time.perf_counter() - tr
        tt = torch.tensor([tr], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rr = torch.tensor([float(api.stats()["blind_rotates"])], dtype=torch.float64, device=xdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        if rank == 0:
            bit = L.bootsSymDecrypt(ctypes.cast(res_ref, lib.LS), ks.ptr)
            assert bit == plain_bit(tmpl_vals), f"sharded match (reference-order combine) bit {bit}"
            L.delete_gate_bootstrapping_ciphertext_array(24, ctypes.cast(res_ref, lib.LS))
            ref_combine = {"match_ms": float(tt.item()) * 1e3, "gates_per_s_all_ranks": float(rr.item()) / float(tt.item()),
                           "blind_rotates_all_ranks": int(rr.item()),
                           "combine": "pairwise tree of the reference's ripple adders (bootsADDNbit, Math.cpp:54-69) + its "
                                      "bit-serial comparator (minimum, Math.cpp:265-292) on rank 0: SURVEY 8(e)'s form, pinned by "
                                      "tests/golden/sharded_match*_digest.json",
                           "checked": "decrypted match bit == plaintext rule"}

    evidence = dist_evidence(dist, L, comm, args, world, rank, local_rank) if use_dist else None
    if evidence is not None:
        evidence["transport"] = transport
        evidence["transport_fallback_reason"] = transport_fallback

    if rank == 0:
        a_br, a_ks, ct = algorithmic_bytes(pp)
        steps = max(1, args.steps)
        value = rotations_all / elapsed
        # two blind-rotate kernels: the 4-wave form, and the 8-wave form for launches of at most one workgroup
        # per CU (the engine's statistics keep them apart).  The roofline block is for whichever took more time
        # in this run -- the 4-wave kernel in every single-GPU match; the 8-wave one where every level is
        # narrow (a few slots per rank) -- and carries the other beside it.
        rot8, ms8, n8 = st["br8_rotations"], st["ms_blind_rotate8"], st["br8_launches"]
        rot4, ms4, n4 = st["blind_rotates"] - rot8, st["ms_blind_rotate"] - ms8, st["br_launches"] - n8
        dom8 = ms8 > ms4
        drot, dms, dn = (rot8, ms8, n8) if dom8 else (rot4, ms4, n4)
        orot, oms, on = (rot4, ms4, n4) if dom8 else (rot8, ms8, n8)
        br_gbps = drot * a_br / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        ks_gbps = st["keyswitches"] * a_ks / (st["ms_keyswitch"] * 1e-3) / 1e9 if st["ms_keyswitch"] else 0.0
        traffic, valu, khash = committed_counters()
        if dom8:
            traffic, valu = None, None        # the committed counter passes profiled blind_rotate4_kernel launches
        out = {
            "metric": "bootstrapped gates/sec (blind rotations/s) over the PEBA1 match path, and end-to-end match ms",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{workload}; TFHE P128 (n={pp.n}, N={pp.N}, k={pp.k}, l={pp.l}, Bg=2^{pp.Bgbit}); "
                                   f"{int(rotations_all / steps)} blind rotations per step over all ranks. "
                                   f"Checked in this run: {checked}. Ciphertext parity with the CPU oracle is "
                                   f"established by the -m gpu tests (every kernel and gate word for word; the whole "
                                   f"128-slot Function_f of this workload, the 256-slot match sharded over 8 ranks, Function_g "
                                   f"and the Hamming match by SHA-256 digest of their ciphertexts against the oracle's), "
                                   f"not re-checked here",
                       "mode": mode, "parallelism": parallelism,
                       "levels_per_step_rank0": int(st["levels"] / steps), "gate_sharing": "off", "dead_gate_elimination": "off"},
            "match_ms": elapsed * 1e3 / steps if mode != "identify" else elapsed * 1e3 / steps / max(1, args.matches),
            # The mandated HBM roofline: ALGORITHMIC bytes (the whole 59 MiB key image per blind
            # rotation, no reuse) per second of blind-rotate launch time, against the 8 TB/s peak.  The kernel
            # is NOT HBM-bound -- the key image is served from L2 / Infinity Cache (`traffic` = measured
            # HBM-side bytes per launch, ~3 % of the algorithmic bytes) -- it is bound by VALU issue:
            # `valu_issue` prices the static instruction mix of a step (profiles/isa_mix.json, from the
            # assembly of the kernels as built) with the measured issue costs (profiles/valu_issue_costs.json)
            # and sets it against the cycles a step takes (`valu`: the SQ counters' view of the same).
            "roofline": {"bound": "valu", "kernel": "blind_rotate8_kernel" if dom8 else "blind_rotate4_kernel", "achieved": br_gbps, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": br_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "achieved_is": "algorithmic bytes / launch time (cache reuse across gates counts as bandwidth)",
                         "valu": valu, "counters_measured_on_kernels_sha16": khash if (traffic or valu) else None,
                         "kernels_sha16": khash,
                         "launches": int(dn), "avg_launch_ms": dms / max(1, dn), "rotations_per_launch": drot / max(1, dn),
                         # shader clock of the timed blind-rotate launches: cycles / 100 MHz reference ticks of every 61st workgroup
                         # of every launch (the same launch is ~15 % slower at the ~2.0 GHz of a cold or power-limited
                         # chip than at ~2.37 GHz) -- explains run-to-run and box-to-box differences
                         "shader_clock_ghz": 0.1 * st["clk_shader_cycles"] / st["clk_ref_ticks"] if st["clk_ref_ticks"] else None,
                         # `value` per GHz of that clock: what compares runs on different boxes (round 5 saw 2.22-2.34 GHz: 103-110k
                         # gates/s for the same code, 46.6-46.8k per GHz; the driver's round-4 line: 44.6k per GHz)
                         "gates_per_s_per_shader_ghz": value / (0.1 * st["clk_shader_cycles"] / st["clk_ref_ticks"]) if st["clk_ref_ticks"] else None,
                         "other_blind_rotate_kernel": {"kernel": "blind_rotate4_kernel" if dom8 else "blind_rotate8_kernel",
                                                       "launches": int(on), "avg_launch_ms": oms / max(1, on),
                                                       "rotations_per_launch": orot / max(1, on)},
                         "all_blind_rotate_algorithmic_GBps": st["blind_rotates"] * a_br / (st["ms_blind_rotate"] * 1e-3) / 1e9 if st["ms_blind_rotate"] else 0.0,
                         "algorithmic_bytes_per_blind_rotate": a_br,
                         "keyswitch_algorithmic_GBps": ks_gbps, "algorithmic_bytes_per_keyswitch": a_ks,
                         "ms_blind_rotate_per_step": st["ms_blind_rotate"] / steps,
                         "ms_keyswitch_per_step": st["ms_keyswitch"] / steps},
        }
        # the binding roofline, for the kernel that dominates THIS run: at N = 1 the single-GPU extras below add the measured
        # side of the 4-wave kernel (4,096 independent gates); where the 8-wave kernel dominates (N > 1: every level of a rank's
        # share is narrow) the measured side is that kernel's launches of the timed steps themselves -- one round each
        clock = 0.1 * st["clk_shader_cycles"] / st["clk_ref_ticks"] if st["clk_ref_ticks"] else None
        out["roofline"]["valu_issue"] = valu_issue_block(None, None, {"ms": ms8, "launches": n8, "shader_clock_ghz": clock} if dom8 else None)
        if mode == "sharded" and world == 1 and phase_ms["combine"]:
            # what the same phases would take with one GPU per rank: the slowest rank's partial sum, then
            # rank 0's combine (the gather of 24 x 2.5 KB per rank is microseconds) -- a projection from
            # timings of logical ranks on ONE device, not a multi-GPU measurement
            k = min(args.steps, len(phase_ms["combine"]))
            ranks = phase_ms["ranks"][-k:]
            out["logical_rank_phases"] = {
                "partial_ms_per_rank": [sum(r[i] for r in ranks) / k for i in range(len(ranks[0]))],
                "combine_ms": sum(phase_ms["combine"][-k:]) / k,
                "projected_match_ms_one_gpu_per_rank": sum(max(r) for r in ranks) / k + sum(phase_ms["combine"][-k:]) / k,
                "note": "projection from logical ranks timed on one device; not measured on several GPUs"}
        if weak is not None:
            out["weak_scaling"] = weak
        if ref_combine is not None:
            out["reference_order_combine"] = ref_combine
        if evidence is not None:
            out["dist"] = evidence
        if world == 1 and mode == "match" and args.extras > 0:
            out.update(extras(api, circuits, identify, lib, pd, pp, ks, probe, tmpl, bound, base, probe_vals, bitsize,
                              plain_bit, last))
            # the N = 1 point of the weak-scaling curve the N > 1 lines carry: independent matches on the one GPU
            out["roofline"]["valu_issue"] = valu_issue_block(out.get("independent_gates_4096"), out.get("independent_gates_sweep"))
            i4 = out["identify_4_matches_one_flush"]
            out["weak_scaling"] = {"gates_per_s_all_ranks": i4["gates_per_s"], "per_gpu": i4["gates_per_s"], "matches_per_gpu": 4,
                                   "group": 4, "seconds": i4["seconds"], "n_gpus": 1,
                                   "note": "independent matches per GPU (1-to-N identification, configs[3]); every recorded gate executed"}
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
