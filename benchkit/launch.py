"""benchkit.launch -- how `bench.py --gpus N` becomes N ranks, which transport their one exchange takes, and the evidence of
who took part.  Runs before / after the timed region, never inside it."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def self_launch(n):
    """The parent of `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
    (subprocess; no exec, and this process never initialises the GPU) in a session of its own, rendezvous on the loopback
    address at a port the launcher picks (`--standalone --local-addr 127.0.0.1`), relays the children's output (rank 0's JSON
    line last, on stdout) and returns the launcher's exit code -- non-zero if any rank failed.  If THIS process is told to stop (SIGTERM, SIGINT: a driver's time limit),
    the launcher and every rank are stopped with it -- SIGTERM to the process group, SIGKILL after a grace period -- so that
    no rank stays behind holding a GPU of a shared box."""
    import signal
    import subprocess
    import time
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher picks a free rendezvous port itself and hands MASTER_ADDR / MASTER_PORT to the ranks (no
    # probe-then-reuse race); --local-addr 127.0.0.1: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", BENCH] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)

    def stop_ranks(grace=10.0):
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except (ProcessLookupError, PermissionError):
                return
            t_end = time.time() + grace
            while time.time() < t_end:
                if proc.poll() is not None:
                    break
                time.sleep(0.1)

    stopped = []

    def on_signal(signum, _frame):
        stopped.append(signum)
        raise KeyboardInterrupt

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    line_json, rc = None, 1
    try:
        for line in proc.stdout:
            if line.startswith("{") and line.rstrip().endswith("}"):
                line_json = line
            else:
                sys.stderr.write(line)                              # launcher chatter, other ranks' prints
        rc = proc.wait()
    except KeyboardInterrupt:
        sys.stderr.write(f"bench.py: stopped by signal {stopped[-1] if stopped else 'SIGINT'}; stopping the {n} ranks\n")
        rc = 130
    finally:
        if proc.poll() is None:
            stop_ranks()
        for sig, h in old.items():
            signal.signal(sig, h)
    if line_json is not None:
        sys.stdout.write(line_json)
        sys.stdout.flush()
    if rc == 0 and line_json is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no JSON line\n")
        rc = 1
    return rc


TRIAL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
rank, world, port, local_rank = %(rank)d, %(world)d, %(port)d, %(local_rank)d
if os.environ.get("PEBA1_BENCH_TRIAL_HANG"):        # test hook: a trial that never comes back
    import time
    time.sleep(3600)
import torch
import torch.distributed as dist
from peba1_amd import api, lib
from peba1_amd import dist as pd
L = lib.load()
L.tfhe_hip_set_device(local_rank)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%d" %% port, rank=rank, world_size=world)
pp = api.ParameterSet(128)
comm = pd.Comm(dist, torch, "cuda")
comm.set_timeout(%(deadline)d)
mine = api.CiphertextArray(pp, 1)
everyone = api.CiphertextArray(pp, world) if rank == 0 else None
pd.gather_samples(comm, everyone.ptr if rank == 0 else None, mine.ptr, 1, pp.ptr)
api.wait()
if rank == 0:
    assert (everyone.words() == mine.words()[0]).all(), "the trial gather moved the wrong words"
print("TRIAL-OK", comm.counters()["status_channel"], flush=True)
os._exit(0)                                          # no destructors: nothing here is worth a clean RCCL teardown
"""


def trial_in_child(rank, world, local_rank, port, budget_s=None):
    """ONE one-sample gather through libpeba1-dist's own RCCL communicator, in a fresh CHILD process per rank (its own gloo
    rendezvous on `port`, its own communicator): whatever happens in there -- RCCL not loadable, ncclCommInitRank failing
    on some rank, a gather that never completes -- ends with that child, and this process has not touched the communicator
    (a hang inside the bench process itself can only end the job).  Returns None if the child reported success,
    else why not.  The child is killed after `budget_s` seconds."""
    import subprocess
    if budget_s is None:
        budget_s = int(os.environ.get("PEBA1_BENCH_TRIAL_BUDGET_S", "150"))
    code = TRIAL_CHILD % {"root": ROOT, "rank": rank, "world": world, "port": port, "local_rank": local_rank,
                          "deadline": max(10, budget_s - 30)}
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=budget_s)
    except subprocess.TimeoutExpired:
        return f"rank {rank}: the trial child did not finish within {budget_s} s (killed)"
    if out.returncode == 0 and "TRIAL-OK" in out.stdout:
        return None
    tail = (out.stderr or out.stdout).strip().splitlines()[-1:] or ["no output"]
    return f"rank {rank}: the trial child exited with code {out.returncode}: {tail[0][:300]}"


def make_comm(pd, api, dist, torch, pp, args, xdev, rank, world, local_rank=0):
    """The communicator of the exchange, agreed on by every rank.  --transport auto (default): libpeba1-dist's own RCCL
    communicator, the collectives enqueued on the library's stream ("cuda").  Before THIS process makes it, the same thing is
    tried in a child process per rank (trial_in_child: communicator + ONE one-sample gather), and the ranks agree on the
    outcome through the torch group.  What that covers: RCCL not loadable or a symbol missing on any rank, no unique id,
    ncclCommInitRank failing or hanging on any rank, a trial gather that fails, moves the wrong words or hangs -- in every
    one of these every rank falls back to the host transport of the same C library carried by torch's own RCCL communicator
    ("torch-cuda": device tensors through torch.distributed; slower per exchange -- a host wait, two copies -- the same
    ciphertexts) and the line says so (`dist.transport`, `dist.transport_fallback_reason`).  What it does not cover: a
    communicator that passes the trial and hangs later -- that ends the job through the library's deadline (exit code 86,
    PEBA1_DIST_TIMEOUT_S) with the waiting rank named, and torch's own communicator failing too (no transport left:
    SystemExit).  gloo rehearsals use the host transport on host tensors ("cpu")."""
    if xdev == "cpu" and args.transport != "torch":
        return pd.Comm(dist, torch, "cpu"), "host callbacks over torch.distributed (gloo)", None

    def agree(why):
        """every rank's verdict -> (all fine, the reasons of those that were not)"""
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        return all(r is None for r in reasons), "; ".join(r for r in reasons if r)

    def attempt(kind):
        comm, why = None, None
        try:
            comm = pd.Comm(dist, torch, kind)       # "cuda": the ranks agree inside, before ncclCommInitRank (peba1_amd/dist.py)
            comm.set_timeout(float(os.environ.get("PEBA1_DIST_TIMEOUT_S", "600")))
        except Exception as e:                                  # noqa: BLE001 -- any failure means "not this transport"
            why = f"rank {rank}: {type(e).__name__}: {e}"
        fine, reasons = agree(why)
        if fine:
            return comm, None
        if comm is not None:
            comm.abandon()                          # not destroyed: it may be half made on the ranks that did succeed
        return None, reasons or "a rank reported failure"

    if args.transport in ("auto", "rccl"):
        port = int(os.environ.get("MASTER_PORT", "29577")) + 1 + (os.getpid() % 7)
        ports = [None] * world
        dist.all_gather_object(ports, port)
        fine, why = agree(trial_in_child(rank, world, local_rank, ports[0]))
        comm = None
        if fine:
            comm, why = attempt("cuda")
        if comm is not None:
            return comm, "rccl: libpeba1-dist's own communicator, collectives on the library's stream", None
        if args.transport == "rccl":
            raise SystemExit(f"--transport rccl: {why}")
        if rank == 0:
            print(f"bench.py: libpeba1-dist's own RCCL communicator is not usable here ({why}); falling back to torch's", file=sys.stderr)
        fallback_reason = why
    else:
        fallback_reason = "--transport torch"
    comm, why = attempt("torch-cuda")
    if comm is None:
        raise SystemExit(f"no usable transport: {why}")
    return comm, "torch.distributed device tensors (torch's RCCL communicator) behind libpeba1-dist's host transport", fallback_reason


def dist_evidence(dist, L, comm, args, world, rank, local_rank):
    """Who took part: every rank's PCI bus id as libtfhe-hip reports it for the device it runs on (all-gathered), the RCCL
    version libpeba1-dist opened, what the communicator has done.  N ranks on N distinct bus ids = N GPUs."""
    import socket
    buf = ctypes.create_string_buffer(64)
    L.tfhe_hip_device_pci_bus_id(buf, 64)
    mine = {"rank": rank, "local_rank": local_rank, "device": int(L.tfhe_hip_get_device()), "pci_bus_id": buf.value.decode(),
            "host": socket.gethostname(), "pid": os.getpid(),
            "collectives": comm.counters() if comm is not None else None}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank != 0:
        return None
    from peba1_amd import dist as pd
    ids = [e["pci_bus_id"] for e in everyone]
    seqs = [(e["collectives"] or {}).get("issue_order_hash") for e in everyone]
    c0 = everyone[0]["collectives"] or {}
    return {"backend": "rccl" if args.backend == "nccl" else "gloo (host-memory rehearsal on shared GPUs; not a measurement)",
            "torch_backend": dist.get_backend(), "world": world,
            "rccl_version": pd.load().peba1_dist_rccl_version() if args.backend == "nccl" else None,
            "devices": ids, "distinct_devices": len(set(ids)), "one_gpu_per_rank": len(set(ids)) == world,
            "status_word_collectives": c0.get("status_word_exchanges"), "data_collectives": {k: c0.get(k) for k in ("gathers", "broadcasts")},
            "library_transport": c0.get("transport"), "status_channel": c0.get("status_channel"),
            # every rank issued the same collectives in the same order (peba1_dist_sequence: count + rolling hash)
            "collectives_issued_rank0": c0.get("collectives_issued"), "same_issue_order_on_every_rank": len(set(seqs)) == 1,
            "ranks": everyone}


