"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the slot-sharded match
(peba1_amd/dist.py) over the plaintext provider -- exercises the slot partition, the
single gather of 24-sample partial sums and the rank-0 combine + comparator."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["PEBA1_ROOT"])
from peba1_amd import dist as pd
t = os.environ["PEBA1_TMP"]
gate = C.CDLL(t + "/libplain_tfhe.so", mode=C.RTLD_GLOBAL)
circ = C.CDLL(t + "/libcircuits_test.so")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
gate.new_default_gate_bootstrapping_parameters.restype = C.c_void_p
gate.new_random_gate_bootstrapping_secret_keyset.restype = C.c_void_p
gate.new_random_gate_bootstrapping_secret_keyset.argtypes = [C.c_void_p]
gate.new_gate_bootstrapping_ciphertext_array.restype = C.c_void_p
gate.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, C.c_void_p]
gate.bootsSymEncrypt.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
gate.bootsSymDecrypt.argtypes = [C.c_void_p, C.c_void_p]
params = gate.new_default_gate_bootstrapping_parameters(128)
key = gate.new_random_gate_bootstrapping_secret_keyset(params)
cloud = key + 24          # &key->cloud: params, lwe_key, tgsw_key pointers precede it
SZ = 24                   # sizeof(LweSample)
def enc(v, bits):
    p = gate.new_gate_bootstrapping_ciphertext_array(bits, params)
    for i in range(bits):
        gate.bootsSymEncrypt(p + i * SZ, (v >> i) & 1, key)
    return p
nslots = int(os.environ["PEBA1_SLOTS"])
tmpl = [(37 * i + 11) % 255 for i in range(nslots)]
probe = [(91 * i + 5) % 256 for i in range(nslots)] if os.environ["PEBA1_CASE"] == "impostor" else [v + 1 for v in tmpl]
lo, hi = pd.shard_slots(nslots, world, rank)
assert all(pd.shard_slots(n, w, r) == pd.shard_slots_c(n, w, r) for n in (1, 7, 12, 128) for w in (1, 2, 3, 8) for r in range(w))
S = [enc(probe[i], 8) for i in range(lo, hi)]
T = [enc(tmpl[i], 8) for i in range(lo, hi)]
bound = enc(int(os.environ["PEBA1_BOUND"]), 24)
fast = os.environ.get("PEBA1_FAST") == "1"
inject = int(os.environ.get("PEBA1_INJECT", "-1"))
if inject >= 0:
    # failure containment: rank `inject` fails locally; every rank must come back from the collective call -- the
    # failed one and rank 0 with an error that says what happened, the others clean -- and nobody hangs in the gather
    comm = pd.Comm(dist, torch, "cpu")
    if rank == inject:
        comm.inject_failure(1)
    try:
        pd.sharded_match(dist, torch, gate, circ, params, cloud, 2, S, T, bound, 8, device="cpu", comm=comm)
        outcome = "clean"
    except RuntimeError as e:
        outcome = str(e)
    want = "injected failure" if rank == inject else (f"rank {inject} reported a failure" if rank == 0 else "clean")
    assert want in outcome, (rank, outcome)
    # the communicator stays usable: the next match goes through
    res = pd.sharded_match(dist, torch, gate, circ, params, cloud, 2, S, T, bound, 8, device="cpu", comm=comm)
    comm.close()
    print("CONTAINED", rank, flush=True)
else:
    res = pd.sharded_match(dist, torch, gate, circ, params, cloud, 2, S, T, bound, 8, device="cpu", fast_combine=fast, fast_partial=fast)
# the ONE encrypted probe of an identification reaches every rank: peba1_dist_broadcast_samples over the host transport
bc = pd.Comm(dist, torch, "cpu")
pb = enc(0xA5 if rank == 1 % world else 0, 8)
pd.broadcast_samples(bc, pb, 8, params, root=1 % world)
assert sum(gate.bootsSymDecrypt(pb + i * SZ, key) << i for i in range(8)) == 0xA5, rank
bc.close()
if rank == 0:
    bit = gate.bootsSymDecrypt(res, key)
    d = sum((a - b) ** 2 for a, b in zip(probe, tmpl))
    assert bit == (1 if d > int(os.environ["PEBA1_BOUND"]) else 0), (bit, d)
    print("OK", bit, d)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    t = str(tmp_path_factory.mktemp("dist"))
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["g++", "-O1", "-std=gnu++11", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(ROOT, "tests/mock/plain_tfhe.cpp"), "-o", t + "/libplain_tfhe.so"])
    subprocess.check_call(["g++", "-O1", "-std=gnu++17", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(ROOT, "peba1_amd/csrc/circuits.cpp"), os.path.join(ROOT, "peba1_amd/csrc/circuits_fast.cpp"), "-o", t + "/libcircuits_test.so"])
    with open(t + "/worker.py", "w") as f:
        f.write(WORKER)
    return t


def test_shard_slots_partition():
    from peba1_amd import dist as pd
    for nslots in (1, 7, 128, 256):
        for world in (1, 2, 3, 8):
            spans = [pd.shard_slots(nslots, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nslots
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,case,bound,fast", [(2, "genuine", 256, 0), (2, "impostor", 256, 0), (3, "genuine", 5, 0),
                                                   (3, "impostor", 256, 1)])
def test_sharded_match_gloo(built, world, case, bound, fast):
    env = dict(os.environ, PEBA1_ROOT=ROOT, PEBA1_TMP=built, PEBA1_SLOTS="12", PEBA1_CASE=case,
               PEBA1_BOUND=str(bound), PEBA1_FAST=str(fast), MASTER_ADDR="127.0.0.1")
    port = 29600 + world * 7 + (1 if case == "impostor" else 0) + 3 * fast
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), built + "/worker.py"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


@pytest.mark.parametrize("world,inject", [(2, 1), (3, 2), (3, 0)])
def test_a_local_failure_is_contained_around_the_gather(built, world, inject):
    """VERDICT r3 item 3 / ADVICE r3: a rank that fails before the exchange still enters it (status word in front of its
    payload), so the others are not left inside the collective: the failed rank and rank 0 get -1 with a message naming
    what happened, the rest return clean, and the same communicator then runs a match normally."""
    env = dict(os.environ, PEBA1_ROOT=ROOT, PEBA1_TMP=built, PEBA1_SLOTS="7", PEBA1_CASE="impostor",
               PEBA1_BOUND="256", PEBA1_FAST="0", PEBA1_INJECT=str(inject), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29650 + 3 * world + inject), built + "/worker.py"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("CONTAINED") == world and "OK" in out.stdout


LOGICAL_WORKER = r'''
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.environ["PEBA1_ROOT"])
from peba1_amd import dist as pd
t = os.environ["PEBA1_TMP"]
gate = C.CDLL(t + "/libplain_tfhe.so", mode=C.RTLD_GLOBAL)
circ = C.CDLL(t + "/libcircuits_test.so")
V = C.c_void_p
gate.new_default_gate_bootstrapping_parameters.restype = V
gate.new_random_gate_bootstrapping_secret_keyset.restype = V
gate.new_random_gate_bootstrapping_secret_keyset.argtypes = [V]
gate.new_gate_bootstrapping_ciphertext_array.restype = V
gate.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, V]
gate.bootsSymEncrypt.argtypes = [V, C.c_int32, V]
gate.bootsSymDecrypt.argtypes = [V, V]
params = gate.new_default_gate_bootstrapping_parameters(128)
key = gate.new_random_gate_bootstrapping_secret_keyset(params)
cloud = key + 24
def enc(v, bits):
    p = gate.new_gate_bootstrapping_ciphertext_array(bits, params)
    for i in range(bits):
        gate.bootsSymEncrypt(p + i * 24, (v >> i) & 1, key)
    return p
nslots = 9
tmpl = [(37 * i + 11) % 255 for i in range(nslots)]
probe = [(91 * i + 5) % 256 for i in range(nslots)]
d = sum((a - b) ** 2 for a, b in zip(probe, tmpl))
S, T = [enc(v, 8) for v in probe], [enc(v, 8) for v in tmpl]
for world in (1, 2, 3, 8, 12):            # 12 > slots: some ranks hold no slot
    seen = []
    for bound in (d - 1, d):
        res = pd.sharded_match_logical(torch, gate, circ, params, cloud, 2, S, T, enc(bound, 24), 8, world, device="cpu",
                                       partial_hook=lambda r, t: seen.append(r))
        assert gate.bootsSymDecrypt(res, key) == (1 if d > bound else 0), (world, bound)
        fast = pd.sharded_match_logical(torch, gate, circ, params, cloud, 2, S, T, enc(bound, 24), 8, world, device="cpu",
                                        fast_combine=True)
        assert gate.bootsSymDecrypt(fast, key) == (1 if d > bound else 0), ("fast combine", world, bound)
        both = pd.sharded_match_logical(torch, gate, circ, params, cloud, 2, S, T, enc(bound, 24), 8, world, device="cpu",
                                        fast_combine=True, fast_partial=True)       # the latency form: both phases depth-optimised
        assert gate.bootsSymDecrypt(both, key) == (1 if d > bound else 0), ("fast partial + combine", world, bound)
        mixed = pd.sharded_match_logical(torch, gate, circ, params, cloud, 2, S, T, enc(bound, 24), 8, world, device="cpu",
                                         fast_partial=True)                           # fast partial sums into the ripple combine
        assert gate.bootsSymDecrypt(mixed, key) == (1 if d > bound else 0), ("fast partial, ripple combine", world, bound)
    assert seen == list(range(world)) * 2
print("LOGICAL-OK")
'''


def test_sharded_match_logical_ranks_plain_provider(built):
    """sharded_match_logical (N logical ranks in one process: what the one-GPU tests and
    bench.py --mode sharded --logical-ranks N run) over the plaintext provider: same slot
    partition, same packed exchange buffers, same combine as the gloo runs above.  In its own
    process: the mock exports the tfhe API with RTLD_GLOBAL."""
    with open(built + "/logical.py", "w") as f:
        f.write(LOGICAL_WORKER)
    out = subprocess.run([sys.executable, built + "/logical.py"], env=dict(os.environ, PEBA1_ROOT=ROOT, PEBA1_TMP=built),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LOGICAL-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("world", [1, 3, 8])
def test_cpp_host_runs_the_sharded_match_without_python(built, world):
    """the multi-GPU path belongs to the C++ host.  A plain C++ program (tests/refcompat/dist_host.cpp)
    links libpeba1-dist's sources, the circuits and the plaintext provider, and runs peba1_sharded_function_f for
    `world` ranks through the host transport: match bit on both sides of the threshold, both combine forms."""
    inc = os.path.join(ROOT, "include")
    exe = os.path.join(built, "dist_host")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-std=gnu++17", "-D__HIP_PLATFORM_AMD__", "-I" + inc, "-I/opt/rocm/include",
                               os.path.join(ROOT, "tests/refcompat/dist_host.cpp"), os.path.join(ROOT, "tests/mock/plain_tfhe.cpp"),
                               os.path.join(ROOT, "peba1_amd/csrc/circuits.cpp"), os.path.join(ROOT, "peba1_amd/csrc/circuits_fast.cpp"),
                               os.path.join(ROOT, "peba1_amd/csrc/dist.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-ldl",
                               "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "DIST-HOST-OK" in out.stdout, out.stdout + out.stderr


def test_libpeba1_dist_exports_every_declared_symbol():
    import re
    txt = open(os.path.join(ROOT, "include", "peba1_dist.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = set(re.findall(r"\b(peba1_\w+)\s*\(", txt)) - {"peba1_gather_fn"}
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "peba1_amd", "libpeba1-dist.so")]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert len(declared) >= 12 and not (declared - exported), declared - exported


def test_bench_launcher_parent_reports_failed_ranks_and_never_loads_the_gpu_library(built):
    """`python bench.py --gpus 2` with no launcher starts its own ranks.  Here (no GPU) both ranks end in
    libtfhe-hip's "no HIP device" abort: the parent must come back non-zero with no JSON line, and must itself never have
    loaded the HIP library (a process that has touched the GPU may not start another one on this pool)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--slots", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "no HIP device" in out.stderr
    # the parent's own modules: import bench and run the launcher's argument handling in-process, then look at what is loaded
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2']; sys.path.insert(0, %r)\n"
            "import bench, subprocess\n"
            "from benchkit.launch import self_launch\n"
            "class P:\n"
            "    stdout = iter(())\n"
            "    pid = 0\n"
            "    def wait(self): return 7\n"
            "    def poll(self): return 7\n"
            "subprocess.Popen = lambda *a, **k: P()\n"
            "rc = self_launch(2)\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert rc == 7, rc\n"
            "assert 'libtfhe-hip' not in maps and 'libamdhip64' not in maps and 'torch' not in sys.modules, 'the launcher touched the GPU stack'\n"
            "print('PARENT-CLEAN')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0 and "PARENT-CLEAN" in out.stdout, out.stdout + out.stderr
