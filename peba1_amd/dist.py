"""Multi-GPU sharding of the match path: a thin Python caller of libpeba1-dist (include/peba1_dist.h).

The sharding itself -- slot partition, the one exchange per match, rank 0's combine -- is host C++
(peba1_amd/csrc/dist.cpp), so that a C++ PEBA1 server needs no Python (north_star: "host code stays C++").
This module only picks the transport and passes pointers:

  * RCCL over xGMI (`device="cuda"` under a torch.distributed "nccl" group, one process per GPU): the library
    builds its own RCCL communicator from a unique id broadcast through the torch group, and runs export ->
    ncclGather -> import on libtfhe-hip's own stream, with no host synchronisation in between;
  * a host-memory exchange (`device="cpu"`: torch.distributed over gloo, passed to the library as its gather
    callback) -- how several processes rehearse the N > 1 path on ONE GPU, and what the CPU tests run over the
    plaintext provider.

Two ways the path shards:
  * one match sharded by slots (BASELINE configs[2]): rank r computes the partial sum of squares of its slots
    (the reference's slot loop, Math.cpp:351-360), ONE gather moves 24 ciphertexts per rank to rank 0, rank 0
    adds the partial sums and runs the comparator (Math.cpp:384).  A tree of partial sums is a different gate DAG
    from the reference's left-to-right ripple: the decrypted distance and match bit are identical, intermediate
    ciphertexts are not -- they are pinned against the oracle evaluating the same DAG
    (tests/golden/sharded_match_digest.json);
  * independent matches (1-to-N identification, configs[3]): no data-path collective; `gather_samples` brings
    the match-bit ciphertexts to rank 0.

`sharded_match_logical` runs the same C phases for N logical ranks in one process (the only way to run configs[2]
at size where one GPU is available).  The gate provider is whatever is loaded RTLD_GLOBAL: libtfhe-hip, or the
tests' plaintext provider.
"""
import ctypes as C
import os

import numpy as np

PARTIAL_BITS = 24
FAST_COMBINE = 1
FAST_PARTIAL = 2          # include/peba1_dist.h PEBA1_DIST_FAST_PARTIAL
IDENTIFY_FAST = 1         # PEBA1_IDENTIFY_FAST

_HERE = os.path.dirname(os.path.abspath(__file__))
DIST_PATH = os.path.join(_HERE, "libpeba1-dist.so")
_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
_BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
_dlib = None


def load():
    """libpeba1-dist.so; the gate provider (and, for RCCL, libtfhe-hip) must already be loaded RTLD_GLOBAL."""
    global _dlib
    if _dlib is None:
        if not os.path.exists(DIST_PATH):
            raise RuntimeError(f"{DIST_PATH} is missing: run __graft_entry__.build()")
        D = C.CDLL(DIST_PATH)
        V, I = C.c_void_p, C.c_int
        D.peba1_dist_shard_slots.restype = None
        D.peba1_dist_shard_slots.argtypes = [I, I, I, C.POINTER(I), C.POINTER(I)]
        D.peba1_dist_unique_id.argtypes = [V]
        D.peba1_dist_init_rccl.restype = V
        D.peba1_dist_init_rccl.argtypes = [V, I, I]
        D.peba1_dist_adopt_rccl.restype = V
        D.peba1_dist_adopt_rccl.argtypes = [V, I, I]
        D.peba1_dist_init_host.restype = V
        D.peba1_dist_init_host.argtypes = [_GATHER_FN, V, I, I]
        D.peba1_dist_destroy.restype = None
        D.peba1_dist_destroy.argtypes = [V]
        D.peba1_dist_rank.argtypes = [V]
        D.peba1_dist_world.argtypes = [V]
        D.peba1_dist_last_error.restype = C.c_char_p
        D.peba1_sharded_function_f.argtypes = [V, V, V, V, I, V, I, V, I]
        D.peba1_sharded_partial_packed.argtypes = [V, V, I, I, V, V, I]
        D.peba1_sharded_combine_packed.argtypes = [V, V, I, V, V, I]
        D.peba1_dist_gather_samples.argtypes = [V, V, V, I, V]
        D.peba1_identify.argtypes = [V, V, V, V, V, I, I, V, I, V, I, I]
        D.peba1_dist_set_host_bcast.restype = None
        D.peba1_dist_set_host_bcast.argtypes = [V, _BCAST_FN]
        D.peba1_dist_broadcast_samples.argtypes = [V, V, I, V, I]
        D.peba1_dist_set_timeout.argtypes = [V, C.c_double]
        D.peba1_dist_rccl_version.argtypes = []
        D.peba1_dist_transport.argtypes = [V]
        D.peba1_dist_counters.restype = None
        D.peba1_dist_counters.argtypes = [V, C.POINTER(C.c_uint64)]
        D.peba1_dist_status_channel.argtypes = [V]
        D.peba1_dist_sequence.restype = None
        D.peba1_dist_sequence.argtypes = [V, C.POINTER(C.c_uint64)]
        D.peba1_dist_inject_failure.restype = None
        D.peba1_dist_inject_failure.argtypes = [V, I]
        _dlib = D
    return _dlib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().peba1_dist_last_error().decode()}")


def shard_slots(nslots, world, rank):
    """Contiguous slot range [lo, hi) of `rank`; earlier ranks take the remainder.  The same rule as
    peba1_dist_shard_slots (the tests compare them); in Python so that planning a partition needs no library."""
    base, rem = divmod(nslots, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_slots_c(nslots, world, rank):
    lo, hi = C.c_int(), C.c_int()
    load().peba1_dist_shard_slots(nslots, world, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


_private = {}


def _new_array(gate_lib, count, params_ptr):
    """new_gate_bootstrapping_ciphertext_array through a private handle of the provider: the prototypes set here
    must not disturb the caller's own bindings of the same library."""
    g = _private.get(gate_lib._name)
    if g is None:
        g = C.CDLL(gate_lib._name)
        g.new_gate_bootstrapping_ciphertext_array.restype = C.c_void_p
        g.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, C.c_void_p]
        _private[gate_lib._name] = g
    return g.new_gate_bootstrapping_ciphertext_array(count, C.cast(params_ptr, C.c_void_p))


def _ptr_array(ptrs):
    arr = (C.c_void_p * max(1, len(ptrs)))()
    for i, p in enumerate(ptrs):
        arr[i] = C.cast(p, C.c_void_p)
    return arr


class Comm:
    """A Peba1Comm over a torch.distributed group: RCCL (device="cuda", nccl backend) or the host transport
    (device="cpu": the group's gather, through host tensors, is the library's callback)."""

    def __init__(self, dist, torch, device="cuda"):
        """device: "cuda" = libpeba1-dist's own RCCL communicator (one GPU per rank, a torch.distributed group of any backend
        carries the 128-byte unique id); "cpu" = host transport, the group's collectives on host tensors (gloo);
        "torch-cuda" = host transport, the group's collectives on DEVICE tensors (a "nccl" group cannot move host tensors):
        the exchange then runs over torch's own RCCL communicator -- the fallback bench.py takes when the library's own
        communicator cannot be made."""
        D = load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._keep = None
        self.transport = device
        if device == "cuda":
            # Agree BEFORE ncclCommInitRank: no deadline covers that call, so a rank that cannot take part (RCCL
            # does not load there, a symbol is missing, no unique id) must be known to every rank before any rank enters it.
            # Every rank probes locally -- peba1_dist_unique_id opens RCCL, resolves every symbol the transport uses and
            # makes an id, all without talking to anyone -- and the torch group (any backend) collects the verdicts.
            buf = C.create_string_buffer(128)
            ready = D.peba1_dist_unique_id(buf) == 0
            mine = (ready, buf.raw if ready and self.rank == 0 else None, None if ready else D.peba1_dist_last_error().decode())
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, mine)
            bad = [f"rank {r}: {v[2]}" for r, v in enumerate(verdicts) if not v[0]]
            if bad:                                     # the same exception on every rank, and nobody is inside RCCL
                raise RuntimeError("cannot create the communicator: " + "; ".join(bad))
            self.ptr = D.peba1_dist_init_rccl(verdicts[0][1], self.world, self.rank)
        else:
            stage = (lambda t: t.cuda()) if device == "torch-cuda" else (lambda t: t)

            def gather(_ctx, send, recv, nbytes, root):
                try:
                    mine = stage(torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy()))
                    if device == "torch-cuda":          # all_gather: the one collective every backend has on device tensors
                        out = [torch.empty_like(mine) for _ in range(self.world)]
                        dist.all_gather(out, mine)
                    else:
                        out = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == root else None
                        dist.gather(mine, out, dst=root)
                    if self.rank == root:
                        whole = torch.cat(out).contiguous().cpu().numpy()    # named: must outlive the copy below
                        C.memmove(recv, whole.ctypes.data, nbytes * self.world)
                    return 0
                except Exception:      # never unwind through the C frame
                    return -1
            def bcast(_ctx, buf, nbytes, root):
                try:
                    view = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
                    t = stage(torch.from_numpy(view.copy()))
                    dist.broadcast(t, src=root)
                    if self.rank != root:
                        whole = t.contiguous().cpu().numpy()
                        C.memmove(buf, whole.ctypes.data, nbytes)
                    return 0
                except Exception:      # never unwind through the C frame
                    return -1
            self._keep = _GATHER_FN(gather)
            self._keep_b = _BCAST_FN(bcast)
            self.ptr = D.peba1_dist_init_host(self._keep, None, self.world, self.rank)
            if self.ptr:
                D.peba1_dist_set_host_bcast(self.ptr, self._keep_b)
        if not self.ptr:
            raise RuntimeError("cannot create the communicator: " + D.peba1_dist_last_error().decode())

    def abandon(self):
        """Forget the communicator WITHOUT destroying it: after a failed trial it may hold a collective that never completes,
        and ncclCommDestroy / the stream wait in front of it would then hang too.  Leaks it on purpose."""
        self.ptr = None

    def close(self):
        if self.ptr:
            load().peba1_dist_destroy(self.ptr)
            self.ptr = None

    def counters(self):
        """What this communicator has done so far (peba1_dist_counters)."""
        out = (C.c_uint64 * 4)()
        load().peba1_dist_counters(self.ptr, out)
        seq = (C.c_uint64 * 2)()
        load().peba1_dist_sequence(self.ptr, seq)
        channel = {0: "in front of the payload (host transport)", 1: "data communicator, provider's stream",
                   2: "own communicator (ncclCommSplit), own stream"}[load().peba1_dist_status_channel(self.ptr)]
        return {"status_word_exchanges": int(out[0]), "gathers": int(out[1]), "broadcasts": int(out[2]),
                "payload_bytes_sent": int(out[3]), "transport": "rccl" if load().peba1_dist_transport(self.ptr) else "host",
                "status_channel": channel, "collectives_issued": int(seq[0]), "issue_order_hash": f"{int(seq[1]):016x}"}

    def set_timeout(self, seconds):
        """Bound of every host wait behind a collective (peba1_dist_set_timeout; default PEBA1_DIST_TIMEOUT_S or 600 s)."""
        _check(load().peba1_dist_set_timeout(self.ptr, float(seconds)), "peba1_dist_set_timeout")

    def inject_failure(self, count=1):
        """Test hook: this rank's next `count` collectives report a local failure (peba1_dist_inject_failure)."""
        load().peba1_dist_inject_failure(self.ptr, int(count))


def _flags(fast_combine, fast_partial):
    return (FAST_COMBINE if fast_combine else 0) | (FAST_PARTIAL if fast_partial else 0)


def sharded_match(dist, torch, gate_lib, circ_lib, params_ptr, cloud_ptr, words, sample_slots, template_slots,
                  bound_ptr, bitsize, device="cuda", fast_combine=False, partial_hook=None, comm=None, fast_partial=False):
    """Slot-sharded Function_f across the ranks of `dist` (peba1_sharded_function_f).  `sample_slots` /
    `template_slots`: this rank's slot arrays (LweSample* each, `bitsize` samples).  Returns the 24-sample result
    array pointer on rank 0 (caller frees it with delete_gate_bootstrapping_ciphertext_array(24, p)), None
    elsewhere.  `partial_hook(rank, tensor)` sees this rank's packed partial sums (an extra evaluation-free export:
    tests hash them).  `comm`: a Comm to reuse (one is made and closed otherwise)."""
    D = load()
    own = comm is None
    if own:
        comm = Comm(dist, torch, device)
    try:
        if partial_hook is not None:
            packed = np.zeros(PARTIAL_BITS * words, dtype=np.int32)
            _check(D.peba1_sharded_partial_packed(_ptr_array(sample_slots), _ptr_array(template_slots), len(sample_slots),
                                                  bitsize, cloud_ptr, packed.ctypes.data_as(C.c_void_p),
                                                  _flags(False, fast_partial)), "partial sum")
            partial_hook(comm.rank, torch.from_numpy(packed))
        result_b = _new_array(gate_lib, PARTIAL_BITS, params_ptr) if comm.rank == 0 else None
        _check(D.peba1_sharded_function_f(comm.ptr, result_b, _ptr_array(sample_slots), _ptr_array(template_slots),
                                          len(sample_slots), bound_ptr, bitsize, cloud_ptr,
                                          _flags(fast_combine, fast_partial)), "peba1_sharded_function_f")
        return result_b
    finally:
        if own:
            comm.close()


def local_partial_packed(cloud_ptr, words, sample_slots, template_slots, bitsize, fast=False):
    """Phase 1 of one (logical) rank: its packed partial sum, [24 * words] int32 (peba1_sharded_partial_packed;
    `fast`: the depth-optimised distance circuit, PEBA1_DIST_FAST_PARTIAL)."""
    packed = np.zeros(PARTIAL_BITS * words, dtype=np.int32)
    _check(load().peba1_sharded_partial_packed(_ptr_array(sample_slots), _ptr_array(template_slots), len(sample_slots),
                                               bitsize, cloud_ptr, packed.ctypes.data_as(C.c_void_p),
                                               _flags(False, fast)), "partial sum")
    return packed


def combine_packed(gate_lib, params_ptr, cloud_ptr, parts, bound_ptr, fast=False):
    """Phase 3 on the calling rank: add the packed partial sums, compare with the bound
    (peba1_sharded_combine_packed).  Returns the 24-sample result array pointer."""
    result_b = _new_array(gate_lib, PARTIAL_BITS, params_ptr)
    packed = np.ascontiguousarray(np.concatenate(parts), dtype=np.int32)
    _check(load().peba1_sharded_combine_packed(result_b, packed.ctypes.data_as(C.c_void_p), len(parts), bound_ptr, cloud_ptr,
                                               FAST_COMBINE if fast else 0), "combine")
    return result_b


def sharded_match_logical(torch, gate_lib, circ_lib, params_ptr, cloud_ptr, words, sample_slots, template_slots,
                          bound_ptr, bitsize, world, device="cuda", partial_hook=None, fast_combine=False, fast_partial=False):
    """The same slot-sharded match with `world` LOGICAL ranks on one device: every rank's phase 1 runs in turn over
    its slot range of the full `sample_slots` / `template_slots` lists, the packed partial sums take the place of
    the gather's output, rank 0's phase 3 follows.  Gate for gate what `world` processes do; only the collective
    is replaced by a list.  `partial_hook(rank, tensor)` sees each rank's packed partial sums (tests hash them)."""
    nslots = len(sample_slots)
    parts = []
    for r in range(world):
        lo, hi = shard_slots(nslots, world, r)
        mine = local_partial_packed(cloud_ptr, words, sample_slots[lo:hi], template_slots[lo:hi], bitsize, fast=fast_partial)
        if partial_hook is not None:
            partial_hook(r, torch.from_numpy(mine))
        parts.append(mine)
    return combine_packed(gate_lib, params_ptr, cloud_ptr, parts, bound_ptr, fast=fast_combine)


def gather_samples(comm, all_ptr, mine_ptr, count, params_ptr):
    """`count` ciphertexts of every rank -> rank 0, rank-major (identification: the match bits)."""
    _check(load().peba1_dist_gather_samples(comm.ptr, all_ptr, mine_ptr, count, params_ptr), "peba1_dist_gather_samples")


def identify(comm, all_ptr, mine_ptr, probe_slots, template_slots, nslots, bound_ptr, bitsize, cloud_ptr, group=4, fast=False):
    """1-to-N identification through the C ABI (peba1_identify): this rank's len(template_slots) / nslots matches of the
    probe, `group` recorded per (pipelined) flush, the match bits left in `mine_ptr` and -- with a communicator --
    gathered into `all_ptr` on rank 0.  `template_slots`: the slot arrays of all local templates, template-major."""
    m_local = len(template_slots) // nslots
    assert m_local * nslots == len(template_slots) and len(probe_slots) == nslots
    _check(load().peba1_identify(comm.ptr if comm is not None else None, all_ptr, mine_ptr, _ptr_array(probe_slots),
                                 _ptr_array(template_slots), m_local, nslots, bound_ptr, bitsize, cloud_ptr, group,
                                 IDENTIFY_FAST if fast else 0), "peba1_identify")


def broadcast_samples(comm, samples_ptr, count, params_ptr, root=0):
    """`count` ciphertexts of `root` -> every rank (peba1_dist_broadcast_samples): the ONE encrypted probe of an
    identification reaches every rank's share of the gallery."""
    _check(load().peba1_dist_broadcast_samples(comm.ptr, samples_ptr, count, params_ptr, root), "peba1_dist_broadcast_samples")


def broadcast_vector(comm, params, key, vec, root=0):
    """Broadcast a circuits.EncryptedVector (one sample array per slot) in ONE collective: the slots' handles are copied
    into one contiguous array (bootsCOPY re-points handles: no data moves), that array is broadcast, and on the other
    ranks the slots are re-pointed at what arrived."""
    from . import api
    from . import lib as _l
    L = _l.load()
    bits = vec.slots[0].count
    flat = api.CiphertextArray(params, len(vec.slots) * bits)
    if comm.rank == root:
        for i, a in enumerate(vec.slots):
            for k in range(bits):
                L.bootsCOPY(flat.at(i * bits + k), a.at(k), key.cloud)
    broadcast_samples(comm, flat.ptr, flat.count, params.ptr, root)
    if comm.rank != root:
        for i, a in enumerate(vec.slots):
            for k in range(bits):
                L.bootsCOPY(a.at(k), flat.at(i * bits + k), key.cloud)
    return vec
