"""Multi-GPU sharding of the match path over torch.distributed (RCCL on ROCm).

Two ways the path shards:
  * independent matches (1-to-N identification, BASELINE configs[3]): rank r runs its own
    matches against its own templates; no data-path collective, only the match-bit
    ciphertexts travel (bench.py --mode identify);
  * one match sharded by slots (BASELINE configs[2]): rank r computes the partial sum of squares
    of its slots (the reference's slot loop, Math.cpp:351-360, over a contiguous slot range),
    ONE gather moves 24 ciphertexts per rank to rank 0, rank 0 adds the partials (tree of
    23-bit adders) and runs the comparator (Math.cpp:384).  A tree of partial sums is a
    different gate DAG from the reference's left-to-right ripple: the decrypted distance and
    match bit are identical, intermediate ciphertexts are not -- they are pinned against the
    oracle evaluating the same DAG (tests/golden/sharded_match_digest.json).

The three phases are separate functions so that the same code serves real ranks
(`sharded_match`: one process per GPU, phases joined by dist.gather) and logical ranks on ONE
device (`sharded_match_logical`: the phases of every rank run one after the other in one
process, the exchange goes through the same packed device buffers) -- the only way to run
BASELINE configs[2] at size where one GPU is available.

The provider of the gate API is passed in (`gate_lib`, `circ_lib`): the product passes
libtfhe-hip / libpeba1-circuits; CPU tests pass a plaintext provider to exercise exactly
this sharding and exchange logic under gloo.
"""
import ctypes as C

import numpy as np

PARTIAL_BITS = 24


def shard_slots(nslots, world, rank):
    """Contiguous slot range [lo, hi) of `rank`; earlier ranks take the remainder."""
    base, rem = divmod(nslots, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = C.cast(p, C.c_void_p)
    return arr


class _Provider:
    """Private ctypes handles of the gate and circuit libraries: the prototypes set here must not
    disturb the callers' own bindings."""

    def __init__(self, gate_lib, circ_lib):
        g = C.CDLL(gate_lib._name)
        c = C.CDLL(circ_lib._name)
        V, I = C.c_void_p, C.c_int32
        g.new_gate_bootstrapping_ciphertext_array.restype = V
        g.new_gate_bootstrapping_ciphertext_array.argtypes = [I, V]
        g.delete_gate_bootstrapping_ciphertext_array.restype = None
        g.delete_gate_bootstrapping_ciphertext_array.argtypes = [I, V]
        for name in ("tfhe_hip_export_samples", "tfhe_hip_import_samples", "tfhe_hip_export_samples_device",
                     "tfhe_hip_import_samples_device"):
            if hasattr(g, name):               # a CPU test provider has no device entry points
                f = getattr(g, name)
                f.restype = C.c_int
                f.argtypes = [V, I, V, V]
        c.peba1_partial_distance.restype = None
        c.peba1_partial_distance.argtypes = [V, V, V, C.c_int, C.c_int, V]
        for name in ("peba1_combine_and_compare", "peba1_combine_and_compare_fast"):
            f = getattr(c, name)
            f.restype = None
            f.argtypes = [V, V, C.c_int, V, V]
        self.g, self.c = g, c
        self.new_arr = g.new_gate_bootstrapping_ciphertext_array
        self.del_arr = g.delete_gate_bootstrapping_ciphertext_array

    def check(self, rc, what):
        if rc != 0:
            msg = b""
            if hasattr(self.g, "tfhe_hip_last_error"):
                self.g.tfhe_hip_last_error.restype = C.c_char_p
                msg = self.g.tfhe_hip_last_error() or b""
            raise RuntimeError(f"{what} failed: {msg.decode()}")


def local_partial(torch, prov, params_ptr, cloud_ptr, words, sample_slots, template_slots, bitsize, device):
    """Phase 1 (every rank): the partial sum of squares of this rank's slots as a packed
    [24 * words] int32 tensor on `device`, ready for the collective."""
    partial = prov.new_arr(PARTIAL_BITS, params_ptr)
    prov.c.peba1_partial_distance(partial, _ptr_array(sample_slots), _ptr_array(template_slots),
                                  len(sample_slots), bitsize, cloud_ptr)
    mine = torch.empty(PARTIAL_BITS * words, dtype=torch.int32, device=device)
    if device == "cuda":
        # flushes the recorded gates, then gathers the 24 slots into `mine` on the library's own
        # stream and waits for it: the buffer is complete when this returns
        prov.check(prov.g.tfhe_hip_export_samples_device(partial, PARTIAL_BITS, params_ptr, C.c_void_p(mine.data_ptr())),
                   "export of the partial sums")
    else:
        buf = np.zeros(PARTIAL_BITS * words, dtype=np.int32)
        prov.check(prov.g.tfhe_hip_export_samples(partial, PARTIAL_BITS, params_ptr, buf.ctypes.data_as(C.c_void_p)),
                   "export of the partial sums")
        mine.copy_(torch.from_numpy(buf))
    prov.del_arr(PARTIAL_BITS, partial)
    return mine


def combine(torch, prov, params_ptr, cloud_ptr, gathered, bound_ptr, device, fast=False):
    """Phase 3 (rank 0): import the gathered partial sums, add them, compare with the bound.
    Returns the 24-sample result array pointer (element 0 is the match bit).
    fast=False: pairwise tree of the reference's ripple adders + its comparator (the DAG the golden
    digest pins); fast=True: carry-save compressor + prefix adder + prefix comparator, ~20 levels
    instead of ~290 for 8 ranks -- what a latency-bound rank 0 wants."""
    if device == "cuda":
        # The library reads these buffers on its own non-blocking stream.  A c10d collective only
        # orders torch's current stream behind the RCCL stream, so the host must wait for the
        # device here, or the import could run before the gather has landed.
        torch.cuda.synchronize()
    parts = []
    for g in gathered:
        p = prov.new_arr(PARTIAL_BITS, params_ptr)
        if device == "cuda":
            prov.check(prov.g.tfhe_hip_import_samples_device(p, PARTIAL_BITS, params_ptr, C.c_void_p(g.data_ptr())),
                       "import of a gathered partial sum")
        else:
            buf = np.ascontiguousarray(g.numpy())
            prov.check(prov.g.tfhe_hip_import_samples(p, PARTIAL_BITS, params_ptr, buf.ctypes.data_as(C.c_void_p)),
                       "import of a gathered partial sum")
        parts.append(p)
    result_b = prov.new_arr(PARTIAL_BITS, params_ptr)
    (prov.c.peba1_combine_and_compare_fast if fast else prov.c.peba1_combine_and_compare)(
        result_b, _ptr_array(parts), len(parts), bound_ptr, cloud_ptr)
    for p in parts:
        prov.del_arr(PARTIAL_BITS, p)
    return result_b


def sharded_match(dist, torch, gate_lib, circ_lib, params_ptr, cloud_ptr, words, sample_slots, template_slots,
                  bound_ptr, bitsize, device="cuda", fast_combine=False, partial_hook=None):
    """Slot-sharded Function_f across the ranks of `dist`.  `sample_slots` / `template_slots`: this
    rank's slot arrays (LweSample* each, `bitsize` samples).  Returns the 24-sample result array
    pointer on rank 0 (caller frees it with delete_gate_bootstrapping_ciphertext_array(24, p)),
    None elsewhere.  device="cuda": the exchange buffers are device tensors (RCCL); "cpu": host tensors
    (gloo -- how several processes rehearse this on one GPU).  `partial_hook(rank, tensor)` sees this
    rank's packed partial sums before the gather."""
    rank, world = dist.get_rank(), dist.get_world_size()
    prov = _Provider(gate_lib, circ_lib)
    mine = local_partial(torch, prov, params_ptr, cloud_ptr, words, sample_slots, template_slots, bitsize, device)
    if partial_hook is not None:
        partial_hook(rank, mine)
    # the exchange: 24 ciphertexts per rank -> rank 0, one collective
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank != 0:
        return None
    return combine(torch, prov, params_ptr, cloud_ptr, gathered, bound_ptr, device, fast=fast_combine)


def sharded_match_logical(torch, gate_lib, circ_lib, params_ptr, cloud_ptr, words, sample_slots, template_slots,
                          bound_ptr, bitsize, world, device="cuda", partial_hook=None, fast_combine=False):
    """The same slot-sharded match with `world` LOGICAL ranks on one device: every rank's phase 1
    runs in turn over its slot range of the full `sample_slots` / `template_slots` lists, the packed
    partial sums take the place of the gather's output, rank 0's phase 3 follows.  Gate for gate and
    buffer for buffer what `world` processes do; only the collective is replaced by a list.
    `partial_hook(rank, tensor)` sees each rank's packed partial sums (tests hash them)."""
    prov = _Provider(gate_lib, circ_lib)
    nslots = len(sample_slots)
    gathered = []
    for r in range(world):
        lo, hi = shard_slots(nslots, world, r)
        mine = local_partial(torch, prov, params_ptr, cloud_ptr, words, sample_slots[lo:hi], template_slots[lo:hi],
                             bitsize, device)
        if partial_hook is not None:
            partial_hook(r, mine)
        gathered.append(mine)
    return combine(torch, prov, params_ptr, cloud_ptr, gathered, bound_ptr, device, fast=fast_combine)
