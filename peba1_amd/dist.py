"""Multi-GPU sharding of the match path over torch.distributed (RCCL on ROCm).

Two ways the path shards:
  * independent matches (1-to-N identification): rank r runs its own Function_f; the only
    exchange is a gather of the match-bit ciphertexts (bench.py does this inline);
  * one match sharded by slots: rank r computes the partial sum of squares of its slots,
    ONE gather moves 24 ciphertexts per rank to rank 0, rank 0 adds the partials (tree of
    23-bit adders) and runs the comparator.  A tree of partial sums is a different gate DAG
    from the reference's left-to-right ripple (Math.cpp:351-360): the decrypted distance
    and match bit are identical, intermediate ciphertexts are not.

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


def sharded_match(dist, torch, gate_lib, circ_lib, params_ptr, cloud_ptr, words, sample_slots, template_slots,
                  bound_ptr, bitsize, device="cuda"):
    """Slot-sharded Function_f.  `sample_slots` / `template_slots`: this rank's slot arrays
    (LweSample* each, `bitsize` samples).  Returns the 24-sample result array pointer on
    rank 0 (caller frees it with delete_gate_bootstrapping_ciphertext_array(24, p)), None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    # private handles: the prototypes set below must not disturb the callers' own bindings
    gate_lib = C.CDLL(gate_lib._name)
    circ_lib = C.CDLL(circ_lib._name)
    new_arr = gate_lib.new_gate_bootstrapping_ciphertext_array
    new_arr.restype = C.c_void_p
    new_arr.argtypes = [C.c_int32, C.c_void_p]
    del_arr = gate_lib.delete_gate_bootstrapping_ciphertext_array
    del_arr.restype = None
    del_arr.argtypes = [C.c_int32, C.c_void_p]

    partial = new_arr(PARTIAL_BITS, params_ptr)
    circ_lib.peba1_partial_distance.restype = None
    circ_lib.peba1_partial_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    circ_lib.peba1_partial_distance(partial, _ptr_array(sample_slots), _ptr_array(template_slots),
                                    len(sample_slots), bitsize, cloud_ptr)

    # the exchange: 24 ciphertexts per rank -> rank 0, one collective
    mine = torch.empty(PARTIAL_BITS * words, dtype=torch.int32, device=device)
    if device == "cuda":
        gate_lib.tfhe_hip_export_samples_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        gate_lib.tfhe_hip_export_samples_device(partial, PARTIAL_BITS, params_ptr, C.c_void_p(mine.data_ptr()))
    else:
        buf = np.zeros(PARTIAL_BITS * words, dtype=np.int32)
        gate_lib.tfhe_hip_export_samples.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        gate_lib.tfhe_hip_export_samples(partial, PARTIAL_BITS, params_ptr, buf.ctypes.data_as(C.c_void_p))
        mine.copy_(torch.from_numpy(buf))
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    del_arr(PARTIAL_BITS, partial)
    if rank != 0:
        return None

    parts = []
    for r in range(world):
        p = new_arr(PARTIAL_BITS, params_ptr)
        if device == "cuda":
            gate_lib.tfhe_hip_import_samples_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
            gate_lib.tfhe_hip_import_samples_device(p, PARTIAL_BITS, params_ptr, C.c_void_p(gathered[r].data_ptr()))
        else:
            buf = np.ascontiguousarray(gathered[r].numpy())
            gate_lib.tfhe_hip_import_samples.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
            gate_lib.tfhe_hip_import_samples(p, PARTIAL_BITS, params_ptr, buf.ctypes.data_as(C.c_void_p))
        parts.append(p)
    result_b = new_arr(PARTIAL_BITS, params_ptr)
    circ_lib.peba1_combine_and_compare.restype = None
    circ_lib.peba1_combine_and_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    circ_lib.peba1_combine_and_compare(result_b, _ptr_array(parts), world, bound_ptr, cloud_ptr)
    for p in parts:
        del_arr(PARTIAL_BITS, p)
    return result_b
