"""1-to-N identification (BASELINE.json configs[3]): one encrypted probe against M enrolled
templates, i.e. M independent runs of the reference's Function_f (Math.cpp:379-387; the loop a
server would put around main.cpp:533-542, one call per enrolled client).

Matches are independent, so they shard across GPUs with no data-path collective (rank r takes
templates r, r + world, ...; bench.py --mode identify) and, on one GPU, they are recorded
`group` at a time: a flush runs the pending gates of the whole group level by level -- the narrow
tail levels of one match (~80 gates wide) are filled by the other matches of the group -- and
releases their slots, so device memory is bounded by `group`, not by M (one pending Function_f
pins ~0.22 M ciphertext slots of a pool that may grow to 4 M slots).

Only the match-bit ciphertext of each match is kept: M x (n+1) words, the one thing that leaves
the GPU.  Plumbing only: every gate runs in libtfhe-hip.
"""
from . import api, circuits
from . import lib as _l


def synthetic_template(base, k):
    """Template k of the synthetic gallery used by bench.py and the tests (k = 0: `base`)."""
    return [(v + 29 * k + 3 * i) % 256 if k else v for i, v in enumerate(base)]


def identify(params, key, probe, templates, bound, bitsize, group=4, on_group=None, circuit=None, comm=None, all_bits=None):
    """Returns a CiphertextArray of len(templates) match-bit ciphertexts: element m encrypts
    (distance(probe, templates[m]) > bound), the reference's polarity.

    probe, templates[m]: circuits.EncryptedVector; bound: 3*bitsize-sample number.
    `group` matches are recorded per flush.  The loop itself is host C++ (libpeba1-dist peba1_identify, VERDICT r3
    item 5: a C++ server reaches configs[3] with no Python in the process); this is its thin caller.
    `comm` (a dist.Comm) + `all_bits` (rank 0: a CiphertextArray of world * len(templates) samples): the match bits of
    every rank are gathered to rank 0 by the same call.
    `circuit`: circuits.function_f (default, the reference's gate sequence) or function_f_fast.
    `on_group(first, count)`: called after each group; given one, the groups are driven from here, one C call each."""
    from . import dist as pd
    circuits.load()                       # libpeba1-circuits, RTLD_GLOBAL: libpeba1-dist resolves the circuits there
    _l.load()
    fast = circuit is circuits.function_f_fast
    if circuit is not None and not fast and circuit is not circuits.function_f:
        raise ValueError("identify runs circuits.function_f or circuits.function_f_fast")
    M = len(templates)
    nslots = len(probe.slots)
    bits = api.CiphertextArray(params, M)
    P = [a.ptr for a in probe.slots]
    if on_group is None:
        T = [a.ptr for t in templates for a in t.slots]
        pd.identify(comm, all_bits.ptr if all_bits is not None else None, bits.ptr, P, T, nslots, bound.ptr, bitsize,
                    key.cloud, group=group, fast=fast)
        return bits
    for first in range(0, M, group):
        count = min(group, M - first)
        T = [a.ptr for t in templates[first:first + count] for a in t.slots]
        pd.identify(None, None, bits.at(first), P, T, nslots, bound.ptr, bitsize, key.cloud, group=group, fast=fast)
        on_group(first, count)
    if comm is not None:
        pd.gather_samples(comm, all_bits.ptr if all_bits is not None else None, bits.ptr, M, params.ptr)
    return bits
