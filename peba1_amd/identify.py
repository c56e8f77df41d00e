"""1-to-N identification (BASELINE.json configs[3]): one encrypted probe against M enrolled
templates, i.e. M independent runs of the reference's Function_f (Math.cpp:379-387; the loop a
server would put around main.cpp:533-542, one call per enrolled client).

Matches are independent, so they shard across GPUs with no data-path collective (rank r takes
templates r, r + world, ...; bench.py --mode identify) and, on one GPU, they are recorded
`group` at a time: a flush runs the pending gates of the whole group level by level -- the narrow
tail levels of one match (~80 gates wide) are filled by the other matches of the group -- and
releases their slots, so device memory is bounded by `group`, not by M (one pending Function_f
pins ~0.22 M ciphertext slots of the 2 M-slot pool).

Only the match-bit ciphertext of each match is kept: M x (n+1) words, the one thing that leaves
the GPU.  Plumbing only: every gate runs in libtfhe-hip.
"""
from . import api, circuits
from . import lib as _l


def synthetic_template(base, k):
    """Template k of the synthetic gallery used by bench.py and the tests (k = 0: `base`)."""
    return [(v + 29 * k + 3 * i) % 256 if k else v for i, v in enumerate(base)]


def identify(params, key, probe, templates, bound, bitsize, group=4, on_group=None, circuit=None):
    """Returns a CiphertextArray of len(templates) match-bit ciphertexts: element m encrypts
    (distance(probe, templates[m]) > bound), the reference's polarity.

    probe, templates[m]: circuits.EncryptedVector; bound: 3*bitsize-sample number.
    `group` matches are recorded per flush; `on_group(first, count)` is called after each flush.
    `circuit`: circuits.function_f (default, the reference's gate sequence) or function_f_fast."""
    L = _l.load()
    circuit = circuit or circuits.function_f
    M = len(templates)
    bits = api.CiphertextArray(params, M)
    was_deferred = api.get_deferred()
    api.set_deferred(True)
    try:
        for first in range(0, M, group):
            count = min(group, M - first)
            for m in range(first, first + count):
                rb = api.CiphertextArray(params, 3 * bitsize)
                circuit(rb, probe, templates[m], bound, bitsize, key)
                L.bootsCOPY(bits.at(m), rb.at(0), key.cloud)      # re-points a handle: no data moves
                rb.close()
            # pipelined: the launches of this group are enqueued and the host goes on recording the next group while the
            # device works; the next flush (or the final wait) completes this one
            api.flush_async()
            if on_group is not None:
                on_group(first, count)
        api.wait()
    finally:
        api.set_deferred(was_deferred)
    return bits
