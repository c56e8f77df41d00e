"""Host logic of the deferred executor (peba1_amd/csrc/scheduler.cpp), no GPU: the slack-aware
levelisation must respect every dependency, keep the DAG's depth, and flatten a
"fat head + narrow tail" DAG of the kind the reference's Function_f produces."""
import numpy as np
import pytest

NOT, MUX = 17, 16


def schedule(ops, unit, balance):
    from peba1_amd import lib
    a = np.ascontiguousarray(np.array(ops, dtype=np.int32).reshape(-1, 5))
    out = np.zeros(len(a), dtype=np.int32)
    depth = lib.load().tfhe_hip_test_schedule(a.ctypes.data_as(lib.I32P), len(a), unit, 1 if balance else 0,
                                              out.ctypes.data_as(lib.I32P))
    return depth, out


def check_valid(ops, lvl, depth):
    producer = {}
    for i, (kind, dst, a, b, c) in enumerate(ops):
        for s in (a, b, c):
            if s in producer:
                p = producer[s]
                if kind == NOT:
                    assert lvl[i] == lvl[p], (i, p)            # a NOT rides on its operand's level
                else:
                    assert lvl[i] > lvl[p] or (ops[p][0] == NOT and lvl[i] > lvl[p]), (i, p, lvl[i], lvl[p])
        producer[dst] = i
        if kind != NOT:
            assert 1 <= lvl[i] <= depth
        else:
            assert 0 <= lvl[i] <= depth


def random_dag(rng, n, ninputs):
    ops, next_slot = [], ninputs
    avail = list(range(ninputs))
    not_outputs = set()
    for _ in range(n):
        r = rng.random()
        if r < 0.05:
            # the recorder never chains NOT on a pending NOT (it aliases the original operand)
            src = int(rng.choice([s for s in avail if s not in not_outputs]))
            ops.append((NOT, next_slot, src, -1, -1))
            not_outputs.add(next_slot)
        elif r < 0.10:
            a, b, c = (int(x) for x in rng.choice(avail, 3))
            ops.append((MUX, next_slot, a, b, c))
        else:
            # bias towards recent values to get depth
            a = int(avail[-1 - int(rng.integers(0, min(len(avail), 40)))])
            b = int(rng.choice(avail))
            ops.append((int(rng.integers(0, 10)), next_slot, a, b, -1))
        avail.append(next_slot)
        next_slot += 1
    return ops


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_dags_respect_dependencies_and_depth(seed):
    rng = np.random.default_rng(seed)
    ops = random_dag(rng, 3000, 16)
    d0, asap = schedule(ops, 8, False)
    d1, bal = schedule(ops, 8, True)
    assert d0 == d1
    check_valid(ops, asap, d0)
    check_valid(ops, bal, d1)
    assert bal.max() == d1


def head_and_tail_dag(nslots=48, head_depth=6, head_width=20):
    """`nslots` independent fat sub-circuits whose results feed a serial accumulation chain:
    ASAP puts every sub-circuit in the first levels and leaves a narrow tail."""
    ops, slot = [], 100
    results = []
    for _ in range(nslots):
        prev = [0, 1]
        for _ in range(head_depth):
            cur = []
            for w in range(head_width):
                ops.append((2, slot, prev[w % len(prev)], prev[(w + 1) % len(prev)], -1))
                cur.append(slot)
                slot += 1
            prev = cur
        results.append(prev[0])
    acc = results[0]
    for k in range(1, nslots):
        for _ in range(3):                      # 3 serial gates per accumulated slot
            ops.append((4, slot, acc, results[k], -1))
            acc = slot
            slot += 1
    return ops


def test_fat_head_narrow_tail_is_flattened():
    ops = head_and_tail_dag()
    unit = 64
    d0, asap = schedule(ops, unit, False)
    d1, bal = schedule(ops, unit, True)
    assert d0 == d1 == 6 + 3 * 47
    check_valid(ops, bal, d1)
    w_asap = np.bincount(asap, minlength=d0 + 1)
    w_bal = np.bincount(bal, minlength=d1 + 1)
    assert w_asap.max() == 48 * 20                       # everything at once
    assert w_bal.max() <= 4 * unit + 1                   # filled to at most the 4-unit width
    # the tail is no longer almost empty: far fewer nearly idle levels
    assert (w_bal[1:] <= 2).sum() < (w_asap[1:] <= 2).sum() // 2


def test_small_flushes_keep_asap_levels():
    ops = [(2, 10, 0, 1, -1), (4, 11, 10, 1, -1), (NOT, 12, 11, -1, -1), (MUX, 13, 12, 0, 1)]
    d, lvl = schedule(ops, 256, True)
    assert d == 3 and list(lvl) == [1, 2, 2, 3]
