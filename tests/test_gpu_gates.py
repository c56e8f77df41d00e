"""GPU parity of the boots* gates (a4-a9) against the CPU oracle: decrypted truth tables
AND ciphertext words, through the upstream-compatible C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUTH = {"AND": lambda a, b: a & b, "OR": lambda a, b: a | b, "XOR": lambda a, b: a ^ b,
         "XNOR": lambda a, b: 1 - (a ^ b), "NAND": lambda a, b: 1 - (a & b), "NOR": lambda a, b: 1 - (a | b),
         "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
         "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b)}


def test_two_input_gates_batched(p128_keys, oracle):
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(21)
    abits = [0, 0, 1, 1]
    bbits = [0, 1, 0, 1]
    a = api.CiphertextArray(pp, 4).encrypt(abits, ks)
    b = api.CiphertextArray(pp, 4).encrypt(bbits, ks)
    wa, wb = a.words(), b.words()
    for name, f in TRUTH.items():
        res = api.CiphertextArray(pp, 4)
        api.gate_batch(name, res, a, b, ks)
        got = res.words()
        for i in range(4):
            assert (got[i] == oks.gate(name, wa[i], wb[i])).all(), (name, i)
        assert list(res.decrypt(ks)) == [f(x, y) for x, y in zip(abits, bbits)], name


def test_single_call_api_immediate_mode(p128_keys, oracle):
    """The drop-in path: one boots* call per gate, result complete on return."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    api.set_deferred(False)                     # strict per-call completion (TFHE_HIP_DEFERRED=0)
    assert L.tfhe_hip_get_deferred() == 0
    L.tfhe_hip_set_encrypt_seed(5)
    x = api.CiphertextArray(pp, 3).encrypt([1, 0, 1], ks)
    wx = x.words()
    r = api.CiphertextArray(pp, 4)
    L.bootsXOR(r.at(0), x.at(0), x.at(1), ks.cloud)
    L.bootsMUX(r.at(1), x.at(0), x.at(1), x.at(2), ks.cloud)
    L.bootsNOT(r.at(2), x.at(0), ks.cloud)
    L.bootsCONSTANT(r.at(3), 1, ks.cloud)
    # host mirror is valid on return in immediate mode
    n = pp.n
    mirror = np.array([list(r.ptr[i].a[0:n]) + [r.ptr[i].b] for i in range(4)], dtype=np.int32)
    assert (mirror[0] == oks.gate("XOR", wx[0], wx[1])).all()
    assert (mirror[1] == oks.mux(wx[0], wx[1], wx[2])).all()
    assert (mirror[2] == oks.gate_not(wx[0])).all()
    assert (mirror[3] == oks.constant(1)).all()
    assert list(r.decrypt(ks)) == [1, 0, 0, 1]
    # result aliasing an input (reference: Math.cpp:272)
    L.bootsAND(x.at(0), x.at(0), x.at(2), ks.cloud)
    assert (x.words()[0] == oks.gate("AND", wx[0], wx[2])).all()


def test_default_mode_records_and_decrypt_observes(oracle):
    """A fresh process with no environment switch records gates (tfhe_hip_get_deferred() == 1): an
    unmodified caller gets batched execution, and everything it can observe through the API --
    decrypted bits, exported words -- is what per-call execution gives."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from peba1_amd import api, lib\n"
            "L = lib.load(); assert L.tfhe_hip_get_deferred() == 1\n"
            "pp = api.ParameterSet(128); ks = api.SecretKeySet(pp, 0x5EBA2, device=True)\n"
            "L.tfhe_hip_set_encrypt_seed(77)\n"
            "x = api.CiphertextArray(pp, 3).encrypt([1, 0, 1], ks); r = api.CiphertextArray(pp, 3)\n"
            "L.bootsXOR(r.at(0), x.at(0), x.at(1), ks.cloud)\n"
            "L.bootsAND(r.at(1), r.at(0), x.at(2), ks.cloud)\n"
            "L.bootsMUX(r.at(2), r.at(1), x.at(1), x.at(0), ks.cloud)\n"
            "st = api.stats(); assert st['blind_rotates'] == 0, st          # nothing has run yet\n"
            "bits = [int(b) for b in r.decrypt(ks)]\n"
            "st = api.stats(); assert st['blind_rotates'] == 4 and st['flushes'] == 1, st\n"
            "np.save(sys.argv[1], r.words()); print('BITS', bits)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        outs = {}
        for mode in ("default", "0"):
            env = {k: v for k, v in os.environ.items() if k != "TFHE_HIP_DEFERRED"}
            body = code
            if mode == "0":
                env["TFHE_HIP_DEFERRED"] = "0"
                body = code.replace("== 1\n", "== 0\n", 1).replace("assert st['blind_rotates'] == 0, st", "pass").replace(
                    "and st['flushes'] == 1", "")
            p = subprocess.run([sys.executable, "-c", body, t + "/w_%s.npy" % mode], env=env, capture_output=True, text=True, timeout=300)
            assert p.returncode == 0 and "BITS [1, 1, 0]" in p.stdout, p.stdout + p.stderr[-2000:]
            outs[mode] = np.load(t + "/w_%s.npy" % mode)
        assert (outs["default"] == outs["0"]).all()            # same ciphertexts either way


def test_mux_truth_table_deferred(p128_keys, oracle):
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(8)
    combos = [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]
    A = api.CiphertextArray(pp, 8).encrypt([t[0] for t in combos], ks)
    B = api.CiphertextArray(pp, 8).encrypt([t[1] for t in combos], ks)
    Cc = api.CiphertextArray(pp, 8).encrypt([t[2] for t in combos], ks)
    wa, wb, wc = A.words(), B.words(), Cc.words()
    R = api.CiphertextArray(pp, 8)
    api.set_deferred(True)
    try:
        for i in range(8):
            L.bootsMUX(R.at(i), A.at(i), B.at(i), Cc.at(i), ks.cloud)
        assert api.flush() == 1          # eight independent MUXes: one level
    finally:
        api.set_deferred(False)
    got = R.words()
    for i, (a, b, c) in enumerate(combos):
        assert (got[i] == oks.mux(wa[i], wb[i], wc[i])).all(), i
    assert list(R.decrypt(ks)) == [b if a else c for a, b, c in combos]


def test_deferred_chain_with_overwrite_and_free(p128_keys, oracle):
    """A dependent chain recorded with the reference's habits: a temporary overwritten,
    copied and freed before the flush (Math.cpp:34-49)."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(13)
    x = api.CiphertextArray(pp, 3).encrypt([1, 1, 0], ks)
    wx = x.words()
    out = api.CiphertextArray(pp, 2)
    api.set_deferred(True)
    try:
        tmp = api.CiphertextArray(pp, 1)
        keep = api.CiphertextArray(pp, 1)
        L.bootsXOR(tmp.at(0), x.at(0), x.at(1), ks.cloud)       # t1 = a ^ b
        L.bootsXOR(out.at(0), tmp.at(0), x.at(2), ks.cloud)     # s  = t1 ^ c
        L.bootsAND(tmp.at(0), x.at(0), x.at(1), ks.cloud)       # overwrite tmp
        L.bootsCOPY(keep.at(0), tmp.at(0), ks.cloud)
        L.bootsAND(tmp.at(0), x.at(0), x.at(2), ks.cloud)       # overwrite again
        L.bootsXOR(out.at(1), keep.at(0), tmp.at(0), ks.cloud)
        tmp.close()
        keep.close()
        assert api.flush() == 2
    finally:
        api.set_deferred(False)
    t1 = oks.gate("XOR", wx[0], wx[1])
    s = oks.gate("XOR", t1, wx[2])
    t2 = oks.gate("AND", wx[0], wx[1])
    t3 = oks.gate("AND", wx[0], wx[2])
    c = oks.gate("XOR", t2, t3)
    got = out.words()
    assert (got[0] == s).all() and (got[1] == c).all()
    assert list(out.decrypt(ks)) == [0, 1]


def test_fresh_samples_are_trivial_zero(p128_keys, oracle):
    """Function_f reads never-written samples; the shim defines them as the
    trivial encryption of bit 0, i.e. what bootsCONSTANT(.., 0) writes."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(2)
    one = api.CiphertextArray(pp, 1).encrypt([1], ks)
    fresh = api.CiphertextArray(pp, 1)
    r = api.CiphertextArray(pp, 1)
    L.bootsXOR(r.at(0), one.at(0), fresh.at(0), ks.cloud)
    assert (fresh.words()[0] == oks.constant(0)).all()
    assert (r.words()[0] == oks.gate("XOR", one.words()[0], oks.constant(0))).all()
    assert fresh.decrypt(ks)[0] == 0 and r.decrypt(ks)[0] == 1


def test_not_of_not_and_not_chains_deferred(p128_keys, oracle):
    """NOT(NOT(x)) recorded in one flush must not race: it aliases x (exact double negation)."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(31)
    x = api.CiphertextArray(pp, 2).encrypt([1, 0], ks)
    wx = x.words()
    r = api.CiphertextArray(pp, 4)
    api.set_deferred(True)
    try:
        L.bootsNOT(r.at(0), x.at(0), ks.cloud)
        L.bootsNOT(r.at(1), r.at(0), ks.cloud)                 # = x[0]
        L.bootsAND(r.at(2), r.at(1), x.at(1), ks.cloud)
        L.bootsNOT(r.at(3), r.at(2), ks.cloud)                 # NOT of a bootstrapped value
        api.flush()
    finally:
        api.set_deferred(False)
    got = r.words()
    assert (got[0] == oks.gate_not(wx[0])).all()
    assert (got[1] == wx[0]).all()
    g = oks.gate("AND", wx[0], wx[1])
    assert (got[2] == g).all() and (got[3] == oks.gate_not(g)).all()
    assert list(r.decrypt(ks)) == [0, 1, 0, 1]


def test_balanced_and_asap_schedules_give_identical_ciphertexts(p128_keys, oracle):
    """A multiplier + adder circuit (fat levels then a carry chain) run with slack-aware level
    filling and with plain ASAP levels: same ciphertext words, correct plaintext."""
    from peba1_amd import api, circuits, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    results = []
    for balance in (1, 0):
        api.set_tuning("balance_levels", balance)
        L.tfhe_hip_set_encrypt_seed(77)
        a = circuits.encrypt_number(pp, 13, 9, ks)
        b = circuits.encrypt_number(pp, 11, 9, ks)
        prod = api.CiphertextArray(pp, 24)
        api.set_deferred(True)
        try:
            circuits.load().peba1_multiply(prod.ptr, a.ptr, b.ptr, 4, ks.cloud)
            api.flush()
        finally:
            api.set_deferred(False)
        results.append(prod.words())
        assert circuits.decrypt_number(prod, ks, 23) == 13 * 11
    api.set_tuning("balance_levels", 1)
    assert (results[0] == results[1]).all()


def test_recording_survives_a_tiny_slot_pool():
    """With a pool far smaller than the circuit, the recorder flushes before the pool runs dry;
    the result is the same number.  Runs in a subprocess (the pool size is fixed at first use)."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
from peba1_amd import api, circuits, lib
L = lib.load()
pp = api.ParameterSet(128)
ks = api.SecretKeySet(pp, 0x5EBA2)
L.tfhe_hip_set_encrypt_seed(3)
a = circuits.encrypt_number(pp, 201, 9, ks)
b = circuits.encrypt_number(pp, 77, 9, ks)
prod = api.CiphertextArray(pp, 24)
api.set_deferred(True)
circuits.load().peba1_multiply(prod.ptr, a.ptr, b.ptr, 8, ks.cloud)     # 1,296 gates, ~2,000 slots
api.flush()
api.set_deferred(False)
assert api.stats()["flushes"] >= 2, api.stats()
assert circuits.decrypt_number(prod, ks, 23) == 201 * 77
print("OK", api.stats()["flushes"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TFHE_HIP_POOL_SLOTS="5000")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_random_gate_sweep_matches_oracle(p128_keys, oracle):
    """256 random gate instances over all ten two-input gate types, inputs that are fresh
    encryptions, trivial constants and previous gate outputs: every output word equals the oracle's."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    rng = np.random.default_rng(2026)
    L.tfhe_hip_set_encrypt_seed(2026)
    nin = 24
    bits = rng.integers(0, 2, nin)
    x = api.CiphertextArray(pp, nin).encrypt(bits, ks)
    L.bootsCONSTANT(x.at(0), 1, ks.cloud)                    # two trivial inputs: every blind-rotate step is skipped
    L.bootsCONSTANT(x.at(1), 0, ks.cloud)
    wx = x.words()
    names = list(api.GATE_CODES)
    per_type = 26
    for name in names[:10]:
        ia = rng.integers(0, nin, per_type)
        ib = rng.integers(0, nin, per_type)
        A = api.CiphertextArray(pp, per_type).set_words(wx[ia])
        B = api.CiphertextArray(pp, per_type).set_words(wx[ib])
        R = api.CiphertextArray(pp, per_type)
        api.gate_batch(name, R, A, B, ks)
        got = R.words()
        want = oks.gate_batch(name, wx[ia], wx[ib], nthreads=16)
        assert (got == want).all(), name


def test_server_side_evaluation_from_files(p128_keys, oracle, tmp_path):
    """the client writes the cloud keyset and ciphertexts, the evaluating side
    loads only those (no secret key), and the result file decrypts and matches the oracle."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(77)
    abits, bbits = [0, 1, 1, 0, 1, 0], [1, 1, 0, 0, 1, 0]
    a = api.CiphertextArray(pp, 6).encrypt(abits, ks)
    b = api.CiphertextArray(pp, 6).encrypt(bbits, ks)
    wa, wb = a.words(), b.words()
    ks.save_cloud(tmp_path / "cloud.key")
    a.save(tmp_path / "a.ct")
    b.save(tmp_path / "b.ct")

    cloud = api.CloudKeySet.load(tmp_path / "cloud.key")        # device image built at first gate
    assert cloud.params.n == pp.n and cloud.params.N == pp.N
    sa = api.CiphertextArray(cloud.params, 6).load(tmp_path / "a.ct")
    sb = api.CiphertextArray(cloud.params, 6).load(tmp_path / "b.ct")
    res = api.CiphertextArray(cloud.params, 6)
    api.set_deferred(True)
    try:
        for i in range(6):
            L.bootsXOR(res.at(i), sa.at(i), sb.at(i), cloud.cloud)
            L.bootsAND(res.at(i), res.at(i), sa.at(i), cloud.cloud)
        res.save(tmp_path / "res.ct")                            # flushes what feeds the samples
    finally:
        api.set_deferred(False)
    sa.close(); sb.close(); res.close()
    cloud.close()

    back = api.CiphertextArray(pp, 6).load(tmp_path / "res.ct")
    got = back.words()
    for i in range(6):
        want = oks.gate("AND", oks.gate("XOR", wa[i], wb[i]), wa[i])
        assert (got[i] == want).all(), i
    assert back.decrypt(ks).tolist() == [(x ^ y) & x for x, y in zip(abits, bbits)]
    # the session keyset still works after the loaded one is released
    api.gate_batch("OR", back, a, b, ks)
    assert back.decrypt(ks).tolist() == [x | y for x, y in zip(abits, bbits)]


def test_edge_cases_empty_ragged_and_wide(p128_keys, oracle):
    """Empty batches and flushes are no-ops; a bad gate code is refused; a batch wider than one
    key-switch chunk (8192 gates: tiled key switch in two chunks, blind rotate in 17 rounds)
    still matches the oracle gate by gate (sampled) and the truth table everywhere."""
    from peba1_amd import api, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    e = api.CiphertextArray(pp, 1)
    assert L.tfhe_hip_gate_batch(api.GATE_CODES["AND"], e.ptr, e.ptr, e.ptr, 0, ks.cloud) == 0
    assert api.flush() == 0
    assert L.tfhe_hip_gate_batch(99, e.ptr, e.ptr, e.ptr, 1, ks.cloud) == -1
    assert b"bad gate code" in L.tfhe_hip_last_error()
    assert L.tfhe_hip_export_samples(e.ptr, 0, pp.ptr, None) == 0
    # wide batch
    G = 8192 + 37
    rng = np.random.default_rng(8192)
    abits, bbits = rng.integers(0, 2, G), rng.integers(0, 2, G)
    L.tfhe_hip_set_encrypt_seed(4242)
    a = api.CiphertextArray(pp, G).encrypt(abits, ks)
    b = api.CiphertextArray(pp, G).encrypt(bbits, ks)
    res = api.CiphertextArray(pp, G)
    api.gate_batch("ORNY", res, a, b, ks)
    assert (res.decrypt(ks) == ((1 - abits) | bbits)).all()
    wa, wb, got = a.words(), b.words(), res.words()
    for i in (0, 15, 16, 8191, 8192, G - 1):
        assert (got[i] == oks.gate("ORNY", wa[i], wb[i])).all(), i


def test_asynchronous_flush_pipelines_circuits_with_the_same_results(p128_keys):
    """tfhe_hip_flush_async: the launches of one recording are enqueued and the caller goes on recording the next --
    here a chain of three multipliers, each reading the one before (so a flush in flight feeds the next recording),
    temporaries freed while their flush is still in flight, a decrypt in the middle.  Same ciphertexts as three
    synchronous flushes."""
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    L = lib.load()
    results = []
    for mode in ("sync", "async"):
        L.tfhe_hip_set_encrypt_seed(4711)
        a = circuits.encrypt_number(pp, 11, 9, ks)
        b = circuits.encrypt_number(pp, 13, 9, ks)
        api.set_deferred(True)
        try:
            prods = []
            x = a
            for k in range(3):
                p = api.CiphertextArray(pp, 24)
                circuits.load().peba1_multiply(p.ptr, x.ptr, b.ptr, 4, ks.cloud)      # 4-bit operands: low bits of x
                (api.flush if mode == "sync" else api.flush_async)()
                if k == 1:
                    assert circuits.decrypt_number(p, ks, 23) == ((11 * 13) % 16) * 13  # observing waits for the flight
                prods.append(p)
                x = p
                del p
            if mode == "async":
                assert api.wait() == 0
            words = np.concatenate([q.words() for q in prods])
        finally:
            api.set_deferred(False)
        results.append(words)
        assert circuits.decrypt_number(prods[0], ks, 23) == 11 * 13
    assert (results[0] == results[1]).all()


def test_concurrent_host_threads_share_the_recorder(p128_keys):
    """Four host threads call the library at once (ctypes releases the GIL): each encrypts its own operands, records
    16-bit Hamming matches in default (recording) mode and decrypts -- a decrypt in one thread flushes whatever the
    others have recorded so far.  The recorder, the slot pool and the flush are serialised by the library's lock;
    every thread must read its own plaintext results.  (The reference is single-threaded; a server matching several
    probes from worker threads is not.)"""
    import threading
    from peba1_amd import api, circuits
    pp, ks, _ = p128_keys
    errors, done = [], []
    w = circuits.hamming_count_bits(16)

    def worker(t):
        try:
            for it in range(3):
                a, b = (0x1234 * (t + 1) + 77 * it) & 0xFFFF, (0xBEEF ^ (0x0F0F * t) ^ (it << 7)) & 0xFFFF
                hd = bin(a ^ b).count("1")
                A = circuits.encrypt_number(pp, a, 16, ks)
                B = circuits.encrypt_number(pp, b, 16, ks)
                cnt = api.CiphertextArray(pp, w)
                circuits.hamming_distance(cnt, A, B, 16, ks)
                outs = []
                for bound in (hd - 1, hd):
                    rb = api.CiphertextArray(pp, w)
                    circuits.hamming_match(rb, A, B, 16, circuits.encrypt_number(pp, max(bound, 0), w, ks), ks)
                    outs.append((max(bound, 0), rb))
                got = circuits.decrypt_number(cnt, ks)                      # observes: flushes
                assert got == hd, (t, it, got, hd)
                for bound, rb in outs:
                    assert rb.decrypt(ks)[0] == (1 if hd > bound else 0), (t, it, bound)
            done.append(t)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    api.set_deferred(True)
    try:
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not any(th.is_alive() for th in threads), "a thread is stuck"
        api.flush()
    finally:
        api.set_deferred(False)
    assert not errors, errors
    assert sorted(done) == [0, 1, 2, 3]
