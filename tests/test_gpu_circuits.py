"""GPU runs of whole circuits through libpeba1-circuits + libtfhe-hip in deferred mode:
decrypted results against plaintext arithmetic, and ciphertext words against the CPU oracle
evaluating the same gate sequence where that is cheap."""
import ctypes as C
import datetime
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_adder_ciphertexts_match_oracle(p128_keys, oracle):
    """bootsADD1bit (Math.cpp:27-50): 7 bootstraps, every output word equal to the oracle's."""
    from peba1_amd import api, circuits, lib
    pp, ks, oks = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(41)
    x = api.CiphertextArray(pp, 3).encrypt([1, 1, 1], ks)       # a, b, carry-in
    w = x.words()
    out = api.CiphertextArray(pp, 1)
    api.set_deferred(True)
    try:
        circuits.load().peba1_add_1bit(out.ptr, x.at(0), x.at(1), x.at(2), ks.cloud)
        api.flush()
    finally:
        api.set_deferred(False)
    t = oks.gate("XOR", w[0], w[1])
    s = oks.gate("XOR", t, w[2])
    ab = oks.gate("AND", w[0], w[1])
    ac = oks.gate("AND", w[0], w[2])
    c1 = oks.gate("XOR", ab, ac)
    cb = oks.gate("AND", w[2], w[1])
    c2 = oks.gate("XOR", c1, cb)
    assert (out.words()[0] == s).all()
    assert (x.words()[2] == c2).all()                            # carry updated in place
    assert out.decrypt(ks)[0] == 1 and x.decrypt(ks)[2] == 1


def test_small_euclidean_match_and_hamming_match(p128_keys):
    """Function_f on 3 slots and a 16-bit Hamming match: decrypted bits equal the plaintext rule
    (distance > bound), SURVEY D2 polarity."""
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(43)
    tmpl_v, probe_v = [12, 200, 77], [15, 190, 78]
    d = sum((a - b) ** 2 for a, b in zip(probe_v, tmpl_v))        # 9 + 100 + 1
    T = circuits.EncryptedVector(pp, tmpl_v, 8, ks)
    S = circuits.EncryptedVector(pp, probe_v, 8, ks)
    api.set_deferred(True)
    try:
        outs = []
        for bound in (d - 1, d, 4000):
            rb = api.CiphertextArray(pp, 24)
            circuits.function_f(rb, S, T, circuits.encrypt_number(pp, bound, 24, ks), 8, ks)
            outs.append((bound, rb))
        a_bits, b_bits = 0xBEEF, 0x1234
        hd = bin(a_bits ^ b_bits).count("1")
        w = circuits.hamming_count_bits(16)
        A = circuits.encrypt_number(pp, a_bits, 16, ks)
        B = circuits.encrypt_number(pp, b_bits, 16, ks)
        cnt = api.CiphertextArray(pp, w)
        circuits.hamming_distance(cnt, A, B, 16, ks)
        hm = []
        for bound in (hd - 1, hd):
            rb = api.CiphertextArray(pp, w)
            circuits.hamming_match(rb, A, B, 16, circuits.encrypt_number(pp, bound, w, ks), ks)
            hm.append((bound, rb))
        api.flush()
    finally:
        api.set_deferred(False)
    for bound, rb in outs:
        assert rb.decrypt(ks)[0] == (1 if d > bound else 0), bound
    assert circuits.decrypt_number(cnt, ks) == hd
    for bound, rb in hm:
        assert rb.decrypt(ks)[0] == (1 if hd > bound else 0), bound


def _sub_ref(a, b, bits):
    """The reference's bootsSUBNbit including its forced sign bit for a zero subtrahend
    (Math.cpp:137-138; DESIGN.md section 2, reference defects)."""
    tb = (((~b) & ((1 << bits) - 1)) + 1) & ((1 << bits) - 1) | (1 << bits)
    s = a + tb
    carry = (s >> (bits + 1)) & 1
    s &= (1 << (bits + 1)) - 1
    return s if carry else (-s) & ((1 << (bits + 1)) - 1)


def test_protocol_p1_driver(p128_keys):
    """Function_f then Function_g as main.cpp:533-586 strings them together;
    the decrypted y follows the reference's arithmetic (its |1 - 0| = 255 included), and the
    client is 'authenticated' exactly when y == r1."""
    from peba1_amd import protocol
    pp, ks, _ = p128_keys
    template = [12, 200, 77]
    for sample, bound in (([13, 201, 78], 256), ([90, 3, 250], 256), ([13, 201, 78], 2)):
        d = sum((a - b) ** 2 for a, b in zip(sample, template))
        out = protocol.run_p1(pp, ks, sample, template, bound, r0=17, r1=99)
        b = 1 if d > bound else 0
        want_y = ((((_sub_ref(1, b, 8) & 255) * 17) & 255) + ((b * 99) & 255)) & 255
        assert out["match_bit"] == b, (sample, bound)
        assert out["y"] == want_y, (sample, bound, out)
        assert out["authenticated"] == (want_y == 99)
        assert out["levels"]["function_f"] > 100 and out["levels"]["function_g"] > 10


def test_sharded_match_over_rccl_world1(p128_keys):
    """peba1_amd/dist.py on the GPU with the nccl (RCCL) backend at world size 1: exercises the
    device-pointer export/import of ciphertexts and the gather; 2 slots."""
    import torch
    import torch.distributed as dist
    from peba1_amd import api, circuits, lib
    from peba1_amd import dist as pd
    pp, ks, _ = p128_keys
    L = lib.load()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")      # one rank: the bootstrap needs no real interface
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            timeout=datetime.timedelta(seconds=120))
    try:
        L.tfhe_hip_set_encrypt_seed(47)
        tmpl_v, probe_v = [100, 3], [90, 7]
        d = sum((a - b) ** 2 for a, b in zip(probe_v, tmpl_v))
        T = circuits.EncryptedVector(pp, tmpl_v, 8, ks)
        S = circuits.EncryptedVector(pp, probe_v, 8, ks)
        bound = circuits.encrypt_number(pp, d - 1, 24, ks)
        api.set_deferred(True)
        res = pd.sharded_match(dist, torch, L, circuits.load(), pp.ptr, ks.cloud, pp.words,
                               [a.ptr for a in S.slots], [a.ptr for a in T.slots], bound.ptr, 8, device="cuda")
        api.set_deferred(False)
        res_ls = C.cast(res, lib.LS)
        assert L.bootsSymDecrypt(res_ls, ks.ptr) == 1               # d > d-1
        L.delete_gate_bootstrapping_ciphertext_array(24, res_ls)
        # the status word (round 4): every collective of libpeba1-dist first all-gathers one status word per rank over
        # RCCL, so a rank that failed locally makes every rank return -1 instead of leaving the others in the gather.
        # The match above went through it; here this rank reports a failure -- nobody enters the gather, the call returns
        # -1 naming the rank -- and the communicator then runs a match normally
        comm = pd.Comm(dist, torch, "cuda")
        try:
            comm.inject_failure(1)
            api.set_deferred(True)
            with pytest.raises(RuntimeError, match=r"rank 0 \(this rank\) failed before the gather: injected failure"):
                pd.sharded_match(dist, torch, L, circuits.load(), pp.ptr, ks.cloud, pp.words, [a.ptr for a in S.slots],
                                 [a.ptr for a in T.slots], bound.ptr, 8, device="cuda", comm=comm)
            res = pd.sharded_match(dist, torch, L, circuits.load(), pp.ptr, ks.cloud, pp.words, [a.ptr for a in S.slots],
                                   [a.ptr for a in T.slots], bound.ptr, 8, device="cuda", comm=comm, fast_combine=True)
            api.set_deferred(False)
            res_ls = C.cast(res, lib.LS)
            assert L.bootsSymDecrypt(res_ls, ks.ptr) == 1
            L.delete_gate_bootstrapping_ciphertext_array(24, res_ls)
            # identification through the C ABI with the gather to rank 0 (peba1_identify): 3 matches, 2 per flush
            tv = [[100, 3], [90, 7], [92, 9]]
            templates = [circuits.EncryptedVector(pp, t, 8, ks) for t in tv]
            from peba1_amd import identify
            all_bits = api.CiphertextArray(pp, 3)
            ib = circuits.encrypt_number(pp, 5, 24, ks)
            pd.broadcast_vector(comm, pp, ks, S, root=0)          # the probe's broadcast: export -> ncclBcast on the library's stream
            bits = identify.identify(pp, ks, S, templates, ib, 8, group=2, comm=comm, all_bits=all_bits)
            want = [1 if sum((a - b) ** 2 for a, b in zip(probe_v, t)) > 5 else 0 for t in tv]
            assert [int(b) for b in all_bits.decrypt(ks)] == want == [int(b) for b in bits.decrypt(ks)] and want == [1, 0, 1]
        finally:
            comm.close()
    finally:
        api.set_deferred(False)
        dist.destroy_process_group()


def test_decrypt_is_ordered_behind_a_stream_ordered_import(p128_keys):
    """tfhe_hip_import_samples_device_async only ENQUEUES the scatter into the slots -- behind whatever
    the caller put on tfhe_hip_stream() before it (libpeba1-dist: an ncclGather that waits for remote ranks).  A decrypt
    right after it used a blocking copy on the null stream, which the library's non-blocking stream does not order: it
    read the slot before the scatter had written it.  Here the words reach the import's source buffer only after ~50 ms
    of other work on the library's stream; the decrypts that follow at once must see them."""
    import numpy as np
    import torch
    from peba1_amd import api, lib
    pp, ks, _ = p128_keys
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(4711)
    n = 64
    bits = np.arange(n) % 3 == 0
    src = api.CiphertextArray(pp, n).encrypt(bits.astype(int), ks)
    words = torch.from_numpy(np.ascontiguousarray(src.words())).cuda()
    stream = torch.cuda.ExternalStream(L.tfhe_hip_stream())
    staging = torch.zeros_like(words)
    busy = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    dst = api.CiphertextArray(pp, n)
    with torch.cuda.stream(stream):
        for _ in range(40):                       # the "remote ranks": tens of milliseconds on the library's stream
            busy = busy @ busy * 1e-3
        staging.copy_(words)                      # ... and only then do the words exist
    rc = L.tfhe_hip_import_samples_device_async(dst.ptr, n, pp.ptr, C.c_void_p(staging.data_ptr()))
    assert rc == 0
    got = dst.decrypt(ks)                         # immediately: no tfhe_hip_wait, no synchronize
    assert list(got) == list(bits.astype(int))
    # the same through the host mirror
    dst2 = api.CiphertextArray(pp, n)
    staging.zero_()
    with torch.cuda.stream(stream):
        for _ in range(40):
            busy = busy @ busy * 1e-3
        staging.copy_(words)
    assert L.tfhe_hip_import_samples_device_async(dst2.ptr, n, pp.ptr, C.c_void_p(staging.data_ptr())) == 0
    assert L.tfhe_hip_sync_samples(dst2.ptr, n) == 0
    assert (dst2.words() == src.words()).all()
    assert L.tfhe_hip_wait() == 0 and L.tfhe_hip_stream_sync() == 0
    torch.cuda.synchronize()


def test_a_host_wait_past_its_deadline_ends_the_process_with_a_message():
    """with "sync_deadline_ms" set, a host wait on the library's stream that outlasts the deadline
    -- here a 2,048-gate launch (~20 ms) against a deadline of 1 ms, standing in for a collective whose peer never arrived
    -- prints what was waited for and by whom and exits with TFHE_HIP_EXIT_DEADLINE (86): non-zero, no retry.  In a child
    process: the exit is the point."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from peba1_amd import api, lib\n"
            "L = lib.load()\n"
            "pp = api.ParameterSet(128); ks = api.SecretKeySet(pp, 0x5EBA2, device=True)\n"
            "api.set_deferred(False)\n"
            "G = 2048\n"
            "a = api.CiphertextArray(pp, G).encrypt(np.ones(G, int), ks); b = api.CiphertextArray(pp, G).encrypt(np.ones(G, int), ks)\n"
            "r = api.CiphertextArray(pp, G)\n"
            "api.gate_batch('AND', r, a, b, ks)                       # no deadline: completes\n"
            "assert int(r.decrypt(ks)[0]) == 1\n"
            "L.tfhe_hip_set_diag_label(b'rank 3 of 8 (test)')\n"
            "api.set_tuning('sync_deadline_ms', 1)\n"
            "print('armed', flush=True)\n"
            "api.gate_batch('AND', r, a, b, ks)                       # ~20 ms of kernels behind a 1 ms deadline\n"
            "print('NOT REACHED', flush=True)\n" % root)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert run.returncode == 86, (run.returncode, run.stdout[-500:], run.stderr[-1500:])
    assert "armed" in run.stdout and "NOT REACHED" not in run.stdout
    assert "did not complete within 1 ms on rank 3 of 8 (test)" in run.stderr


@pytest.mark.parametrize("fixture,circuit", [("function_f_digest.json", "function_f"),
                                             ("function_f_fast_digest.json", "function_f_fast"),
                                             ("function_f_p2048_digest.json", "function_f")])
def test_function_f_ciphertexts_match_oracle_digest(p128_keys, fixture, circuit):
    """Whole-circuit ciphertext parity: a complete 2-slot Function_f (3,438 bootstrapped gates,
    incl. 24 XNOR + 48 MUX) on the GPU reproduces, bit for bit, the SHA-256 of the 24 output
    ciphertexts that the CPU oracle produced through the same circuit library
    (tests/golden/make_function_f_digest.py, ~10 CPU-minutes).  Same for the optimised DAG
    (542 blind rotations incl. ANDNY/ANDYN and MUX full adders), and for the same circuit under the
    N = 2048 parameter set of BASELINE configs[4] (the split kernel form; oracle: ~35 CPU-minutes)."""
    import hashlib
    import json
    from types import SimpleNamespace
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", fixture)) as f:
        g = json.load(f)
    assert g["key_seed"] == 0x5EBA2                      # the session keyset of conftest.py
    own_keys = None
    if g["params"] == "P2048":
        pp = api.ParameterSet(p2048=True)
        own_keys = ks = api.SecretKeySet(pp, g["key_seed"], device=True)
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(g["encrypt_seed"])
    bits = g["bits"]
    T, S = [], []
    for t, s in zip(g["template"], g["probe"]):          # encryption order is part of the fixture
        T.append(circuits.encrypt_number(pp, t, bits, ks))
        S.append(circuits.encrypt_number(pp, s, bits, ks))
    bound = circuits.encrypt_number(pp, g["bound"], 3 * bits, ks)
    rb = api.CiphertextArray(pp, 3 * bits)
    api.reset_stats()
    api.set_deferred(True)
    try:
        getattr(circuits, circuit)(rb, SimpleNamespace(slots=S), SimpleNamespace(slots=T), bound, bits, ks)
        api.flush()
    finally:
        api.set_deferred(False)
    st = api.stats()
    # the recorder shares the result of a gate recorded twice with the same operands (reuse_gates)
    # (and drops gates whose result nothing can observe: eliminate_dead); a shared or dropped gate is 1 or 2 rotations
    assert st["blind_rotates"] <= g["blind_rotates"] <= st["blind_rotates"] + 2 * (st["reused_gates"] + st["dead_gates"])
    words = rb.words()
    try:
        assert hashlib.sha256(words[0].tobytes()).hexdigest() == g["result_b0_sha256"]
        assert hashlib.sha256(words.tobytes()).hexdigest() == g["result_b_sha256"]
        assert rb.decrypt(ks)[0] == g["match_bit"]
    finally:
        if own_keys is not None:
            own_keys.close()


def test_function_f_128_slots_ciphertexts_match_oracle_digest(p128_keys):
    """BASELINE configs[1] at its own size, word for word (VERDICT r2 top item; SURVEY 8(d) cfg2: "memcmp vs CPU oracle on
    every output ciphertext"): the complete 128-slot x 8-bit Function_f of /root/reference/src/Math.cpp:379-387 -- 215,544
    blind rotations, the 6,276-wide first level, the 260-level ripple tail, 377 scheduled levels with slack moves -- on the
    inputs of SURVEY 8(c) (template (37 i + 11) mod 255, genuine probe = template + 1), against the encrypted bounds 256
    and 0, in the library's default mode.  All 24 output ciphertexts of both runs hash to what the CPU oracle produced
    through the same circuit library (tests/golden/make_function_f_digest.py --slots128: the recorded DAG evaluated level
    by level on host threads, about an hour of 7 cores)."""
    import hashlib
    import json
    from types import SimpleNamespace
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "function_f_128_digest.json")) as f:
        g = json.load(f)
    assert g["key_seed"] == 0x5EBA2 and g["nslots"] == 128 and g["bits"] == 8
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(g["encrypt_seed"])
    template = [(37 * i + 11) % 255 for i in range(g["nslots"])]
    probe = [t + 1 for t in template]
    assert sum((a - b) ** 2 for a, b in zip(probe, template)) == g["distance"]
    T, S = [], []
    for t, s in zip(template, probe):                    # encryption order is part of the fixture
        T.append(circuits.encrypt_number(pp, t, g["bits"], ks))
        S.append(circuits.encrypt_number(pp, s, g["bits"], ks))
    bounds = [circuits.encrypt_number(pp, b, 3 * g["bits"], ks) for b in g["bounds"]]
    api.set_deferred(True)                               # the library's default mode (earlier tests switch it off)
    try:
        for run, bound in zip(g["runs"], bounds):
            rb = api.CiphertextArray(pp, 3 * g["bits"])
            api.reset_stats()
            circuits.function_f(rb, SimpleNamespace(slots=S), SimpleNamespace(slots=T), bound, g["bits"], ks)
            words = rb.words()                           # runs the pending gates
            st = api.stats()
            assert st["levels"] in (376, 377)            # 377 recorded; the deepest level holds only results nobody reads
            # the recorder shares the result of a gate recorded twice with the same operands (a shared gate is 1 or 2 rotations)
            assert st["blind_rotates"] <= run["blind_rotates_recorded"] <= st["blind_rotates"] + 2 * (st["reused_gates"] + st["dead_gates"])
            assert hashlib.sha256(words[0].tobytes()).hexdigest() == run["result_b0_sha256"], run["bound"]
            assert hashlib.sha256(words.tobytes()).hexdigest() == run["result_b_sha256"], run["bound"]
            assert rb.decrypt(ks)[0] == run["match_bit"] == (1 if g["distance"] > run["bound"] else 0)
    finally:
        api.set_deferred(False)


@pytest.mark.parametrize("fixture", ["function_f_2_folded_digest.json", "function_f_128_folded_digest.json"])
def test_constant_folding_reproduces_the_oracles_folded_digest(p128_keys, fixture):
    """Round 6, opt-in tuning "fold_constants": a gate with a public constant operand (a trivial sample: bootsCONSTANT, a fresh
    sample, a copy of either) is answered without a bootstrap -- the constant, the other operand, or its negation; a MUX with a
    constant data operand becomes a two-input gate.  The reference's own Function_f (/root/reference/src/Math.cpp:379-387,
    unchanged gate sequence) then bootstraps 38 % of its gates: 82,499 of 215,544 at 128 slots.  The folded circuit's
    ciphertexts are NOT TFHE's words (TFHE bootstraps every gate) -- so the oracle's provider folds by the same rule
    (oracle/boots_oracle.c orc_boots_set_fold) and THIS test compares with its digest, word for word; the match bits are the
    unfolded run's.  Default off: every other test in the suite runs with TFHE's words."""
    import hashlib
    import json
    from types import SimpleNamespace
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", fixture)
    if not os.path.exists(path):
        pytest.fail(f"{fixture} is missing: tests/golden/make_function_f_digest.py --slots128 --fold [--nslots N]")
    with open(path) as f:
        g = json.load(f)
    assert g["key_seed"] == 0x5EBA2 and g["constant_folding"] is True
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(g["encrypt_seed"])
    template = [(37 * i + 11) % 255 for i in range(g["nslots"])]
    probe = [t + 1 for t in template]
    T, S = [], []
    for t, s in zip(template, probe):                    # encryption order is part of the fixture
        T.append(circuits.encrypt_number(pp, t, g["bits"], ks))
        S.append(circuits.encrypt_number(pp, s, g["bits"], ks))
    bounds = [circuits.encrypt_number(pp, b, 3 * g["bits"], ks) for b in g["bounds"]]
    api.set_deferred(True)
    api.set_tuning("fold_constants", 1)
    api.set_tuning("reuse_gates", 0)                     # so that the executed rotations are the oracle's recorded count exactly
    api.set_tuning("eliminate_dead", 0)
    try:
        for run, bound in zip(g["runs"], bounds):
            rb = api.CiphertextArray(pp, 3 * g["bits"])
            api.reset_stats()
            circuits.function_f(rb, SimpleNamespace(slots=S), SimpleNamespace(slots=T), bound, g["bits"], ks)
            words = rb.words()                           # runs the pending gates
            st = api.stats()
            assert st["blind_rotates"] == run["blind_rotates_recorded"] and st["folded_gates"] > st["blind_rotates"]
            assert hashlib.sha256(words[0].tobytes()).hexdigest() == run["result_b0_sha256"], run["bound"]
            assert hashlib.sha256(words.tobytes()).hexdigest() == run["result_b_sha256"], run["bound"]
            assert rb.decrypt(ks)[0] == run["match_bit"] == (1 if g["distance"] > run["bound"] else 0)
        # ... and with the library's defaults on top (gate sharing, dead-gate elimination): the same words from fewer rotations
        api.set_tuning("reuse_gates", 1)
        api.set_tuning("eliminate_dead", 1)
        rb = api.CiphertextArray(pp, 3 * g["bits"])
        api.reset_stats()
        circuits.function_f(rb, SimpleNamespace(slots=S), SimpleNamespace(slots=T), bounds[0], g["bits"], ks)
        assert hashlib.sha256(rb.words()[0].tobytes()).hexdigest() == g["runs"][0]["result_b0_sha256"]
        assert api.stats()["blind_rotates"] < g["runs"][0]["blind_rotates_recorded"]
    finally:
        api.set_tuning("fold_constants", 0)
        api.set_tuning("reuse_gates", 1)
        api.set_tuning("eliminate_dead", 1)
        api.set_deferred(False)


def test_constant_folding_truth_tables(p128_keys):
    """Every two-input gate with a constant on either side, NOT of a constant, and every MUX with constants among its operands,
    folded ("fold_constants" 1) against the same call bootstrapped (default): the same decrypted bit for every value of the
    non-constant operands -- and no blind rotation where the rule says none is needed."""
    import itertools
    from peba1_amd import api, lib
    pp, ks, _ = p128_keys
    L = lib.load()
    gates = ["NAND", "OR", "AND", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
    fn = {g: getattr(L, "boots" + g) for g in gates}
    tt = {"NAND": lambda a, b: 1 - (a & b), "OR": lambda a, b: a | b, "AND": lambda a, b: a & b, "NOR": lambda a, b: 1 - (a | b),
          "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b), "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
          "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b)}
    x = api.CiphertextArray(pp, 2).encrypt([0, 1], ks)               # the variable operand: an encrypted 0 and an encrypted 1
    k = api.CiphertextArray(pp, 2)
    L.bootsCONSTANT(k.at(0), 0, ks.cloud); L.bootsCONSTANT(k.at(1), 1, ks.cloud)
    api.set_deferred(True)
    api.set_tuning("fold_constants", 1)
    try:
        api.reset_stats()
        cases = []
        for g in gates:
            for xv, kv, const_first in itertools.product((0, 1), (0, 1), (False, True)):
                r = api.CiphertextArray(pp, 1)
                if const_first:
                    fn[g](r.at(0), k.at(kv), x.at(xv), ks.cloud); want = tt[g](kv, xv)
                else:
                    fn[g](r.at(0), x.at(xv), k.at(kv), ks.cloud); want = tt[g](xv, kv)
                cases.append((g, xv, kv, const_first, r, want))
        for kv in (0, 1):
            r = api.CiphertextArray(pp, 1)
            L.bootsNOT(r.at(0), k.at(kv), ks.cloud)
            cases.append(("NOT", None, kv, True, r, 1 - kv))
        # MUX(a, b, c) = a ? b : c with every operand a constant (0 / 1) or a variable (None)
        mux_rot = 0
        for sel, b_, c_ in itertools.product((0, 1, None), repeat=3):
            if sel is None and b_ is None and c_ is None:
                continue
            for va, vb, vc in itertools.product((0, 1), repeat=3):
                ops = [k.at(o) if o is not None else x.at(v) for o, v in ((sel, va), (b_, vb), (c_, vc))]
                a_val, b_val, c_val = (o if o is not None else v for o, v in ((sel, va), (b_, vb), (c_, vc)))
                r = api.CiphertextArray(pp, 1)
                L.bootsMUX(r.at(0), ops[0], ops[1], ops[2], ks.cloud)
                cases.append(("MUX", (sel, b_, c_), (va, vb, vc), None, r, b_val if a_val else c_val))
        api.flush()
        st = api.stats()
        # two-input gates and NOTs with a constant operand never rotate; a MUX does only as a two-input gate on two variables
        assert st["folded_gates"] == len(cases) - 2 and st["blind_rotates"] <= 4 * 8
        for name, p1, p2, p3, r, want in cases:
            assert int(r.decrypt(ks)[0]) == want, (name, p1, p2, p3)
    finally:
        api.set_tuning("fold_constants", 0)
        api.set_deferred(False)


def test_function_g_and_hamming_ciphertexts_match_oracle_digests(p128_keys):
    """the protocol's second function and the Hamming workload were checked at decrypt level only.
    peba1_function_g (Math.cpp:390-417, b = 1: selects r1; 2,874 blind rotations) and peba1_hamming_match on 16-bit words
    (both sides of the threshold) reproduce, word for word, the SHA-256 of the output ciphertexts the CPU oracle produced
    through the same circuit library (tests/golden/make_function_f_digest.py --small-circuits)."""
    import hashlib
    import json
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    L = lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "function_g_digest.json")) as f:
        g = json.load(f)
    assert g["key_seed"] == 0x5EBA2
    L.tfhe_hip_set_encrypt_seed(g["encrypt_seed"])
    eb = circuits.encrypt_number(pp, g["b"], g["bits"], ks)          # encryption order is part of the fixture
    e0 = circuits.encrypt_number(pp, g["r0"], g["bits"], ks)
    e1 = circuits.encrypt_number(pp, g["r1"], g["bits"], ks)
    res = api.CiphertextArray(pp, g["bits"])
    api.set_deferred(True)
    try:
        circuits.function_g(res, eb, e0, e1, g["bits"], ks)
        assert hashlib.sha256(res.words().tobytes()).hexdigest() == g["result_sha256"]
    finally:
        api.set_deferred(False)
    assert circuits.decrypt_number(res, ks) == g["value"] == g["r1"]

    with open(os.path.join(root, "tests", "golden", "hamming16_digest.json")) as f:
        h = json.load(f)
    L.tfhe_hip_set_encrypt_seed(h["encrypt_seed"])
    ea = circuits.encrypt_number(pp, h["a"], h["nbits"], ks)
    eb = circuits.encrypt_number(pp, h["b"], h["nbits"], ks)
    assert circuits.hamming_count_bits(h["nbits"]) == h["count_bits"]
    bounds = [circuits.encrypt_number(pp, r["bound"], h["count_bits"], ks) for r in h["runs"]]
    api.set_deferred(True)
    try:
        for run, bound in zip(h["runs"], bounds):
            rb = api.CiphertextArray(pp, h["count_bits"])
            circuits.hamming_match(rb, ea, eb, h["nbits"], bound, ks)
            assert hashlib.sha256(rb.words().tobytes()).hexdigest() == run["result_b_sha256"], run["bound"]
            assert rb.decrypt(ks)[0] == run["match_bit"] == (1 if h["distance"] > run["bound"] else 0)
    finally:
        api.set_deferred(False)


def test_arithmetic_building_blocks_ciphertexts_match_oracle_digests(p128_keys):
    """VERDICT r5 "missing 5": the reference's arithmetic helpers one by one at ciphertext level -- ADDN, TwoSComplement, ABS
    (both signs), SUBN, Multiply and the three shift helpers (/root/reference/src/Math.cpp:54-250; ABS and the shifts are not on
    the protocol's path, so no whole-circuit digest covers them) -- on the known-answer operands of SURVEY 8(c): SHA-256 of each
    result's ciphertext words == the oracle's (tests/golden/arith_helpers_digest.json, make_function_f_digest.py --arith),
    the decrypted values the plaintext ones, the executed blind rotations the oracle's count."""
    import hashlib
    import json
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    L, Lc = lib.load(), circuits.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "arith_helpers_digest.json")) as f:
        g = json.load(f)
    assert g["key_seed"] == 0x5EBA2
    bits = g["bits"]
    L.tfhe_hip_set_encrypt_seed(g["encrypt_seed"])
    E = {k: circuits.encrypt_number(pp, v, bits, ks) for k, v in g["operands"].items()}      # dict order = encryption order
    carry = api.CiphertextArray(pp, 1)
    calls = {
        "add_nbit(122, 204)": lambda r: Lc.peba1_add_nbit(r.ptr, E["a"].ptr, E["b"].ptr, carry.ptr, bits, ks.cloud),
        "twos_complement(5)": lambda r: Lc.peba1_twos_complement(r.ptr, E["c"].ptr, bits, ks.cloud),
        "abs(-100)": lambda r: Lc.peba1_abs(r.ptr, E["neg"].ptr, bits, ks.cloud),
        "abs(37)": lambda r: Lc.peba1_abs(r.ptr, E["pos"].ptr, bits, ks.cloud),
        "sub_nbit(122, 204)": lambda r: Lc.peba1_sub_nbit(r.ptr, E["a"].ptr, E["b"].ptr, bits, ks.cloud),
        "multiply(122, 204)": lambda r: Lc.peba1_multiply(r.ptr, E["a"].ptr, E["b"].ptr, bits, ks.cloud),
        "shift_left(0x35, 3)": lambda r: Lc.peba1_shift_left(r.ptr, E["s"].ptr, bits, 3, ks.cloud),
        "shift_right(0x35, 2)": lambda r: Lc.peba1_shift_right(r.ptr, E["s"].ptr, bits, 2, ks.cloud),
        "shift_left_inplace(0x35, 1)": lambda r: Lc.peba1_shift_left_inplace(r.ptr, bits, 1, ks.cloud),
    }
    assert [c["name"] for c in g["cases"]] == list(calls)
    api.set_deferred(True)
    api.set_tuning("reuse_gates", 0)          # the oracle's count is of every recorded gate
    api.set_tuning("eliminate_dead", 0)
    try:
        for c in g["cases"]:
            res = E["s"] if c["name"].startswith("shift_left_inplace") else api.CiphertextArray(pp, c["samples"])
            api.reset_stats()
            calls[c["name"]](res)
            words = res.words()                                                        # runs the pending gates
            assert hashlib.sha256(words.tobytes()).hexdigest() == c["sha256"], c["name"]
            assert circuits.decrypt_number(res, ks) == c["value"], c["name"]
            assert api.stats()["blind_rotates"] == c["blind_rotates"], c["name"]
            if "carry" in c:
                assert int(carry.decrypt(ks)[0]) == c["carry"]
    finally:
        api.set_tuning("reuse_gates", 1)
        api.set_tuning("eliminate_dead", 1)
        api.set_deferred(False)


def test_gate_reuse_is_transparent(p128_keys):
    """reuse_gates: a 3-slot Function_f with and without sharing of identical pending gates --
    fewer blind rotations, the same 24 output ciphertexts."""
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    out, rots = [], []
    try:
        for reuse, dead in ((1, 1), (0, 0), (0, 1)):
            api.set_tuning("reuse_gates", reuse)
            api.set_tuning("eliminate_dead", dead)
            lib.load().tfhe_hip_set_encrypt_seed(333)
            T = circuits.EncryptedVector(pp, [12, 200, 77], 8, ks)
            S = circuits.EncryptedVector(pp, [15, 190, 78], 8, ks)
            bound = circuits.encrypt_number(pp, 100, 24, ks)
            rb = api.CiphertextArray(pp, 24)
            api.reset_stats()
            api.set_deferred(True)
            try:
                circuits.function_f(rb, S, T, bound, 8, ks)
                api.flush()
            finally:
                api.set_deferred(False)
            st = api.stats()
            out.append(rb.words())
            rots.append((st["blind_rotates"], st["reused_gates"], st["dead_gates"]))
    finally:
        api.set_tuning("reuse_gates", 1)
        api.set_tuning("eliminate_dead", 1)
    assert (out[0] == out[1]).all() and (out[2] == out[1]).all()
    assert rots[1][1] == 0 and rots[1][2] == 0 and rots[0][1] > 0 and rots[0][0] < rots[1][0], rots
    # dead-gate elimination alone: the dropped carries of the reference's adders (5 gates per adder whose carry-out
    # nothing reads), nothing else changes
    assert rots[2][1] == 0 and rots[2][2] > 100 and 0 < rots[1][0] - rots[2][0] <= 2 * rots[2][2], rots
    assert rb.decrypt(ks)[0] == 1                                   # 9 + 100 + 1 > 100


def test_optimised_match_full_size(p128_keys):
    """peba1_function_f_fast on the full 128 x 8 bit match (SURVEY 8c inputs): genuine -> 0,
    impostor -> 1, as the plaintext rule and the reference's circuit give; about 7x fewer
    blind rotations and 5x fewer levels than the reference's DAG."""
    from peba1_amd import api, circuits, lib
    pp, ks, _ = p128_keys
    lib.load().tfhe_hip_set_encrypt_seed(128)
    tmpl = [(37 * i + 11) % 255 for i in range(128)]
    T = circuits.EncryptedVector(pp, tmpl, 8, ks)
    bound = circuits.encrypt_number(pp, 256, 24, ks)
    api.set_deferred(True)
    try:
        for probe, want in (([t + 1 for t in tmpl], 0), ([(91 * i + 5) % 256 for i in range(128)], 1)):
            S = circuits.EncryptedVector(pp, probe, 8, ks)
            rb = api.CiphertextArray(pp, 24)
            api.reset_stats()
            circuits.function_f_fast(rb, S, T, bound, 8, ks)
            levels = api.flush()
            st = api.stats()
            assert rb.decrypt(ks).tolist() == [want] + [0] * 23
            assert 20000 < st["blind_rotates"] <= 29536 and levels < 110, (st["blind_rotates"], levels)
        dist = api.CiphertextArray(pp, 24)
        circuits.euclidean_distance_fast(dist, S, T, 8, ks)
        assert circuits.decrypt_number(dist, ks) == 1400950          # SURVEY 8c known answer
    finally:
        api.set_deferred(False)


def test_reference_object_code_on_the_gpu_library():
    """The literal drop-in: the reference's own src/Math.cpp, compiled where it lies and LINKED
    against libtfhe-hip.so (oracle/Makefile `ref` -> oracle/_ref/refdriver_hip, built where
    /root/reference exists), runs the known-answer scenarios of SURVEY 8c on the GPU in the
    library's default execution mode (recording; results observed through bootsSymDecrypt): 8-bit adder / subtractor / multiplier, both 128-slot distances, Function_f at two
    bounds for genuine and impostor -- values equal to tests/golden/circuit_known_answers.json."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "refdriver_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/refdriver_hip not built (needs /root/reference at build time)")
    # no environment switch, no extra call: the binary is the reference's code as it stands
    env = {k: v for k, v in os.environ.items() if k != "TFHE_HIP_DEFERRED"}
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout)
    with open(os.path.join(root, "tests", "golden", "circuit_known_answers.json")) as f:
        want = json.load(f)
    assert set(got) == set(want)
    for name, w in want.items():
        assert got[name]["value"] == w["value"], (name, got[name]["value"], w["value"])
        # executed rotations: at most what the circuit records (identical pending gates are shared)
        assert got[name]["blind_rotates"] <= w["blind_rotates"], name
        assert (got[name]["blind_rotates"] > 0) == (w["blind_rotates"] > 0), name     # the shift helpers bootstrap nothing


def test_reference_program_runs_unmodified_on_the_gpu_library():
    """The whole reference program -- src/main.cpp, Math.cpp, Client.cpp, unmodified, compiled where
    they lie and linked against libtfhe-hip.so (oracle/Makefile `ref` -> oracle/_ref/tfhe_protocol_hip)
    -- runs on the GPU with no switch of any kind: its 5 x 128 per-operation checks, both
    128-slot distances and protocol P_1 (about 0.7 M bootstrapped gates) print, line for line, what
    the same sources print over the plaintext-bit provider on the CPU
    (tests/golden/reference_main_output.txt; duration lines dropped; time() fixed for both,
    tests/refcompat/fixed_time.c).  Includes the reference's own overflow (Function_g writes one
    sample past an array), which the library refuses instead of following."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "tfhe_protocol_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tfhe_protocol_hip not built (needs /root/reference at build time)")
    env = {k: v for k, v in os.environ.items() if k != "TFHE_HIP_DEFERRED"}
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2000:]
    keep = [l for l in out.stdout.splitlines() if not re.search(r"seconds|Function [fg]( bitwise)?: ", l)]
    with open(os.path.join(root, "tests", "golden", "reference_main_output.txt")) as f:
        want = f.read().splitlines()
    assert keep == want, next(((i, a, b) for i, (a, b) in enumerate(zip(keep, want)) if a != b), (len(keep), len(want)))
