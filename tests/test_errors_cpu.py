"""CPU tests (no GPU) of the library's error channel: conditions a caller can recover from are
reported through tfhe_hip_last_error() and leave the call without effect -- they must not abort
the host process (the tests would die with SIGABRT if they did).  The device-side cases (slot
pool exhaustion, a sample used with a key of another LWE dimension) are in tests/test_gpu_errors.py.
Also: default randomness is not a fixed constant."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def L():
    from peba1_amd import lib
    lb = lib.load()
    lb.tfhe_hip_clear_error()
    return lb


def _err(L):
    return L.tfhe_hip_last_error().decode()


def _foreign_sample(n):
    """An LweSample this library did not allocate (what a caller linking a stale upstream object,
    or passing stack memory, would hand in): plain memory, zero words in front of `a`."""
    from peba1_amd import lib
    buf = np.zeros(n + 16, dtype=np.int32)
    s = lib.LweSample()
    s.a = C.cast(buf.ctypes.data + 8 * 4, C.POINTER(C.c_int32))
    s.b = 0
    s.slot = 5            # claims a device slot: forces the header lookup
    return s, buf


def test_foreign_sample_is_refused_not_fatal(L):
    from peba1_amd import api
    pp = api.ParameterSet(128)
    s, keep = _foreign_sample(pp.n)
    out = np.zeros(pp.words, dtype=np.int32)
    rc = L.tfhe_hip_export_samples(C.byref(s), 1, pp.ptr, out.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == -1 and "not allocated by new_gate_bootstrapping_ciphertext_array" in _err(L)
    L.tfhe_hip_clear_error()
    rc = L.tfhe_hip_import_samples(C.byref(s), 1, pp.ptr, out.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == -1 and "not allocated" in _err(L)
    assert s.slot == 5 and not keep.any()                    # untouched
    L.tfhe_hip_clear_error()
    assert L.tfhe_hip_sync_samples(C.byref(s), 1) == -1 and "not allocated" in _err(L)
    # decrypting it: no abort, an error, bit 0
    ks = api.SecretKeySet(pp, 1, device=False)
    L.tfhe_hip_clear_error()
    assert L.bootsSymDecrypt(C.byref(s), ks.ptr) == 0 and "not allocated" in _err(L)
    ks.close()


def test_delete_of_a_foreign_array_is_reported(L):
    from peba1_amd import lib
    raw = np.zeros(64, dtype=np.int32)
    p = C.cast(raw.ctypes.data + 64, lib.LS)
    L.delete_gate_bootstrapping_ciphertext_array(1, p)
    assert "not an array base pointer" in _err(L)


def test_null_key_is_refused(L):
    from peba1_amd import api
    pp = api.ParameterSet(128)
    a = api.CiphertextArray(pp, 3)
    L.bootsAND(a.at(0), a.at(1), a.at(2), None)
    assert "null cloud key" in _err(L)
    L.tfhe_hip_clear_error()
    L.bootsMUX(a.at(0), a.at(1), a.at(2), a.at(2), None)
    assert "null cloud key" in _err(L)
    L.tfhe_hip_clear_error()
    L.bootsNOT(a.at(0), a.at(1), None)
    assert "null cloud key" in _err(L)
    assert L.tfhe_hip_gate_batch(2, a.ptr, a.ptr, a.ptr, 1, None) == -1


def test_foreign_and_truncated_files_return_null(L, tmp_path):
    from peba1_amd import api
    pp = api.ParameterSet(128)
    # an upstream-looking text file, a truncated container, a container of the wrong kind
    upstream = tmp_path / "upstream.key"
    upstream.write_bytes(b"-----BEGIN LWEPARAMS-----\nn: 630\n" + b"\0" * 64)
    good = tmp_path / "params.bin"
    pp.save(good)
    cut = tmp_path / "cut.bin"
    cut.write_bytes(good.read_bytes()[:10])       # inside the 24-byte header
    for path, what in ((upstream, "not a libtfhe-hip file"), (cut, "short read")):
        for loader in (L.new_tfheGateBootstrappingParameterSet_fromFile, L.new_tfheGateBootstrappingCloudKeySet_fromFile,
                       L.new_tfheGateBootstrappingSecretKeySet_fromFile):
            L.tfhe_hip_clear_error()
            with api._CFile(path, "rb") as fp:
                got = loader(fp)
            assert not got and what in _err(L), (path, _err(L))
    L.tfhe_hip_clear_error()
    with api._CFile(good, "rb") as fp:
        assert not L.new_tfheGateBootstrappingCloudKeySet_fromFile(fp)
    assert "object kind" in _err(L)
    # a ciphertext import from a foreign file leaves the sample untouched
    a = api.CiphertextArray(pp, 1)
    ks = api.SecretKeySet(pp, 3, device=False)
    a.encrypt([1], ks)
    before = a.words().copy()
    L.tfhe_hip_clear_error()
    with api._CFile(upstream, "rb") as fp:
        L.import_gate_bootstrapping_ciphertext_fromFile(fp, a.at(0), pp.ptr)
    assert "not a libtfhe-hip file" in _err(L) and (a.words() == before).all()
    ks.close()


def test_import_export_follow_the_parameter_set_not_the_last_key(L):
    """import/export used the pool of the last key that ran a gate.  Without a device
    the host path must already follow `params`: two parameter sets interleaved, word counts kept."""
    from peba1_amd import api
    p128, p80 = api.ParameterSet(128), api.ParameterSet(80)
    assert p128.words == 631 and p80.words == 501
    k128, k80 = api.SecretKeySet(p128, 5, device=False), api.SecretKeySet(p80, 6, device=False)
    a, b = api.CiphertextArray(p128, 2).encrypt([1, 0], k128), api.CiphertextArray(p80, 2).encrypt([0, 1], k80)
    wa, wb = a.words(), b.words()
    b2 = api.CiphertextArray(p80, 2).set_words(wb)
    a2 = api.CiphertextArray(p128, 2).set_words(wa)
    assert (a2.words() == wa).all() and (b2.words() == wb).all()
    assert list(a2.decrypt(k128)) == [1, 0] and list(b2.decrypt(k80)) == [0, 1]
    # fresh samples export as the trivial encryption of 0
    z = api.CiphertextArray(p80, 1).words()
    assert not z[0, :-1].any() and z[0, -1] == -(1 << 29)
    k128.close(); k80.close()


def test_default_randomness_is_not_a_constant(L):
    """the drop-in keygen / encryption must not repeat across processes.  Two host-side
    seeds from the OS, and two encryptions of the same bit after re-seeding from the OS, differ;
    tfhe_hip_set_encrypt_seed still reproduces."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from peba1_amd import api\n"
            "pp = api.ParameterSet(128); ks = api.SecretKeySet(pp, 9, device=False)\n"
            "a = api.CiphertextArray(pp, 1).encrypt([1], ks)\n"
            "print(int(a.words()[0, 0]), int(a.words()[0, 1]))\n" % ROOT)
    outs = {subprocess.check_output([sys.executable, "-c", code]).decode().strip() for _ in range(2)}
    assert len(outs) == 2, "two processes drew the same encryption mask without a fixed seed"
    from peba1_amd import api
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, 9, device=False)
    L.tfhe_hip_set_encrypt_seed(123)
    w1 = api.CiphertextArray(pp, 1).encrypt([1], ks).words()
    L.tfhe_hip_set_encrypt_seed(123)
    w2 = api.CiphertextArray(pp, 1).encrypt([1], ks).words()
    assert (w1 == w2).all()
    ks.close()


def test_kernel_form_admissibility_predicate():
    """every blind-rotate kernel form has an explicit admissible (l, Bgbit) range derived from its own
    lazy-arithmetic bounds (peba1_amd/csrc/br_forms.hpp).  Built-in sets: every form of the set's ring (N = 2048 has
    the split form only).  Custom gadgets: l = 4 / Bg = 2^8 stays inside all four forms at N = 1024; N = 2048 / l = 6 /
    Bgbit = 4 only inside the split form without its eleven-table first step; l = 8 at N = 2048 inside none;
    N = 1024 / l = 8 / Bg = 2^4 only inside the split and 2-wave forms; l = 9 / Bg = 2^3 only inside the 2-wave form
    (why that form stays in the library)."""
    from peba1_amd import lib
    ok = lib.load().tfhe_hip_test_form_admissible
    WIDE4, SPLIT, WAVE8, WAVE2 = range(4)
    for N, l, B in ((1024, 3, 7), (1024, 2, 10)):
        assert all(ok(f, N, l, B, t) for f in range(4) for t in range(3))
    assert [ok(f, 2048, 3, 6, 1) for f in range(4)] == [0, 1, 0, 0]
    assert [ok(f, 1024, 4, 8, 1) for f in range(4)] == [1, 1, 1, 1]
    assert [ok(SPLIT, 2048, 6, 4, t) for t in range(3)] == [1, 0, 1]
    assert not any(ok(f, 2048, 8, 4, t) for f in range(4) for t in range(3))
    assert not ok(WIDE4, 1024, 8, 4, 0) and ok(WAVE2, 1024, 8, 4, 0) and ok(SPLIT, 1024, 8, 4, 0)
    assert [ok(f, 1024, 9, 3, 0) for f in range(4)] == [0, 0, 0, 1]
    assert not ok(4, 1024, 3, 7, 0)                                             # no fifth form
    assert not ok(WIDE4, 4096, 3, 7, 0) and not ok(WIDE4, 1024, 5, 7, 0)          # ring size / l * Bgbit > 32
