"""tfhe_io.h entry points: parameter / key / ciphertext files round-trip
bit-exactly, a foreign file is refused.  No GPU: loaded keysets build their device image at
the first gate only."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

from peba1_amd import api


@pytest.fixture(scope="module")
def small_key():
    params = api.ParameterSet(custom=(32, 1024, 1, 2, 8, 4, 2, 2.0 ** -30, 2.0 ** -40, 2.0 ** -6))
    return params, api.SecretKeySet(params, 0xF11E, device=False)


def test_parameter_set_roundtrip(tmp_path):
    for ps in (api.ParameterSet(128), api.ParameterSet(80), api.ParameterSet(p2048=True)):
        ps.save(tmp_path / "p.bin")
        q = api.ParameterSet.load(tmp_path / "p.bin")
        assert (q.n, q.N, q.k, q.l, q.Bgbit, q.ks_t, q.ks_basebit) == (ps.n, ps.N, ps.k, ps.l, ps.Bgbit, ps.ks_t, ps.ks_basebit)
        a, b = ps.ptr.contents, q.ptr.contents
        assert a.in_out_params.contents.alpha_min == b.in_out_params.contents.alpha_min
        assert a.tgsw_params.contents.tlwe_params.contents.alpha_min == b.tgsw_params.contents.tlwe_params.contents.alpha_min


def test_secret_and_cloud_keyset_roundtrip(small_key, tmp_path):
    params, key = small_key
    key.save(tmp_path / "secret.key")
    key.save_cloud(tmp_path / "cloud.key")
    sk = api.SecretKeySet.load(tmp_path / "secret.key")
    ck = api.CloudKeySet.load(tmp_path / "cloud.key")
    assert np.array_equal(sk.lwe_key(), key.lwe_key())
    assert np.array_equal(sk.tlwe_key(), key.tlwe_key())
    for k2 in (sk, ck):
        assert np.array_equal(k2.bk(), key.bk())
        assert np.array_equal(k2.ksk(), key.ksk())
        assert k2.params.n == params.n and k2.params.N == params.N
    # the cloud file is the secret file minus the two secret vectors
    assert (tmp_path / "secret.key").stat().st_size - (tmp_path / "cloud.key").stat().st_size == 4 * (params.n + params.N)
    with pytest.raises(TypeError):
        ck.lwe_key()
    # a ciphertext made under the original key decrypts under the reloaded one
    ct = api.CiphertextArray(params, 8).encrypt([1, 0, 1, 1, 0, 0, 1, 0], key)
    assert ct.decrypt(sk).tolist() == [1, 0, 1, 1, 0, 0, 1, 0]
    ck.save(tmp_path / "cloud2.key")
    assert (tmp_path / "cloud2.key").read_bytes() == (tmp_path / "cloud.key").read_bytes()
    sk.close()
    ck.close()


def test_ciphertext_roundtrip(small_key, tmp_path):
    params, key = small_key
    bits = [1, 0, 0, 1, 1]
    ct = api.CiphertextArray(params, len(bits)).encrypt(bits, key)
    ct.save(tmp_path / "ct.bin")
    assert (tmp_path / "ct.bin").stat().st_size == len(bits) * (24 + 4 * params.words)
    back = api.CiphertextArray(params, len(bits)).load(tmp_path / "ct.bin")
    assert np.array_equal(back.words(), ct.words())
    assert back.decrypt(key).tolist() == bits


def test_foreign_or_truncated_file_is_refused(small_key, tmp_path):
    """Malformed files are refused through the error channel (loaders return NULL, the Python
    mirror raises with tfhe_hip_last_error()); the process is NOT aborted."""
    from peba1_amd import api
    params, key = small_key
    key.save_cloud(tmp_path / "cloud.key")
    blob = (tmp_path / "cloud.key").read_bytes()
    (tmp_path / "foreign.key").write_bytes(b"\x00" * 64)
    (tmp_path / "short.key").write_bytes(blob[: len(blob) // 2])
    (tmp_path / "params.bin").write_bytes(blob)      # a cloud key where a parameter set is expected
    for name, loader, needle in (("foreign.key", api.CloudKeySet, "not a libtfhe-hip file"),
                                 ("short.key", api.CloudKeySet, "short read"),
                                 ("short.key", api.SecretKeySet, "expected 3"),
                                 ("params.bin", api.ParameterSet, "expected 1")):
        with pytest.raises(ValueError, match=needle):
            loader.load(tmp_path / name)
    # still alive and working afterwards
    back = api.CloudKeySet.load(tmp_path / "cloud.key")
    assert back.params.n == params.n
    back.close()


def test_hostile_headers_fail_fast_and_unsupported_shapes_are_refused(small_key, tmp_path):
    """(1) a 100-byte file whose parameter record promises gigabytes of key is refused before anything is
    allocated (payload size vs the record, then bounded chunked reads); (2) a keyset whose shape the engine cannot
    evaluate (N = 4096, or a gadget outside every kernel form) is refused by the loader with the reason, instead of
    loading and aborting at the first gate; (3) nothing terminates the process."""
    import struct
    import time
    params, key = small_key
    key.save_cloud(tmp_path / "cloud.key")
    blob = (tmp_path / "cloud.key").read_bytes()
    hdr, rec = blob[:24], blob[24:24 + 56]
    n, N, k, l, Bgbit, ks_t, ks_basebit, pad = struct.unpack("<8i", rec[:32])
    assert (n, N, l, Bgbit) == (params.n, params.N, params.l, params.Bgbit)

    def record(**kw):
        v = dict(n=n, N=N, k=k, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_basebit=ks_basebit, pad=0)
        v.update(kw)
        return struct.pack("<8i", *[v[f] for f in ("n", "N", "k", "l", "Bgbit", "ks_t", "ks_basebit", "pad")]) + rec[32:]

    cases = {
        # the engine's largest shapes, header untouched: payload size disagrees with the record
        "huge.key": (hdr + record(n=1024, N=2048, l=4, Bgbit=8) + b"\0" * 20, "payload size does not match"),
        # header forged to agree: the short read is noticed after at most one 16 MB chunk
        "forged.key": (hdr[:16] + struct.pack("<Q", 56 + 4 * (1024 * 8 * 2 * 2048 + 2048 * ks_t * 4 * 1025))
                       + record(n=1024, N=2048, l=4, Bgbit=8) + b"\0" * 20, "short read"),
        "ring4096.key": (hdr + record(N=4096) + blob[80:200], "unsupported parameter set in file"),
        "gadget.key": (hdr + record(N=2048, l=8, Bgbit=4) + blob[80:200], "every blind-rotate kernel form"),
    }
    for name, (data, needle) in cases.items():
        (tmp_path / name).write_bytes(data)
        t = time.time()
        with pytest.raises(ValueError, match=needle):
            api.CloudKeySet.load(tmp_path / name)
        assert time.time() - t < 5.0, name
    back = api.CloudKeySet.load(tmp_path / "cloud.key")                  # still alive and working
    assert back.params.n == params.n
    back.close()
