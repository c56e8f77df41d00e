"""Thin Python mirror of the tfhe gate API as served by libtfhe-hip.so.

Names follow the reference's vocabulary (parameter set, secret/cloud keyset,
LweSample arrays, boots* gates; /root/reference/src/Math.cpp, src/main.cpp).
Everything here is plumbing over the C ABI; the arithmetic is in the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import lib as _l

GATE_CODES = {"NAND": 0, "OR": 1, "AND": 2, "NOR": 3, "XOR": 4, "XNOR": 5,
              "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9}


def _i32p(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_l.I32P)


def last_error():
    return _l.load().tfhe_hip_last_error().decode()


class _CFile:
    """A C FILE* for the tfhe_io.h entry points."""
    _libc = None

    def __init__(self, path, mode):
        if _CFile._libc is None:
            libc = C.CDLL(None)
            libc.fopen.restype = C.c_void_p
            libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
            libc.fclose.argtypes = [C.c_void_p]
            _CFile._libc = libc
        self.fp = _CFile._libc.fopen(str(path).encode(), mode.encode())
        if not self.fp:
            raise OSError("cannot open %s" % path)

    def __enter__(self):
        return self.fp

    def __exit__(self, *exc):
        _CFile._libc.fclose(self.fp)
        self.fp = None


class ParameterSet:
    """new_default_gate_bootstrapping_parameters (main.cpp:21) or an explicit tuple."""

    def __init__(self, minimum_lambda=128, custom=None, p2048=False, _ptr=None):
        L = _l.load()
        if _ptr is not None:
            self.ptr = _ptr
        elif custom is not None:
            self.ptr = L.tfhe_hip_new_parameters(*custom)
        elif p2048:
            self.ptr = L.tfhe_hip_new_p2048_parameters()
        else:
            self.ptr = L.new_default_gate_bootstrapping_parameters(minimum_lambda)
        if not self.ptr:
            raise ValueError("parameter set rejected: " + last_error())
        self.n = self.ptr.contents.in_out_params.contents.n
        tg = self.ptr.contents.tgsw_params.contents
        self.N = tg.tlwe_params.contents.N
        self.k = tg.tlwe_params.contents.k
        self.l, self.Bgbit = tg.l, tg.Bgbit
        self.ks_t, self.ks_basebit = self.ptr.contents.ks_t, self.ptr.contents.ks_basebit
        self.words = self.n + 1

    def save(self, path):
        with _CFile(path, "wb") as fp:
            _l.load().export_tfheGateBootstrappingParameterSet_toFile(fp, self.ptr)

    @classmethod
    def load(cls, path):
        with _CFile(path, "rb") as fp:
            ptr = _l.load().new_tfheGateBootstrappingParameterSet_fromFile(fp)
        if not ptr:
            raise ValueError("cannot load a parameter set from %s: %s" % (path, last_error()))
        return cls(_ptr=ptr)


class SecretKeySet:
    """new_random_gate_bootstrapping_secret_keyset (main.cpp:22) with an explicit seed."""

    def __init__(self, params, seed, device=True, _ptr=None):
        L = _l.load()
        self.params = params
        if _ptr is not None:
            self.ptr = _ptr
        else:
            f = L.tfhe_hip_new_secret_keyset_seeded if device else L.tfhe_hip_new_secret_keyset_seeded_host
            self.ptr = f(params.ptr, seed)
        if not self.ptr:
            raise RuntimeError("keygen failed: " + last_error())
        self.cloud = C.pointer(self.ptr.contents.cloud)   # &key->cloud, main.cpp:23

    def save(self, path):
        """export_tfheGateBootstrappingSecretKeySet_toFile (tfhe_io.h)."""
        with _CFile(path, "wb") as fp:
            _l.load().export_tfheGateBootstrappingSecretKeySet_toFile(fp, self.ptr)

    def save_cloud(self, path):
        """export_tfheGateBootstrappingCloudKeySet_toFile of &key->cloud: what the server gets."""
        with _CFile(path, "wb") as fp:
            _l.load().export_tfheGateBootstrappingCloudKeySet_toFile(fp, self.cloud)

    @classmethod
    def load(cls, path):
        with _CFile(path, "rb") as fp:
            ptr = _l.load().new_tfheGateBootstrappingSecretKeySet_fromFile(fp)
        if not ptr:
            raise ValueError("cannot load a secret keyset from %s: %s" % (path, last_error()))
        return cls(ParameterSet(_ptr=C.cast(ptr.contents.params, _l.PS)), None, _ptr=ptr)

    def close(self):
        if self.ptr:
            _l.load().delete_gate_bootstrapping_secret_keyset(self.ptr)
            self.ptr = None

    def _arr(self, f, owner):
        cnt = C.c_int64()
        p = f(owner, C.byref(cnt))
        return np.ctypeslib.as_array(p, shape=(cnt.value,))

    def lwe_key(self):
        return self._arr(_l.load().tfhe_hip_key_lwe, self.ptr)

    def tlwe_key(self):
        return self._arr(_l.load().tfhe_hip_key_tlwe, self.ptr)

    def bk(self):
        return self._arr(_l.load().tfhe_hip_key_bk, self.cloud)

    def ksk(self):
        return self._arr(_l.load().tfhe_hip_key_ksk, self.cloud)


class CloudKeySet(SecretKeySet):
    """A cloud keyset on its own (new_tfheGateBootstrappingCloudKeySet_fromFile): evaluates
    gates, cannot encrypt or decrypt."""

    def __init__(self, ptr):
        self.ptr = None
        self.cloud = ptr
        self.params = ParameterSet(_ptr=C.cast(ptr.contents.params, _l.PS))

    @classmethod
    def load(cls, path):
        with _CFile(path, "rb") as fp:
            ptr = _l.load().new_tfheGateBootstrappingCloudKeySet_fromFile(fp)
        if not ptr:
            raise ValueError("cannot load a cloud keyset from %s: %s" % (path, last_error()))
        return cls(ptr)

    def save(self, path):
        with _CFile(path, "wb") as fp:
            _l.load().export_tfheGateBootstrappingCloudKeySet_toFile(fp, self.cloud)

    save_cloud = save

    def close(self):
        if self.cloud:
            _l.load().delete_gate_bootstrapping_cloud_keyset(self.cloud)
            self.cloud = None

    def lwe_key(self):
        raise TypeError("a cloud keyset holds no secret key")

    tlwe_key = lwe_key


class CiphertextArray:
    """new_gate_bootstrapping_ciphertext_array / delete_... (Math.cpp:28-30,47-49)."""

    def __init__(self, params, count):
        self.params, self.count = params, count
        self.ptr = _l.load().new_gate_bootstrapping_ciphertext_array(count, params.ptr)

    def close(self):
        if self.ptr:
            _l.load().delete_gate_bootstrapping_ciphertext_array(self.count, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def at(self, i):
        return C.cast(C.addressof(self.ptr.contents) + i * C.sizeof(_l.LweSample), _l.LS)

    def encrypt(self, bits, key):
        L = _l.load()
        for i, b in enumerate(bits):
            L.bootsSymEncrypt(self.at(i), int(b), key.ptr)
        return self

    def decrypt(self, key):
        L = _l.load()
        return np.array([L.bootsSymDecrypt(self.at(i), key.ptr) for i in range(self.count)], dtype=np.int32)

    def save(self, path):
        """count x export_gate_bootstrapping_ciphertext_toFile into one file."""
        L = _l.load()
        with _CFile(path, "wb") as fp:
            for i in range(self.count):
                L.export_gate_bootstrapping_ciphertext_toFile(fp, self.at(i), self.params.ptr)

    def load(self, path):
        L = _l.load()
        with _CFile(path, "rb") as fp:
            for i in range(self.count):
                L.tfhe_hip_clear_error()
                L.import_gate_bootstrapping_ciphertext_fromFile(fp, self.at(i), self.params.ptr)
                if last_error():
                    raise ValueError("cannot load ciphertext %d from %s: %s" % (i, path, last_error()))
        return self

    def words(self):
        out = np.zeros((self.count, self.params.words), dtype=np.int32)
        rc = _l.load().tfhe_hip_export_samples(self.ptr, self.count, self.params.ptr, _i32p(out))
        if rc != 0:
            raise RuntimeError(last_error())
        return out

    def set_words(self, w):
        w = np.ascontiguousarray(w, dtype=np.int32).reshape(self.count, self.params.words)
        rc = _l.load().tfhe_hip_import_samples(self.ptr, self.count, self.params.ptr, _i32p(w))
        if rc != 0:
            raise RuntimeError(last_error())
        return self


def gate_batch(name, result, a, b, key):
    rc = _l.load().tfhe_hip_gate_batch(GATE_CODES[name], result.ptr, a.ptr, b.ptr, result.count, key.cloud)
    if rc != 0:
        raise RuntimeError(last_error())


def set_deferred(on):
    _l.load().tfhe_hip_set_deferred(1 if on else 0)


def get_deferred():
    return bool(_l.load().tfhe_hip_get_deferred())


def flush():
    return _l.load().tfhe_hip_flush()


def flush_async():
    """Enqueue the pending gates and return while the device works (tfhe_hip_flush_async); wait() completes it."""
    return _l.load().tfhe_hip_flush_async()


def wait():
    return _l.load().tfhe_hip_wait()


def set_tuning(name, value):
    if _l.load().tfhe_hip_set_tuning(name.encode(), int(value)) != 0:
        raise ValueError(last_error())


def stats():
    s = _l.Stats()
    _l.load().tfhe_hip_get_stats(C.byref(s))
    return {f: getattr(s, f) for f, _ in s._fields_}


def reset_stats():
    _l.load().tfhe_hip_reset_stats()


def kernel_negacyclic(key, ip, tp):
    ip = np.ascontiguousarray(ip, dtype=np.int32)
    tp = np.ascontiguousarray(tp, dtype=np.int32)
    res = np.zeros_like(tp)
    rc = _l.load().tfhe_hip_kernel_negacyclic(key.cloud, _i32p(ip), _i32p(tp), _i32p(res), ip.shape[0])
    if rc != 0:
        raise RuntimeError(last_error())
    return res


def kernel_bootstrap_woks(key, lin, want_acc=False):
    p = key.params
    lin = np.ascontiguousarray(lin, dtype=np.int32).reshape(-1, p.words)
    u = np.zeros((lin.shape[0], p.k * p.N + 1), dtype=np.int32)
    acc = np.zeros((lin.shape[0], (p.k + 1) * p.N), dtype=np.int32) if want_acc else None
    rc = _l.load().tfhe_hip_kernel_bootstrap_woks(key.cloud, _i32p(lin), lin.shape[0], _i32p(u),
                                                  _i32p(acc) if want_acc else None)
    if rc != 0:
        raise RuntimeError(last_error())
    return (u, acc) if want_acc else u


def kernel_keyswitch(key, u):
    p = key.params
    u = np.ascontiguousarray(u, dtype=np.int32).reshape(-1, p.k * p.N + 1)
    out = np.zeros((u.shape[0], p.words), dtype=np.int32)
    rc = _l.load().tfhe_hip_kernel_keyswitch(key.cloud, _i32p(u), u.shape[0], _i32p(out))
    if rc != 0:
        raise RuntimeError(last_error())
    return out
