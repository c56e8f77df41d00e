"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/tfhe_oracle.h for the parity status: "parity unpinned"
at ciphertext level, pinned by truth tables and SURVEY.md 8c known answers).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GATES = {"NAND": 0, "OR": 1, "AND": 2, "NOR": 3, "XOR": 4, "XNOR": 5,
         "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9}


class OrcParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("k", C.c_int32), ("l", C.c_int32),
                ("Bgbit", C.c_int32), ("ks_t", C.c_int32), ("ks_basebit", C.c_int32),
                ("ks_stdev", C.c_double), ("bk_stdev", C.c_double), ("max_stdev", C.c_double)]


class OrcRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


class OrcKeySet(C.Structure):
    _fields_ = [("p", OrcParams), ("lwe_key", C.POINTER(C.c_int32)), ("tlwe_key", C.POINTER(C.c_int32)),
                ("bk", C.POINTER(C.c_int32)), ("ksk", C.POINTER(C.c_int32)),
                ("bk_ntt", C.POINTER(C.c_uint64))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        i32p = C.POINTER(C.c_int32)
        L.orc_params_default.argtypes = [C.POINTER(OrcParams), C.c_int32]
        L.orc_params_p2048.argtypes = [C.POINTER(OrcParams)]
        L.orc_bk_words.restype = C.c_size_t
        L.orc_bk_words.argtypes = [C.POINTER(OrcParams)]
        L.orc_ksk_words.restype = C.c_size_t
        L.orc_ksk_words.argtypes = [C.POINTER(OrcParams)]
        L.orc_rng_seed.argtypes = [C.POINTER(OrcRng), C.c_uint64]
        L.orc_rng_next.restype = C.c_uint64
        L.orc_rng_next.argtypes = [C.POINTER(OrcRng)]
        L.orc_rng_gauss.restype = C.c_double
        L.orc_rng_gauss.argtypes = [C.POINTER(OrcRng), C.c_double]
        L.orc_keygen.restype = C.POINTER(OrcKeySet)
        L.orc_keygen.argtypes = [C.POINTER(OrcParams), C.c_uint64]
        L.orc_keyset_free.argtypes = [C.POINTER(OrcKeySet)]
        L.orc_encrypt_bit.argtypes = [C.POINTER(OrcKeySet), C.POINTER(OrcRng), C.c_int32, i32p]
        L.orc_phase.restype = C.c_int32
        L.orc_phase.argtypes = [C.POINTER(OrcKeySet), i32p]
        L.orc_decrypt_bit.restype = C.c_int32
        L.orc_decrypt_bit.argtypes = [C.POINTER(OrcKeySet), i32p]
        L.orc_modswitch.restype = C.c_int32
        L.orc_modswitch.argtypes = [C.c_int32, C.c_int32]
        L.orc_modswitch_to_torus.restype = C.c_int32
        L.orc_modswitch_to_torus.argtypes = [C.c_int32, C.c_int32]
        L.orc_negacyclic_schoolbook.argtypes = [i32p, i32p, i32p, C.c_int32]
        L.orc_negacyclic_ntt.argtypes = [i32p, i32p, i32p, C.c_int32]
        L.orc_decompose.argtypes = [i32p, i32p, C.POINTER(OrcParams)]
        L.orc_cmux_rotate.argtypes = [C.POINTER(OrcKeySet), C.c_int32, C.c_int32, i32p, C.c_int]
        L.orc_blind_rotate.argtypes = [C.POINTER(OrcKeySet), i32p, C.c_int32, C.c_int32, i32p, C.c_int]
        L.orc_sample_extract.argtypes = [C.POINTER(OrcParams), i32p, i32p]
        L.orc_keyswitch.argtypes = [C.POINTER(OrcKeySet), i32p, i32p]
        L.orc_bootstrap_woks.argtypes = [C.POINTER(OrcKeySet), i32p, C.c_int32, i32p, C.c_int]
        L.orc_gate_prelude.argtypes = [C.POINTER(OrcParams), C.c_int, i32p, i32p, i32p]
        L.orc_gate2.argtypes = [C.POINTER(OrcKeySet), C.c_int, i32p, i32p, i32p, C.c_int]
        L.orc_mux.argtypes = [C.POINTER(OrcKeySet), i32p, i32p, i32p, i32p, C.c_int]
        L.orc_not.argtypes = [C.POINTER(OrcParams), i32p, i32p]
        L.orc_constant.argtypes = [C.POINTER(OrcParams), i32p, C.c_int32]
        L.orc_gate2_batch.argtypes = [C.POINTER(OrcKeySet), C.c_int, i32p, i32p, i32p, C.c_int32, C.c_int32]
        # fft_standin.c (use_ntt = 4): the AVX2 + FMA fp64-FFT stand-in for upstream's CPU path, not an oracle mode
        L.orc_fft4_available.restype = C.c_int
        L.orc_fft4_negacyclic.argtypes = [i32p, i32p, i32p, C.c_int32]
        _LIB = L
    return _LIB


def _p(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def params(name="P128"):
    p = OrcParams()
    if name == "P128":
        assert lib().orc_params_default(C.byref(p), 128) == 0
    elif name == "P80":
        assert lib().orc_params_default(C.byref(p), 80) == 0
    elif name == "P2048":
        assert lib().orc_params_p2048(C.byref(p)) == 0
    else:
        raise ValueError(name)
    return p


def custom_params(n, N, l, Bgbit, ks_t=8, ks_basebit=2, ks_stdev=2.0 ** -15, bk_stdev=2.0 ** -25, k=1):
    p = OrcParams()
    p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit = n, N, k, l, Bgbit, ks_t, ks_basebit
    p.ks_stdev, p.bk_stdev, p.max_stdev = ks_stdev, bk_stdev, 0.012467
    return p


class Rng:
    def __init__(self, seed):
        self.r = OrcRng()
        lib().orc_rng_seed(C.byref(self.r), seed)

    def next(self):
        return lib().orc_rng_next(C.byref(self.r))

    def gauss(self, sigma):
        return lib().orc_rng_gauss(C.byref(self.r), sigma)


class KeySet:
    """Keys regenerated from (params, seed) -- identical on the product side by specification."""

    def __init__(self, p, seed):
        self.p = p
        self.seed = seed
        self.h = lib().orc_keygen(C.byref(p), seed)
        self.n, self.N, self.k = p.n, p.N, p.k

    def close(self):
        if self.h:
            lib().orc_keyset_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # numpy views of the key arrays
    def lwe_key(self):
        return np.ctypeslib.as_array(self.h.contents.lwe_key, shape=(self.n,))

    def tlwe_key(self):
        return np.ctypeslib.as_array(self.h.contents.tlwe_key, shape=(self.k * self.N,))

    def bk(self):
        return np.ctypeslib.as_array(self.h.contents.bk, shape=(lib().orc_bk_words(C.byref(self.p)),))

    def ksk(self):
        return np.ctypeslib.as_array(self.h.contents.ksk, shape=(lib().orc_ksk_words(C.byref(self.p)),))

    def encrypt(self, rng, bits):
        bits = np.atleast_1d(np.asarray(bits, dtype=np.int32))
        out = np.zeros((len(bits), self.n + 1), dtype=np.int32)
        for i, b in enumerate(bits):
            lib().orc_encrypt_bit(self.h, C.byref(rng.r), int(b), _p(out[i]))
        return out

    def decrypt(self, cts):
        cts = np.ascontiguousarray(cts, dtype=np.int32).reshape(-1, self.n + 1)
        return np.array([lib().orc_decrypt_bit(self.h, _p(c)) for c in cts], dtype=np.int32)

    def phase(self, ct):
        return lib().orc_phase(self.h, _p(np.ascontiguousarray(ct, dtype=np.int32)))

    def gate(self, name, ca, cb, use_ntt=True):
        """use_ntt: False/0 schoolbook, True/1 Goldilocks NTT, 2 two-prime evaluator (same words)."""
        out = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_gate2(self.h, GATES[name], _p(out), _p(np.ascontiguousarray(ca)), _p(np.ascontiguousarray(cb)),
                        int(use_ntt))
        return out

    def gate_batch(self, name, ca, cb, nthreads=1, use_ntt=2):
        """use_ntt=3 is the approximate fp64-FFT stand-in for upstream's CPU path (not the oracle)."""
        ca = np.ascontiguousarray(ca, dtype=np.int32)
        cb = np.ascontiguousarray(cb, dtype=np.int32)
        out = np.zeros_like(ca)
        lib().orc_gate2_batch_mode(self.h, GATES[name], _p(out), _p(ca), _p(cb), ca.shape[0], nthreads, int(use_ntt))
        return out

    def mux(self, a, b, c, use_ntt=True):
        out = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_mux(self.h, _p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)),
                      _p(np.ascontiguousarray(c)), int(use_ntt))
        return out

    def gate_not(self, a):
        out = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_not(C.byref(self.p), _p(out), _p(np.ascontiguousarray(a)))
        return out

    def constant(self, v):
        out = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_constant(C.byref(self.p), _p(out), int(v))
        return out

    def prelude(self, name, ca, cb):
        out = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_gate_prelude(C.byref(self.p), GATES[name], _p(np.ascontiguousarray(ca)),
                               _p(np.ascontiguousarray(cb)), _p(out))
        return out

    def modswitch_ct(self, lin):
        M = 2 * self.N
        return np.array([lib().orc_modswitch(int(x), M) for x in lin], dtype=np.int32)

    def blind_rotate(self, bara, barb, mu=1 << 29, use_ntt=True):
        acc = np.zeros((self.k + 1) * self.N, dtype=np.int32)
        lib().orc_blind_rotate(self.h, _p(np.ascontiguousarray(bara, dtype=np.int32)), int(barb), int(mu),
                               _p(acc), 1 if use_ntt else 0)
        return acc

    def cmux_rotate(self, i, barai, acc, use_ntt=True):
        acc = np.ascontiguousarray(acc, dtype=np.int32).copy()
        lib().orc_cmux_rotate(self.h, int(i), int(barai), _p(acc), 1 if use_ntt else 0)
        return acc

    def sample_extract(self, acc):
        u = np.zeros(self.k * self.N + 1, dtype=np.int32)
        lib().orc_sample_extract(C.byref(self.p), _p(np.ascontiguousarray(acc, dtype=np.int32)), _p(u))
        return u

    def keyswitch(self, u):
        ct = np.zeros(self.n + 1, dtype=np.int32)
        lib().orc_keyswitch(self.h, _p(np.ascontiguousarray(u, dtype=np.int32)), _p(ct))
        return ct

    def bootstrap_woks(self, lin, mu=1 << 29, use_ntt=True):
        u = np.zeros(self.k * self.N + 1, dtype=np.int32)
        lib().orc_bootstrap_woks(self.h, _p(np.ascontiguousarray(lin, dtype=np.int32)), int(mu), _p(u),
                                 1 if use_ntt else 0)
        return u


def negacyclic(ip, tp, ntt):
    ip = np.ascontiguousarray(ip, dtype=np.int32)
    tp = np.ascontiguousarray(tp, dtype=np.int32)
    res = np.zeros_like(tp)
    f = lib().orc_negacyclic_ntt if ntt else lib().orc_negacyclic_schoolbook
    f(_p(res), _p(ip), _p(tp), len(tp))
    return res


def decompose(poly, p):
    poly = np.ascontiguousarray(poly, dtype=np.int32)
    out = np.zeros((p.l, len(poly)), dtype=np.int32)
    lib().orc_decompose(_p(out), _p(poly), C.byref(p))
    return out
