"""Python plumbing for libpeba1-circuits.so (include/peba1_circuits.h): the reference's
encrypted circuits and protocol function f (Math.cpp:27-417) over the boots* gate API."""
import ctypes as C
import os

from . import api
from . import lib as _l

_circ = None


def load():
    global _circ
    if _circ is None:
        _l.load()   # provider of the boots* symbols first (RTLD_GLOBAL)
        if not os.path.exists(_l.CIRCUITS_PATH):
            raise RuntimeError(f"{_l.CIRCUITS_PATH} is missing: run __graft_entry__.build()")
        Lc = C.CDLL(_l.CIRCUITS_PATH)
        LS, CK = _l.LS, _l.CK
        LSP = C.POINTER(LS)
        sig = {
            "peba1_add_1bit": [LS, LS, LS, LS, CK],
            "peba1_add_nbit": [LS, LS, LS, LS, C.c_int, CK],
            "peba1_twos_complement": [LS, LS, C.c_int, CK],
            "peba1_abs": [LS, LS, C.c_int, CK],
            "peba1_sub_nbit": [LS, LS, LS, C.c_int, CK],
            "peba1_shift_left": [LS, LS, C.c_int, C.c_int, CK],
            "peba1_shift_right": [LS, LS, C.c_int, C.c_int, CK],
            "peba1_shift_left_inplace": [LS, C.c_int, C.c_int, CK],
            "peba1_multiply": [LS, LS, LS, C.c_int, CK],
            "peba1_compare_bit": [LS, LS, LS, LS, LS, CK],
            "peba1_minimum": [LS, LS, LS, LS, C.c_int, CK],
            "peba1_euclidean_distance": [LS, LSP, LSP, C.c_int, C.c_int, CK],
            "peba1_function_f": [LS, LSP, LSP, C.c_int, LS, C.c_int, CK],
            "peba1_function_g": [LS, LS, LS, LS, C.c_int, CK],
            "peba1_euclidean_distance_fast": [LS, LSP, LSP, C.c_int, C.c_int, CK],
            "peba1_function_f_fast": [LS, LSP, LSP, C.c_int, LS, C.c_int, CK],
            "peba1_partial_distance": [LS, LSP, LSP, C.c_int, C.c_int, CK],
            "peba1_combine_and_compare": [LS, LSP, C.c_int, LS, CK],
            "peba1_combine_and_compare_fast": [LS, LSP, C.c_int, LS, CK],
            "peba1_hamming_distance": [LS, LS, LS, C.c_int, CK],
            "peba1_hamming_match": [LS, LS, LS, C.c_int, LS, CK],
        }
        for name, args in sig.items():
            f = getattr(Lc, name)
            f.restype = None
            f.argtypes = args
        Lc.peba1_hamming_count_bits.restype = C.c_int
        Lc.peba1_hamming_count_bits.argtypes = [C.c_int]
        _circ = Lc
    return _circ


def _ptr_array(arrays):
    arr = (_l.LS * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ptr
    return arr


class EncryptedVector:
    """A template / sample: nslots features of `bitsize` bits, one LweSample array per slot
    (the reference's std::vector<LweSample*>, main.cpp:53-70)."""

    def __init__(self, params, values, bitsize, key):
        self.slots = []
        for v in values:
            a = api.CiphertextArray(params, bitsize)
            a.encrypt([(int(v) >> j) & 1 for j in range(bitsize)], key)
            self.slots.append(a)

    def to_device(self):
        for a in self.slots:
            a.set_words(a.words())
        return self


def encrypt_number(params, value, bits, key):
    a = api.CiphertextArray(params, bits)
    return a.encrypt([(int(value) >> j) & 1 for j in range(bits)], key)


def decrypt_number(arr, key, bits=None):
    d = arr.decrypt(key)
    bits = len(d) if bits is None else bits
    return sum(int(d[i]) << i for i in range(bits))


def function_f(result_b, sample, template, bound, bitsize, key):
    """Function_f(result_b, a=sample, b=template, bound, bitsize, cloud_key), Math.cpp:379."""
    load().peba1_function_f(result_b.ptr, _ptr_array(sample.slots), _ptr_array(template.slots), len(sample.slots),
                            bound.ptr, bitsize, key.cloud)


def function_f_fast(result_b, sample, template, bound, bitsize, key):
    """Same result as function_f through the optimised DAG of circuits_fast.cpp (not the
    reference's gate sequence; about 8x fewer bootstraps, 5x less depth)."""
    load().peba1_function_f_fast(result_b.ptr, _ptr_array(sample.slots), _ptr_array(template.slots),
                                 len(sample.slots), bound.ptr, bitsize, key.cloud)


def euclidean_distance_fast(result, sample, template, bitsize, key):
    load().peba1_euclidean_distance_fast(result.ptr, _ptr_array(sample.slots), _ptr_array(template.slots),
                                         len(sample.slots), bitsize, key.cloud)


def function_g(result, result_b, r0, r1, bitsize, key):
    """Function_g(result, result_b, r0, r1, bitsize, cloud_key), Math.cpp:390: (1-b)*r0 + b*r1 on
    `bitsize`-sample numbers (the reference's heap overflow, SURVEY D4, fixed)."""
    load().peba1_function_g(result.ptr, result_b.ptr, r0.ptr, r1.ptr, bitsize, key.cloud)


def euclidean_distance(result, sample, template, bitsize, key):
    load().peba1_euclidean_distance(result.ptr, _ptr_array(sample.slots), _ptr_array(template.slots),
                                    len(sample.slots), bitsize, key.cloud)


def partial_distance(partial, sample_slots, template_slots, bitsize, key):
    load().peba1_partial_distance(partial.ptr, _ptr_array(sample_slots), _ptr_array(template_slots),
                                  len(sample_slots), bitsize, key.cloud)


def combine_and_compare(result_b, partials, bound, key):
    load().peba1_combine_and_compare(result_b.ptr, _ptr_array(partials), len(partials), bound.ptr, key.cloud)


def hamming_match(result_b, a, b, nbits, bound, key):
    """result_b[0] = (popcount(a XOR b) > bound); a, b: CiphertextArray of nbits samples."""
    load().peba1_hamming_match(result_b.ptr, a.ptr, b.ptr, nbits, bound.ptr, key.cloud)


def hamming_distance(count, a, b, nbits, key):
    load().peba1_hamming_distance(count.ptr, a.ptr, b.ptr, nbits, key.cloud)


def hamming_count_bits(nbits):
    return load().peba1_hamming_count_bits(nbits)
