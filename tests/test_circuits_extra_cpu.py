"""Circuits beyond the reference's Function_f path, checked on the CPU over the plaintext
provider (in a subprocess, so the provider's boots* symbols cannot collide with
libtfhe-hip loaded by other tests): the Hamming-distance match,
Function_g with the reference's overflow fixed (8f.3), and the slot-sharded distance."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys, random
t = os.environ["PEBA1_TMP"]
gate = C.CDLL(t + "/libplain_tfhe.so", mode=C.RTLD_GLOBAL)
circ = C.CDLL(t + "/libcircuits_test.so")
V = C.c_void_p
gate.new_default_gate_bootstrapping_parameters.restype = V
gate.new_random_gate_bootstrapping_secret_keyset.restype = V
gate.new_random_gate_bootstrapping_secret_keyset.argtypes = [V]
gate.new_gate_bootstrapping_ciphertext_array.restype = V
gate.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, V]
gate.bootsSymEncrypt.argtypes = [V, C.c_int32, V]
gate.bootsSymDecrypt.argtypes = [V, V]
gate.mock_bootstraps.restype = C.c_int64
params = gate.new_default_gate_bootstrapping_parameters(128)
key = gate.new_random_gate_bootstrapping_secret_keyset(params)
cloud = key + 24
SZ = 24
def enc(v, bits):
    p = gate.new_gate_bootstrapping_ciphertext_array(bits, params)
    for i in range(bits):
        gate.bootsSymEncrypt(p + i * SZ, (v >> i) & 1, key)
    return p
def dec(p, bits):
    return sum(gate.bootsSymDecrypt(p + i * SZ, key) << i for i in range(bits))
def arr(n):
    return gate.new_gate_bootstrapping_ciphertext_array(n, params)

circ.peba1_hamming_count_bits.restype = C.c_int
circ.peba1_hamming_distance.argtypes = [V, V, V, C.c_int, V]
circ.peba1_hamming_match.argtypes = [V, V, V, C.c_int, V, V]
assert [circ.peba1_hamming_count_bits(n) for n in (1, 2, 3, 4, 127, 128, 255, 256)] == [1, 2, 2, 3, 7, 8, 8, 9]
rnd = random.Random(5)
for nbits in (1, 2, 3, 7, 16, 100, 128, 256):
    a = rnd.getrandbits(nbits); b = rnd.getrandbits(nbits)
    if nbits == 128:
        b = a ^ ((1 << 128) - 1)                 # maximum distance
    w = circ.peba1_hamming_count_bits(nbits)
    ca, cb, cnt = enc(a, nbits), enc(b, nbits), arr(w)
    gate.mock_reset()
    circ.peba1_hamming_distance(cnt, ca, cb, nbits, cloud)
    hd = bin(a ^ b).count("1")
    assert dec(cnt, w) == hd, (nbits, dec(cnt, w), hd)
    if nbits == 128:
        print("hamming128 bootstraps", gate.mock_bootstraps())
    for bound in (0, hd - 1, hd, hd + 1):
        if bound < 0 or bound >= (1 << w):
            continue
        rb = arr(w)
        circ.peba1_hamming_match(rb, ca, cb, nbits, enc(bound, w), cloud)
        assert dec(rb, w) == (1 if hd > bound else 0), (nbits, bound)

# Function_g: (1-b)*r0 + b*r1 on `bitsize` bits (reference Math.cpp:390-417 with D4 fixed).
# The circuit is the reference's gate for gate, INCLUDING its handling of a zero subtrahend:
# bootsSUBNbit forces the sign bit of "-0" (Math.cpp:137-138), so |1 - 0| comes out as 255 and
# Function_g selects r0 correctly only when b = 1.  Model of the reference's subtraction:
def sub_ref(a, b, bits):
    tb = (((~b) & ((1 << bits) - 1)) + 1) & ((1 << bits) - 1) | (1 << bits)
    s = a + tb
    carry = (s >> (bits + 1)) & 1
    s &= (1 << (bits + 1)) - 1
    return s if carry else (-s) & ((1 << (bits + 1)) - 1)
assert sub_ref(122, 204, 8) == 82 and sub_ref(204, 122, 8) == 82 and sub_ref(1, 1, 8) == 0 and sub_ref(1, 0, 8) == 255
circ.peba1_function_g.argtypes = [V, V, V, V, C.c_int, V]
circ.peba1_sub_nbit.argtypes = [V, V, V, C.c_int, V]
for a_, b_ in ((1, 0), (1, 1), (200, 3), (3, 200), (0, 0)):
    r = arr(9)
    circ.peba1_sub_nbit(r, enc(a_, 8), enc(b_, 8), 8, cloud)
    assert dec(r, 9) == sub_ref(a_, b_, 8), (a_, b_, dec(r, 9))
for b, r0, r1 in ((1, 17, 99), (0, 17, 99), (1, 255, 0), (0, 0, 255)):
    res = arr(8)
    circ.peba1_function_g(res, enc(b, 8), enc(r0, 8), enc(r1, 8), 8, cloud)
    want = (((sub_ref(1, b, 8) & 255) * r0 & 255) + (b * r1 & 255)) & 255
    assert dec(res, 8) == want, (b, r0, r1, dec(res, 8), want)
    if b == 1:
        assert want == r1                        # the protocol's intent holds on this branch

# slot-sharded distance: partials + combine == the unsharded distance and comparator
circ.peba1_partial_distance.argtypes = [V, V, V, C.c_int, C.c_int, V]
circ.peba1_combine_and_compare.argtypes = [V, V, C.c_int, V, V]
nslots = 10
tmpl = [(37 * i + 11) % 255 for i in range(nslots)]
probe = [(91 * i + 5) % 256 for i in range(nslots)]
d = sum((x - y) ** 2 for x, y in zip(probe, tmpl))
for parts in (1, 2, 3, 5):
    partials = []
    for r in range(parts):
        lo, hi = r * nslots // parts, (r + 1) * nslots // parts
        S = (V * (hi - lo))(*[enc(probe[i], 8) for i in range(lo, hi)])
        T = (V * (hi - lo))(*[enc(tmpl[i], 8) for i in range(lo, hi)])
        p = arr(24)
        circ.peba1_partial_distance(p, S, T, hi - lo, 8, cloud)
        assert dec(p, 24) == sum((probe[i] - tmpl[i]) ** 2 for i in range(lo, hi))
        partials.append(p)
    for bound in (d - 1, d, d + 1):
        rb = arr(24)
        circ.peba1_combine_and_compare(rb, (V * parts)(*partials), parts, enc(bound, 24), cloud)
        assert dec(rb, 24) == (1 if d > bound else 0), (parts, bound)

# optimised Function_f (circuits_fast.cpp): same decrypted outputs as the reference's circuit and
# as the plaintext rule, through a different DAG
circ.peba1_function_f.argtypes = [V, V, V, C.c_int, V, C.c_int, V]
circ.peba1_function_f_fast.argtypes = [V, V, V, C.c_int, V, C.c_int, V]
circ.peba1_euclidean_distance_fast.argtypes = [V, V, V, C.c_int, C.c_int, V]
def vec(vals, bits):
    return (V * len(vals))(*[enc(v, bits) for v in vals])
rnd = random.Random(11)
cases = [([0], [0]), ([255], [0]), ([0], [255]), ([255] * 5, [0] * 5), ([7, 200, 13], [7, 200, 13])]
for n in (1, 2, 3, 8, 17):
    cases.append(([rnd.randrange(256) for _ in range(n)], [rnd.randrange(256) for _ in range(n)]))
for probe, tmpl in cases:
    d = sum((x - y) ** 2 for x, y in zip(probe, tmpl))
    r = arr(24)
    circ.peba1_euclidean_distance_fast(r, vec(probe, 8), vec(tmpl, 8), len(probe), 8, cloud)
    assert dec(r, 24) == d, (probe, tmpl, dec(r, 24), d)
    for bound in sorted({0, max(d - 1, 0), d, d + 1, (1 << 24) - 1}):
        rb, rb_ref = arr(24), arr(24)
        circ.peba1_function_f_fast(rb, vec(probe, 8), vec(tmpl, 8), len(probe), enc(bound, 24), 8, cloud)
        assert dec(rb, 24) == (1 if d > bound else 0), (probe, tmpl, bound)
        # the reference's own circuit agrees wherever its subtraction is right: bootsSUBNbit
        # mishandles a zero subtrahend (Math.cpp:137-138), which in HE_EuclideanDistance is
        # the sample value (it computes b - a): (0 - 255)^2 comes out as 1
        if len(probe) <= 3 and 0 not in probe and 0 not in tmpl:
            circ.peba1_function_f(rb_ref, vec(probe, 8), vec(tmpl, 8), len(probe), enc(bound, 24), 8, cloud)
            assert dec(rb_ref, 24) == dec(rb, 24)
# other bit sizes (4-bit slots: 12-bit arithmetic; sums wrap modulo 2^12 as in the reference)
for probe, tmpl in (([15, 0, 9], [0, 15, 3]), ([15] * 20, [0] * 20)):
    d = sum((x - y) ** 2 for x, y in zip(probe, tmpl)) % (1 << 12)
    r = arr(12)
    circ.peba1_euclidean_distance_fast(r, vec(probe, 4), vec(tmpl, 4), len(probe), 4, cloud)
    assert dec(r, 12) == d, (probe, tmpl, dec(r, 12), d)
# the full-size match: 128 slots x 8 bit (SURVEY 8c inputs), gate count of the optimised DAG
tmpl = [(37 * i + 11) % 255 for i in range(128)]
for probe, want in (([t + 1 for t in tmpl], 0), ([(91 * i + 5) % 256 for i in range(128)], 1)):
    rb = arr(24)
    gate.mock_reset()
    circ.peba1_function_f_fast(rb, vec(probe, 8), vec(tmpl, 8), 128, enc(256, 24), 8, cloud)
    assert dec(rb, 24) == want
    nb = gate.mock_bootstraps()
print("function_f_fast bootstraps", nb)
assert nb < 40000
print("OK")
'''


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    t = str(tmp_path_factory.mktemp("extra"))
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["g++", "-O1", "-std=gnu++11", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(ROOT, "tests/mock/plain_tfhe.cpp"), "-o", t + "/libplain_tfhe.so"])
    subprocess.check_call(["g++", "-O1", "-std=gnu++17", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(ROOT, "peba1_amd/csrc/circuits.cpp"), os.path.join(ROOT, "peba1_amd/csrc/circuits_fast.cpp"), "-o", t + "/libcircuits_test.so"])
    with open(t + "/worker.py", "w") as f:
        f.write(WORKER)
    return t


def test_hamming_function_g_and_sharded_distance(built):
    out = subprocess.run([sys.executable, built + "/worker.py"], env=dict(os.environ, PEBA1_TMP=built),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
    print(out.stdout)
