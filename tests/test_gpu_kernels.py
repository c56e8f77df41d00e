"""GPU parity of each HIP kernel against the CPU oracle, word for word (bit-exact:
all arithmetic is integer / Torus32).  Call path: Python -> C ABI (include/tfhe_hip.h)
-> HIP kernels.  Reference semantics: SURVEY.md Appendix A.3; reference call sites
/root/reference/src/Math.cpp:34-43."""
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_key_material_identical(p128_keys):
    _, ks, oks = p128_keys
    assert (ks.lwe_key() == oks.lwe_key()).all()
    assert (ks.tlwe_key() == oks.tlwe_key()).all()
    assert (ks.bk() == oks.bk()).all()
    assert (ks.ksk() == oks.ksk()).all()


def test_negacyclic_ntt_exact(p128_keys, oracle):
    """a15: negacyclic polynomial multiply through the two-prime device NTT == schoolbook mod 2^32."""
    from peba1_amd import api
    _, ks, _ = p128_keys
    rng = np.random.default_rng(7)
    count, N = 12, 1024
    ip = rng.integers(-64, 64, (count, N), dtype=np.int64).astype(np.int32)
    tp = rng.integers(-2**31, 2**31, (count, N), dtype=np.int64).astype(np.int32)
    # edge cases: extreme digits against extreme torus values, zeros, a monomial
    ip[0, :] = -64; tp[0, :] = -2**31
    ip[1, :] = 63; tp[1, :] = 2**31 - 1
    ip[2, :] = 0
    ip[3, :] = 0; ip[3, 1023] = 1
    got = api.kernel_negacyclic(ks, ip, tp)
    for c in range(count):
        want = oracle.negacyclic(ip[c], tp[c], ntt=False)
        assert (got[c] == want).all(), f"poly {c}"


@pytest.mark.parametrize("pname", ["P128", "P2048"])
def test_split_transform_negacyclic_exact(oracle, pname):
    """a15 through the split transforms of the 8-wave kernel form: stage 0 outside, two half-size
    transforms on the sub-tree twiddle tables, the key image read in place, the last inverse stage
    and the CRT after one exchange == schoolbook mod 2^32 (N = 1024 -> two 512-point transforms,
    N = 2048 -> two 1024-point ones)."""
    from peba1_amd import api
    pp = api.ParameterSet(128) if pname == "P128" else api.ParameterSet(p2048=True)
    ks = api.SecretKeySet(pp, 0x51, device=True)
    try:
        rng = np.random.default_rng(11)
        count, N = 8, pp.N
        # gadget-digit sized multiplicands (the product must stay inside the two-prime range, ntt_field.hpp)
        ip = rng.integers(-64, 64, (count, N), dtype=np.int64).astype(np.int32)
        tp = rng.integers(-2**31, 2**31, (count, N), dtype=np.int64).astype(np.int32)
        ip[0, :] = -64; tp[0, :] = -2**31
        ip[1, :] = 63; tp[1, :] = 2**31 - 1
        ip[2, :] = 0
        ip[3, :] = 0; ip[3, N - 1] = 1
        ip[4, :] = 0; ip[4, N // 2] = 1                       # only the upper half: stage 0 alone
        ip[5, :] = rng.integers(-512, 512, N)                  # the widest digits any accepted set has (Bg = 2^10)
        ip[6, :] = 0; ip[6, 1] = -512; ip[6, N // 2 + 1] = 511
        try:
            api.set_tuning("br_variant", 2)
            got = api.kernel_negacyclic(ks, ip, tp)
        finally:
            api.set_tuning("br_variant", -1)
        assert (got == api.kernel_negacyclic(ks, ip, tp)).all(), "split and unsplit transforms differ"
        for c in range(count):
            assert (got[c] == oracle.negacyclic(ip[c], tp[c], ntt=False)).all(), f"poly {c}"
    finally:
        ks.close()


def test_blind_rotate_matches_oracle(p128_keys, oracle):
    """a11-a14, a16: modswitch + blind rotate + extract on real ciphertext combinations."""
    from peba1_amd import api
    _, ks, oks = p128_keys
    r = oracle.Rng(11)
    cts = oks.encrypt(r, [1, 1, 0, 1])
    lins = np.stack([oks.prelude("AND", cts[0], cts[1]), oks.prelude("XOR", cts[2], cts[3])])
    u, acc = api.kernel_bootstrap_woks(ks, lins, want_acc=True)
    for c in range(2):
        bar = oks.modswitch_ct(lins[c])
        want_acc = oks.blind_rotate(bar[:-1], bar[-1])
        assert (acc[c] == want_acc).all(), f"accumulator {c}"
        assert (u[c] == oks.sample_extract(want_acc)).all(), f"extract {c}"


@pytest.mark.parametrize("variant", [4, 0])
def test_both_blind_rotate_forms_are_bit_exact(p128_keys, oracle, variant):
    """The 2-wave kernel (br_variant 4: the admissibility fallback) and the 4-wave kernel compute the same integers."""
    from peba1_amd import api
    _, ks, oks = p128_keys
    r = oracle.Rng(23)
    cts = oks.encrypt(r, [0, 1, 1, 1, 0, 0])
    lins = np.stack([oks.prelude("OR", cts[0], cts[1]), oks.prelude("XNOR", cts[2], cts[3]),
                     oks.prelude("ANDNY", cts[4], cts[5])])
    api.set_tuning("br_variant", variant)
    api.set_tuning("br8_max_rotations", 0)           # three rotations would otherwise take the 8-wave form
    try:
        u, acc = api.kernel_bootstrap_woks(ks, lins, want_acc=True)
    finally:
        api.set_tuning("br_variant", -1)
        api.set_tuning("br8_max_rotations", 1 << 30)
    for c in range(3):
        bar = oks.modswitch_ct(lins[c])
        want_acc = oks.blind_rotate(bar[:-1], bar[-1])
        assert (acc[c] == want_acc).all(), f"accumulator {c}"
        assert (u[c] == oks.sample_extract(want_acc)).all()


def test_blind_rotate_edge_inputs(p128_keys, oracle):
    """all-zero mask (every step skipped), barb = 0, and abar values at the wrap points."""
    from peba1_amd import api
    pp, ks, oks = p128_keys
    lin = np.zeros((3, pp.words), dtype=np.int32)
    lin[1, :] = np.int32(1 << 21)              # every abar = 1
    lin[2, ::2] = np.int32(-(1 << 21))         # abar = 2N-1 on even positions
    lin[2, 5] = np.int32(1 << 31 >> 0) if False else np.int32(-2**31)   # abar = N
    u, acc = api.kernel_bootstrap_woks(ks, lin, want_acc=True)
    for c in range(3):
        bar = oks.modswitch_ct(lin[c])
        want = oks.blind_rotate(bar[:-1], bar[-1])
        assert (acc[c] == want).all(), f"case {c}"


def test_keyswitch_matches_oracle(p128_keys, oracle):
    """a17: key switch of arbitrary extracted samples through the per-gate kernel -- five samples (cut into 32 coefficient
    ranges, partial sums + reduce) and, unsplit, a launch wide enough that every gate is one workgroup (16,400 gates with the
    tiled kernels off: compared with the oracle on a sample of rows and with the tiled default on all of them)."""
    from peba1_amd import api
    pp, ks, oks = p128_keys
    rng = np.random.default_rng(3)
    u = rng.integers(-2**31, 2**31, (5, pp.N + 1), dtype=np.int64).astype(np.int32)
    u[0, :] = 0                      # all digits zero: nothing subtracted
    u[1, :-1] = -1                   # all digits 3 after rounding offset wraps
    got = api.kernel_keyswitch(ks, u)
    for c in range(5):
        assert (got[c] == oks.keyswitch(u[c])).all(), f"sample {c}"
    many = rng.integers(-2**31, 2**31, (16400, pp.N + 1), dtype=np.int64).astype(np.int32)
    many[:5] = u
    tiled = api.kernel_keyswitch(ks, many)
    api.set_tuning("ks_tile", 0)
    try:
        unsplit = api.kernel_keyswitch(ks, many)
    finally:
        api.set_tuning("ks_tile", 16)
    assert (unsplit == tiled).all()
    for c in (0, 1, 2, 5, 8191, 8192, 16399):        # 8,192 = the chunk boundary of the tiled launches
        assert (unsplit[c] == oks.keyswitch(many[c])).all(), f"sample {c} of the wide launch"


@pytest.fixture(scope="module")
def keys_by_set(oracle, p128_keys):
    """(parameter set, product keys on the device, oracle keys) of P128 (the session's), P80 and P2048, made on first use."""
    from peba1_amd import api
    made = {"P128": p128_keys}
    own = []

    def get(pname):
        if pname not in made:
            seed = {"P80": 0x80, "P2048": 0x2048}[pname]
            pp = api.ParameterSet(80) if pname == "P80" else api.ParameterSet(p2048=True)
            ks = api.SecretKeySet(pp, seed, device=True)
            own.append(ks)
            made[pname] = (pp, ks, oracle.KeySet(oracle.params(pname), seed))
        return made[pname]
    yield get
    for ks in own:
        ks.close()


def _oracle_rows(fn, rows):
    """fn over the rows on the host's threads (ctypes releases the GIL; the oracle is re-entrant)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(16, os.cpu_count() or 4)) as ex:
        return np.stack(list(ex.map(fn, rows)))


@pytest.mark.parametrize("pname,tile,count", [("P128", 16, 32), ("P128", 16, 45), ("P128", 24, 48), ("P128", 24, 61), ("P128", 32, 64),
                                              ("P128", 32, 77), ("P80", 16, 45), ("P80", 24, 61), ("P80", 32, 77),
                                              ("P2048", 16, 45), ("P2048", 24, 61), ("P2048", 32, 77)])
def test_tiled_keyswitch_matches_oracle(keys_by_set, oracle, pname, tile, count):
    """a17, wide launches: one pass over a range's KSK rows serves a tile of gates; full and ragged last tiles, all-zero and
    all-three digits.  EVERY row of every form against the oracle, for the three parameter sets -- the
    index form of the key switch has one instantiation per row width (128, 192, 320 threads: P80, P128, P2048) and tile
    (16, 24, 32), and names physical registers: all nine are launched here on random words."""
    from peba1_amd import api
    pp, ks, oks = keys_by_set(pname)
    rng = np.random.default_rng(1000 + count + pp.n)
    u = rng.integers(-2**31, 2**31, (count, pp.k * pp.N + 1), dtype=np.int64).astype(np.int32)
    u[0, :] = 0
    u[1, :-1] = -1
    u[count - 1, :-1] = 0x40000000   # digit 1 at the first position only
    want = _oracle_rows(oks.keyswitch, u)
    api.set_tuning("ks_tile", tile)
    try:
        forms = {}
        forms["index"] = api.kernel_keyswitch(ks, u)     # rows in pinned registers, picked through the VGPR index mode: the default
        api.set_tuning("ks_index", 0)                    # rows in thread-private LDS strips (tiles of 16 whatever ks_tile says)
        forms["strips"] = api.kernel_keyswitch(ks, u)
        api.set_tuning("ks_tile", 0)
        forms["per gate"] = api.kernel_keyswitch(ks, u)
    finally:
        api.set_tuning("ks_tile", 16)
        api.set_tuning("ks_index", 1)
    for name, got in forms.items():
        bad = np.argwhere((got != want).any(axis=1)).ravel()
        assert bad.size == 0, f"{pname} tile {tile}, form {name}: rows {bad[:8]} differ from the oracle"


@pytest.mark.parametrize("pname,rotations,switches", [("P128", 512, 1024), ("P80", 256, 1024), ("P2048", 64, 256)])
def test_random_input_parity_soak_in_every_launch_form(keys_by_set, oracle, pname, rotations, switches):
    """SURVEY section 4 tier 4 inside the driver-run suite (the long form is tools/parity_soak.py ->
    profiles/): uniformly random LWE inputs rotated in each of the five launch forms a default build can take -- the wide
    launch with and without its tail round on the 8-wave form, launches of 256 (the 8-wave form at N = 1024), the 4-wave
    form without digit tables, the split form -- and EVERY word of every extracted sample (a signed permutation of the
    whole accumulator) compared with the oracle, whose exact product runs over one 64-bit prime: arithmetic independent of
    the kernels' two 27-bit primes + signed CRT.  The circuit digests see a rotation only through later gates' modulus
    switches, which hide low bits; this does not.  Then random extracted samples through the default key switch (pinned
    registers, VGPR index mode) at tiles 16 / 24 / 32 and through the LDS-strip form."""
    from peba1_amd import api
    pp, ks, oks = keys_by_set(pname)
    rng = np.random.default_rng(pp.N + rotations)
    lin = rng.integers(-2**31, 2**31, (rotations, pp.words), dtype=np.int64).astype(np.int32)
    want = _oracle_rows(oks.bootstrap_woks, lin)
    defaults = {"br_tail8": 1, "br_variant": -1, "br_digit_table": 1, "br8_max_rotations": 1 << 30}
    wide = rotations if pp.N == 2048 else max(rotations, 600)      # N = 1024: 600 > 2 x 256 CUs, so the wide launch has a tail round
    pad = np.concatenate([lin, lin[:wide - rotations]]) if wide > rotations else lin
    forms = [("default wide launch", {}, pad, wide),
             ("one launch per level (br_tail8 = 0)", {"br_tail8": 0}, pad, wide),
             ("launches of at most 256", {}, lin, 256),
             ("4-wave form, no digit table", {"br_variant": 0, "br_digit_table": 0, "br8_max_rotations": 0}, lin, rotations),
             ("split form", {"br_variant": 2}, lin, rotations)]
    try:
        for label, tunings, inputs, chunk in forms:
            for k, v in tunings.items():
                api.set_tuning(k, v)
            got = np.concatenate([api.kernel_bootstrap_woks(ks, inputs[i:i + chunk]) for i in range(0, len(inputs), chunk)])[:rotations]
            bad = np.argwhere((got != want).any(axis=1)).ravel()
            assert bad.size == 0, f"{pname}, {label}: rotations {bad[:8]} differ from the oracle"
            for k in tunings:
                api.set_tuning(k, defaults[k])
    finally:
        for k, v in defaults.items():
            api.set_tuning(k, v)
    u = rng.integers(-2**31, 2**31, (switches, pp.k * pp.N + 1), dtype=np.int64).astype(np.int32)
    want_ks = _oracle_rows(oks.keyswitch, u)
    try:
        for tile, index in ((16, 1), (24, 1), (32, 1), (16, 0)):
            api.set_tuning("ks_tile", tile)
            api.set_tuning("ks_index", index)
            got = api.kernel_keyswitch(ks, u)
            bad = np.argwhere((got != want_ks).any(axis=1)).ravel()
            assert bad.size == 0, f"{pname}, key switch tile {tile} index form {index}: rows {bad[:8]} differ from the oracle"
    finally:
        api.set_tuning("ks_tile", 16)
        api.set_tuning("ks_index", 1)


# ---------------------------------------------------------------- BASELINE configs[4]: N = 2048
@pytest.fixture(scope="module")
def p2048_keys(keys_by_set):
    """High-security set (N=2048, Bg=2^6, l=3; n=1024, ks 8x2 bit fixed by this repo): one keyset per module."""
    return keys_by_set("P2048")


def test_p2048_negacyclic_and_gates(p2048_keys, oracle):
    from peba1_amd import api, lib
    pp, ks, oks = p2048_keys
    assert (ks.bk() == oks.bk()).all() and (ks.ksk() == oks.ksk()).all()
    rng = np.random.default_rng(11)
    ip = rng.integers(-32, 32, (3, 2048), dtype=np.int64).astype(np.int32)
    tp = rng.integers(-2**31, 2**31, (3, 2048), dtype=np.int64).astype(np.int32)
    ip[0, :] = -32; tp[0, :] = -2**31
    got = api.kernel_negacyclic(ks, ip, tp)
    for c in range(3):
        assert (got[c] == oracle.negacyclic(ip[c], tp[c], ntt=False)).all(), c
    # one blind rotation + key switch, word for word
    r = oracle.Rng(5)
    cts = oks.encrypt(r, [1, 0, 1])
    lin = oks.prelude("XOR", cts[0], cts[1])
    u, acc = api.kernel_bootstrap_woks(ks, lin[None, :], want_acc=True)
    bar = oks.modswitch_ct(lin)
    want_acc = oks.blind_rotate(bar[:-1], bar[-1])
    assert (acc[0] == want_acc).all()
    assert (u[0] == oks.sample_extract(want_acc)).all()
    assert (api.kernel_keyswitch(ks, u)[0] == oks.keyswitch(u[0])).all()
    # wide launch: tiled key switch (320 threads, ranges of 64 coefficients)
    uw = rng.integers(-2**31, 2**31, (40, 2048 + 1), dtype=np.int64).astype(np.int32)
    gw = api.kernel_keyswitch(ks, uw)
    for c in (0, 17, 39):
        assert (gw[c] == oks.keyswitch(uw[c])).all(), c
    # gates through the public API
    L = lib.load()
    L.tfhe_hip_set_encrypt_seed(9)
    a = api.CiphertextArray(pp, 4).encrypt([0, 0, 1, 1], ks)
    b = api.CiphertextArray(pp, 4).encrypt([0, 1, 0, 1], ks)
    res = api.CiphertextArray(pp, 4)
    api.gate_batch("NAND", res, a, b, ks)
    wa, wb, got = a.words(), b.words(), res.words()
    assert (got[1] == oks.gate("NAND", wa[1], wb[1])).all()
    assert list(res.decrypt(ks)) == [1, 1, 1, 0]
    m = api.CiphertextArray(pp, 1)
    L.bootsMUX(m.at(0), a.at(2), b.at(1), b.at(0), ks.cloud)
    assert (m.words()[0] == oks.mux(wa[2], wb[1], wb[0])).all() and m.decrypt(ks)[0] == 1


# ---------------------------------------------------------------- the lambda<=80 set (n=500, l=2, Bg=2^10)
def test_p80_blind_rotate_keyswitch_and_gates(oracle):
    """new_default_gate_bootstrapping_parameters(80): gadget length and base are run-time
    values in the kernels; the widest digits (|d| <= 512) stay inside the two-prime bound."""
    from peba1_amd import api, lib
    seed = 0x80
    pp = api.ParameterSet(80)
    assert (pp.n, pp.N, pp.l, pp.Bgbit) == (500, 1024, 2, 10)
    ks = api.SecretKeySet(pp, seed, device=True)
    oks = oracle.KeySet(oracle.params("P80"), seed)
    try:
        assert (ks.bk() == oks.bk()).all() and (ks.ksk() == oks.ksk()).all()
        rng = np.random.default_rng(80)
        ip = rng.integers(-512, 512, (2, 1024), dtype=np.int64).astype(np.int32)
        tp = rng.integers(-2**31, 2**31, (2, 1024), dtype=np.int64).astype(np.int32)
        ip[0, :] = -512; tp[0, :] = -2**31
        got = api.kernel_negacyclic(ks, ip, tp)
        for c in range(2):
            assert (got[c] == oracle.negacyclic(ip[c], tp[c], ntt=False)).all(), c
        r = oracle.Rng(6)
        cts = oks.encrypt(r, [1, 1, 0])
        lin = np.stack([oks.prelude("AND", cts[0], cts[1]), oks.prelude("XNOR", cts[1], cts[2])])
        u, acc = api.kernel_bootstrap_woks(ks, lin, want_acc=True)
        for c in range(2):
            bar = oks.modswitch_ct(lin[c])
            want_acc = oks.blind_rotate(bar[:-1], bar[-1])
            assert (acc[c] == want_acc).all(), c
            assert (u[c] == oks.sample_extract(want_acc)).all(), c
        ksw = api.kernel_keyswitch(ks, u)
        for c in range(2):
            assert (ksw[c] == oks.keyswitch(u[c])).all(), c
        uw = rng.integers(-2**31, 2**31, (35, 1024 + 1), dtype=np.int64).astype(np.int32)   # tiled kernel, 128 threads
        gw = api.kernel_keyswitch(ks, uw)
        for c in (0, 16, 34):
            assert (gw[c] == oks.keyswitch(uw[c])).all(), c
        L = lib.load()
        L.tfhe_hip_set_encrypt_seed(8)
        a = api.CiphertextArray(pp, 4).encrypt([0, 0, 1, 1], ks)
        b = api.CiphertextArray(pp, 4).encrypt([0, 1, 0, 1], ks)
        res = api.CiphertextArray(pp, 4)
        wa, wb = a.words(), b.words()
        for name, want in (("XOR", [0, 1, 1, 0]), ("OR", [0, 1, 1, 1])):
            api.gate_batch(name, res, a, b, ks)
            got = res.words()
            for i in range(4):
                assert (got[i] == oks.gate(name, wa[i], wb[i])).all(), (name, i)
            assert list(res.decrypt(ks)) == want
        m = api.CiphertextArray(pp, 1)
        L.bootsMUX(m.at(0), a.at(1), b.at(1), b.at(0), ks.cloud)
        assert (m.words()[0] == oks.mux(wa[1], wb[1], wb[0])).all() and m.decrypt(ks)[0] == 0
    finally:
        ks.close()


@pytest.mark.parametrize("pname", ["P128", "P80"])
def test_external_product_single_step(oracle, p128_keys, pname):
    """a14 in isolation: ONE CMUX step ACC <- ACC + BK_i (.) ((X^abar - 1) ACC) -- gadget
    decomposition, l forward NTTs per input polynomial, multiply-accumulate against row i of the key
    image, inverse NTTs, CRT -- against the oracle's orc_cmux_rotate on the same accumulator.
    The input sample has exactly one non-zero mask word, so every other step of the blind rotation
    is skipped (tfhe_blindRotate skips abar = 0) and the accumulator that comes back is the test
    vector after that single external product.  Step indices at both ends and in the middle,
    rotations by 1, N-1, N, N+1 and 2N-1 (sign wrap of X^abar), several test-vector offsets."""
    from peba1_amd import api
    if pname == "P128":
        pp, ks, oks = p128_keys
        own = False
    else:
        pp = api.ParameterSet(80)
        ks = api.SecretKeySet(pp, 0x80E1, device=True)
        oks = oracle.KeySet(oracle.params("P80"), 0x80E1)
        own = True
    try:
        N, n = pp.N, pp.n
        unit = 1 << (31 - 10)                        # torus value whose modulus switch is 1 (N = 1024)
        cases = [(0, 1, 0), (n - 1, N - 1, 5), (n // 2, N, 2 * N - 1), (7, N + 1, N), (n - 2, 2 * N - 1, 1),
                 (3, 517, 1300)]
        lin = np.zeros((len(cases), pp.words), dtype=np.int32)
        for c, (i, abar, bbar) in enumerate(cases):
            lin[c, i] = np.int64(abar * unit).astype(np.int32)
            lin[c, n] = np.int64(bbar * unit).astype(np.int32)
        u, acc = api.kernel_bootstrap_woks(ks, lin, want_acc=True)
        for c, (i, abar, bbar) in enumerate(cases):
            bar = oks.modswitch_ct(lin[c])
            assert bar[i] == abar and bar[n] == bbar and np.count_nonzero(bar[:n]) == 1
            testvec = oks.blind_rotate(np.zeros(n, dtype=np.int32), bbar)          # no step taken
            want = oks.cmux_rotate(i, abar, testvec)
            assert (acc[c] == want).all(), f"{pname} step {i} abar {abar} bbar {bbar}"
            assert (want != testvec).any()
            assert (u[c] == oks.sample_extract(want)).all()
    finally:
        if own:
            ks.close()


@pytest.mark.parametrize("pname", ["P128", "P80", "P2048"])
def test_every_selectable_kernel_form_is_bit_exact(oracle, pname):
    """Tunings never change results: for every parameter set, every combination of the blind-rotate
    form ("br_variant": 4-wave, split, 2-wave), the digit table of the first NTT step ("br_digit_table"; ignored where the
    gadget digits are wider than 7 bits) and the 8-wave form for narrow launches ("br8_max_rotations";
    N = 1024 only) gives the oracle's accumulator, on gate preludes, on the
    sign-wrap edge inputs, on a 200-wide launch and on one random launch wide enough to share CUs."""
    from peba1_amd import api
    pp = {"P128": lambda: api.ParameterSet(128), "P80": lambda: api.ParameterSet(80),
          "P2048": lambda: api.ParameterSet(p2048=True)}[pname]()
    seed = 0xF0 + pp.n
    ks = api.SecretKeySet(pp, seed, device=True)
    oks = oracle.KeySet(oracle.params(pname), seed)
    try:
        r = oracle.Rng(3)
        cts = oks.encrypt(r, [1, 1, 0, 1])
        unit = np.int64(1) << (31 - (10 if pp.N == 1024 else 11))
        lins = [oks.prelude("AND", cts[0], cts[1]), oks.prelude("XNOR", cts[2], cts[3])]
        edge = np.zeros((2, pp.words), dtype=np.int32)
        edge[0, :] = np.int32(unit)                              # every abar = 1
        edge[1, ::2] = np.int32(-unit)                           # abar = 2N-1
        edge[1, 5] = np.int32(-2**31)                            # abar = N
        lins = np.concatenate([np.stack(lins), edge])
        want = []
        for c in range(len(lins)):
            bar = oks.modswitch_ct(lins[c])
            want.append(oks.blind_rotate(bar[:-1], bar[-1]))
        rng = np.random.default_rng(17)
        many = rng.integers(-2**31, 2**31, (600, pp.words), dtype=np.int64).astype(np.int32)
        ref_many = None
        # (form, digit table, 8-wave limit): the small batch runs the 8-wave form where enabled (N = 1024; at N = 2048
        # "br_variant" 0 is the split form too -- the ring has no other)
        forms = [(0, t, b8) for t in (1, 0) for b8 in (1 << 30, 0)] if pp.N == 1024 else []
        # the split form (8 waves, half transforms); table 1 = the most the set allows (stage 0 and the first
        # radix-4 step at Bgbit <= 6), 2 = stage 0 only, 0 = none
        forms += [(2, t, 0) for t in (1, 2, 0)]
        forms += [(4, 1, 0)] if pp.N == 1024 else []            # the 2-wave form
        try:
            for variant, table, br8_max in forms:
                api.set_tuning("br_variant", variant)
                api.set_tuning("br_digit_table", table)
                api.set_tuning("br8_max_rotations", br8_max)
                u, acc = api.kernel_bootstrap_woks(ks, lins, want_acc=True)
                for c in range(len(lins)):
                    assert (acc[c] == want[c]).all(), (pname, variant, table, br8_max, c)
                    assert (u[c] == oks.sample_extract(want[c])).all()
                um = api.kernel_bootstrap_woks(ks, many)
                if ref_many is None:
                    ref_many = um
                    # every row (every fifth at N = 2048) against the oracle's product over one 64-bit prime, on all
                    # host threads (ctypes releases the GIL, the oracle is re-entrant); the other forms must equal this one
                    rows = list(range(0, 600, 1 if pp.N == 1024 else 5)) + [599]
                    with ThreadPoolExecutor(min(32, os.cpu_count() or 8)) as ex:
                        wants = list(ex.map(lambda c: oks.bootstrap_woks(many[c]), rows))
                    for c, w in zip(rows, wants):
                        assert (um[c] == w).all(), (pname, "random row", c)
                assert (um == ref_many).all(), (pname, variant, table, br8_max)
                ul = api.kernel_bootstrap_woks(ks, many[:200])           # 200 workgroups: one per CU, the 8-wave form's range
                assert (ul == ref_many[:200]).all(), (pname, variant, table, br8_max, "200-wide")
            # 600 = one full round of two workgroups per CU + 88: by default the 88 run as a second launch of the
            # 8-wave form ("br_tail8"); as one launch the words are the same
            api.set_tuning("br_variant", -1)
            api.set_tuning("br_digit_table", 1)
            api.set_tuning("br8_max_rotations", 1 << 30)
            for tail8 in (0, 1):
                api.set_tuning("br_tail8", tail8)
                assert (api.kernel_bootstrap_woks(ks, many) == ref_many).all(), (pname, "br_tail8", tail8)
        finally:
            api.set_tuning("br_variant", -1)
            api.set_tuning("br_digit_table", 1)
            api.set_tuning("br8_max_rotations", 1 << 30)
            api.set_tuning("br_tail8", 1)
    finally:
        ks.close()


@pytest.mark.parametrize("shape", [(1024, 4, 8), (2048, 6, 4), (1024, 8, 4)])
def test_custom_gadgets_at_the_limit_of_the_kernel_forms(oracle, shape):
    """gadgets other than the built-in ones.  l = 4 / Bg = 2^8 (N = 1024) is inside every form of its ring;
    N = 2048 / l = 6 / Bg = 2^4 only inside the split form without its eleven-table first step; N = 1024 / l = 8 /
    Bg = 2^4 only inside the split and 2-wave forms.  Whatever form the tunings ask for, the engine runs an admissible
    one (br_forms.hpp) and the accumulators are the oracle's."""
    from peba1_amd import api
    N, l, Bgbit = shape
    n = 24                                            # short blind rotation: the gadget is what is under test
    pp = api.ParameterSet(custom=(n, N, 1, l, Bgbit, 8, 2, 2.0 ** -15, 2.0 ** -25, 0.012467))
    seed = 0xC0 + l
    ks = api.SecretKeySet(pp, seed, device=True)
    oks = oracle.KeySet(oracle.custom_params(n=n, N=N, l=l, Bgbit=Bgbit), seed)
    try:
        rng = np.random.default_rng(l)
        lins = rng.integers(-2**31, 2**31, (300, pp.words), dtype=np.int64).astype(np.int32)
        want = [oks.bootstrap_woks(lins[c]) for c in (0, 1, 299)]
        try:
            for variant, table in ((-1, 1), (0, 1), (2, 1), (2, 0), (4, 1)):
                api.set_tuning("br_variant", variant)
                api.set_tuning("br_digit_table", table)
                u = api.kernel_bootstrap_woks(ks, lins)
                for k, c in enumerate((0, 1, 299)):
                    assert (u[c] == want[k]).all(), (shape, variant, table, c)
                u8 = api.kernel_bootstrap_woks(ks, lins[:2])            # narrow launch: the 8-wave form where admissible
                assert (u8[0] == want[0]).all() and (u8[1] == want[1]).all(), (shape, variant, table, "narrow")
        finally:
            api.set_tuning("br_variant", -1)
            api.set_tuning("br_digit_table", 1)
        # and whole gates through the public API
        r = oracle.Rng(5)
        cts = oks.encrypt(r, [1, 0, 1, 1])
        a = api.CiphertextArray(pp, 2).set_words(cts[0:2])
        b = api.CiphertextArray(pp, 2).set_words(cts[2:4])
        res = api.CiphertextArray(pp, 2)
        api.gate_batch("AND", res, a, b, ks)
        got = res.words()
        for i in range(2):
            assert (got[i] == oks.gate("AND", cts[i], cts[2 + i])).all()
    finally:
        ks.close()


def test_parity_kit_words_are_what_the_gpu_computes():
    """tests/golden/parity_kit: the kit's expected output words (what someone with upstream tfhe compares against
    upstream's exact bootstrap, check_against_upstream.cpp) are reproduced by the HIP path from the kit's inputs under
    the kit's keys -- every case, and the extracted sample of case 0 before its key switch."""
    import json
    from peba1_amd import api, lib
    kit = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "parity_kit")
    with open(os.path.join(kit, "kit.json")) as f:
        meta = json.load(f)
    p = meta["params"]
    rd = lambda name, shape: np.fromfile(os.path.join(kit, name), dtype="<i4").reshape(shape)
    inputs = rd("inputs.i32", (8, p["n"] + 1))
    expected = rd("expected.i32", (len(meta["cases"]), p["n"] + 1))
    extracted = rd("extracted.i32", (p["k"] * p["N"] + 1,))
    pp = api.ParameterSet(custom=(p["n"], p["N"], p["k"], p["l"], p["Bgbit"], p["ks_t"], p["ks_basebit"],
                                  p["ks_stdev"], p["bk_stdev"], p["max_stdev"]))
    ks = api.SecretKeySet(pp, meta["key_seed"], device=True)
    try:
        assert hashlib.sha256(np.ascontiguousarray(ks.bk()).tobytes()).hexdigest() == meta["sha256"]["bk.i32"]
        L = lib.load()
        ct = api.CiphertextArray(pp, 8).set_words(inputs)
        out = api.CiphertextArray(pp, len(meta["cases"]))
        gates = {"AND": L.bootsAND, "XOR": L.bootsXOR, "OR": L.bootsOR, "XNOR": L.bootsXNOR, "NAND": L.bootsNAND}
        for i, c in enumerate(meta["cases"]):
            if c[0] == "MUX":
                L.bootsMUX(out.at(i), ct.at(c[1]), ct.at(c[2]), ct.at(c[3]), ks.cloud)
            else:
                gates[c[0]](out.at(i), ct.at(c[1]), ct.at(c[2]), ks.cloud)
        got = out.words()
        for i, c in enumerate(meta["cases"]):
            assert (got[i] == expected[i]).all(), c
        assert list(out.decrypt(ks)) == meta["expected_bits"]
        c0 = meta["cases"][0]
        sa = {"AND": (1, 1, -meta["mu"])}[c0[0]]
        lin = (sa[0] * inputs[c0[1]].astype(np.int64) + sa[1] * inputs[c0[2]].astype(np.int64))
        lin[-1] += sa[2]
        lin = (lin % (1 << 32)).astype(np.uint32).view(np.int32)
        u = api.kernel_bootstrap_woks(ks, lin[None, :])
        assert (u[0] == extracted).all()
    finally:
        ks.close()
