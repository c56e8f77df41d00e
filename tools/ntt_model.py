"""Lane/register-level numpy model of the wave64 negacyclic NTT used by the HIP
blind-rotate kernel (peba1_amd/csrc/ntt_wave.hpp).  Development aid: checks the
layout / twiddle-index formulas against a textbook stage-loop NTT.

Layouts for N = 1024 (index j = b9..b0), 64 lanes x 16 registers:
  L0: reg = b9..b6, lane = b5..b0            (natural: j = 64*reg + lane)
  L1: reg = b5..b2, lane = (b9..b6, b1, b0)
  L2: reg = b3..b0, lane = b9..b4            (j = 16*lane + reg)
forward (CT, natural in -> bit-reversed out): stages 0-3 in L0, 4-7 in L1, 8-9 in L2
inverse (GS): stages 9-8 in L2, 7-4 in L1, 3-0 in L0.
"""
import numpy as np

P0 = 134111233
P1 = 134176769


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def find_psi(P, N):
    # primitive 2N-th root of unity mod P
    from sympy import primitive_root
    g = primitive_root(P)
    psi = pow(g, (P - 1) // (2 * N), P)
    assert pow(psi, N, P) == P - 1
    return psi


def tables(P, N):
    logn = N.bit_length() - 1
    psi = find_psi(P, N)
    ipsi = pow(psi, P - 2, P)
    W = [0] * N
    IW = [0] * N
    a = b = 1
    for i in range(N):
        j = bitrev(i, logn)
        W[j] = a
        IW[j] = b
        a = a * psi % P
        b = b * ipsi % P
    return W, IW


def ref_fwd(x, W, P):
    x = [int(v) for v in x]
    N = len(x)
    m, ln = 1, N // 2
    while m < N:
        for i in range(m):
            w = W[m + i]
            for j in range(2 * i * ln, 2 * i * ln + ln):
                u, v = x[j], x[j + ln] * w % P
                x[j], x[j + ln] = (u + v) % P, (u - v) % P
        m, ln = m * 2, ln // 2
    return x


def ref_inv(x, IW, P):
    x = [int(v) for v in x]
    N = len(x)
    m, ln = N // 2, 1
    while m >= 1:
        for i in range(m):
            w = IW[m + i]
            for j in range(2 * i * ln, 2 * i * ln + ln):
                u, v = x[j], x[j + ln]
                x[j], x[j + ln] = (u + v) % P, (u - v) * w % P
        m, ln = m // 2, ln * 2
    ninv = pow(N, P - 2, P)
    return [v * ninv % P for v in x]


# ---- wave model (N = 1024) -------------------------------------------------
def j_of(layout, lane, reg):
    if layout == 0:
        return 64 * reg + lane
    if layout == 1:
        return ((lane >> 2) << 6) | (reg << 2) | (lane & 3)
    return 16 * lane + reg


def transpose(X, src, dst):
    Y = np.zeros_like(X)
    pos = {}
    for lane in range(64):
        for reg in range(16):
            pos[j_of(src, lane, reg)] = X[lane, reg]
    for lane in range(64):
        for reg in range(16):
            Y[lane, reg] = pos[j_of(dst, lane, reg)]
    return Y


def tw_index(stage, layout, lane, reg):
    """Index into W for the butterfly whose 'a' element sits in (lane, reg)."""
    j = j_of(layout, lane, reg)
    return (1 << stage) + (j >> (10 - stage))


def tw_index_formula(stage, lane, reg):
    """The closed forms the kernel uses (must equal tw_index)."""
    if stage < 4:      # L0: top `stage` bits of reg
        return (1 << stage) + (reg >> (4 - stage))
    if stage < 8:      # L1: a = lane>>2 supplies b9..b6, then top (stage-4) bits of reg
        return (1 << stage) + ((lane >> 2) << (stage - 4)) + (reg >> (8 - stage))
    # L2: lane supplies b9..b4, then top (stage-6) bits of reg
    return (1 << stage) + (lane << (stage - 6)) + (reg >> (10 - stage))


def wave_fwd(x, W, P):
    X = np.zeros((64, 16), dtype=object)
    for lane in range(64):
        for reg in range(16):
            X[lane, reg] = int(x[j_of(0, lane, reg)])
    layout = 0
    for stage in range(10):
        if stage == 4:
            X = transpose(X, 0, 1); layout = 1
        if stage == 8:
            X = transpose(X, 1, 2); layout = 2
        # register bit paired at this stage
        rb = {0: 3, 1: 2, 2: 1, 3: 0, 4: 3, 5: 2, 6: 1, 7: 0, 8: 1, 9: 0}[stage]
        for lane in range(64):
            for reg in range(16):
                if reg & (1 << rb):
                    continue
                ti = tw_index_formula(stage, lane, reg)
                assert ti == tw_index(stage, layout, lane, reg)
                a, b = X[lane, reg], X[lane, reg | (1 << rb)]
                t = b * W[ti] % P
                X[lane, reg], X[lane, reg | (1 << rb)] = (a + t) % P, (a - t) % P
    return X  # layout L2, bit-reversed-order semantics


def wave_inv(X, IW, P):
    X = X.copy()
    layout = 2
    for stage in range(9, -1, -1):
        if stage == 7:
            X = transpose(X, 2, 1); layout = 1
        if stage == 3:
            X = transpose(X, 1, 0); layout = 0
        rb = {0: 3, 1: 2, 2: 1, 3: 0, 4: 3, 5: 2, 6: 1, 7: 0, 8: 1, 9: 0}[stage]
        for lane in range(64):
            for reg in range(16):
                if reg & (1 << rb):
                    continue
                ti = tw_index_formula(stage, lane, reg)
                a, b = X[lane, reg], X[lane, reg | (1 << rb)]
                X[lane, reg], X[lane, reg | (1 << rb)] = (a + b) % P, (a - b) * IW[ti] % P
    return X  # layout L0, natural order, unscaled (x N)


def negacyclic_ref(a, b, mod):
    N = len(a)
    r = [0] * N
    for i in range(N):
        for j in range(N):
            k = i + j
            if k < N:
                r[k] = (r[k] + int(a[i]) * int(b[j])) % mod
            else:
                r[k - N] = (r[k - N] - int(a[i]) * int(b[j])) % mod
    return r


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    N = 1024
    for P in (P0, P1):
        W, IW = tables(P, N)
        x = rng.integers(0, P, N)
        ref = ref_fwd(x, W, P)
        Xw = wave_fwd(x, W, P)
        got = [Xw[j >> 4, j & 15] for j in range(N)]
        assert got == ref, "forward mismatch"
        back = wave_inv(Xw, IW, P)
        ninv = pow(N, P - 2, P)
        for lane in range(64):
            for reg in range(16):
                assert back[lane, reg] * ninv % P == x[64 * reg + lane]
        # convolution theorem on a small-digit polynomial
        d = rng.integers(-64, 64, N)
        t = rng.integers(0, P, N)
        fa = ref_fwd([int(v) % P for v in d], W, P)
        fb = ref_fwd(t, W, P)
        prod = ref_inv([u * v % P for u, v in zip(fa, fb)], IW, P)
        assert prod == negacyclic_ref(d, t, P)
        print("P =", P, "ok")
