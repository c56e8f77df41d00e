#!/usr/bin/env python3
"""Exact-integer model of the radix-4 butterflies with merged Montgomery reductions used by
peba1_amd/csrc/ntt_wave.hpp (round 2), for N = 1024 and 2048.  Development aid: checks
  * the radix-4 formulas and their twiddle tables against the textbook radix-2 stage loop,
  * the pass / pair / single structure (4+4+2 and 5+5+1 stages) and which index bits select twiddles,
  * the signed-lazy magnitude bounds (forward: no reduction; inverse: which steps renormalise)
    by worst-case interval propagation AND by running extreme inputs through the emulated
    32-bit arithmetic (every intermediate asserted to fit int32 / int64).

Forward pair (stages s, s+1; x0, x1 = partner in stage s+1, x2 = partner in stage s, x3 = both):
    A = redc(x2 w1)            S = redc(x1 w2 + x3 w1w2)         S' = redc(x1 w3 - x3 ... ) see code
    y0 = (x0 + A) + S   y1 = (x0 + A) - S   y2 = (x0 - A) + S'   y3 = (x0 - A) - S'
    11 multiplier-class + 6 add instructions for what two radix-2 stages do in 12 + 8.
Inverse pair (stages s+1 then s, Gentleman-Sande):
    s0 = x0 + x1, s1 = x2 + x3, d0 = x0 - x1, d1 = x2 - x3
    y0 = s0 + s1 [renormalised: redc((s0 + s1) R)]   y2 = redc((s0 - s1) iw1)
    y1 = redc(d0 iw2 + d1 iw3)                        y3 = redc(d0 iw1 iw2 - d1 iw1 iw3)
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from ntt_model import P0, P1, tables, ref_fwd, ref_inv  # noqa: E402

R = 1 << 32


def redc(T, P):
    """signed Montgomery reduction as the kernel does it: |result| <= |T|/2^32 + P/2"""
    assert -(1 << 63) <= T < (1 << 63), "64-bit accumulator overflow"
    pinv = (-pow(P, -1, R)) % R
    m = ((T & 0xFFFFFFFF) * pinv) & 0xFFFFFFFF
    if m >= 1 << 31:
        m -= 1 << 32
    U = T + m * P
    assert U % R == 0
    r = U >> 32
    assert -(1 << 31) <= r < (1 << 31)
    return r


def i32(v):
    assert -(1 << 31) <= v < (1 << 31), f"int32 overflow {v}"
    return v


def passes(logn):
    rb = logn - 6
    return [list(range(0, rb)), list(range(rb, 2 * rb)), list(range(2 * rb, logn))]


def fwd_steps(logn):
    """[(s,) or (s, s+1)] ascending, pairs from the start of each pass"""
    out = []
    for p in passes(logn):
        i = 0
        while i < len(p):
            if i + 1 < len(p):
                out.append((p[i], p[i + 1])); i += 2
            else:
                out.append((p[i],)); i += 1
    return out


def inv_steps(logn):
    """descending, pairs from the end of each pass: (s+1, s) listed as (s, s+1)"""
    out = []
    for p in reversed(passes(logn)):
        i = len(p) - 1
        while i >= 0:
            if i - 1 >= 0:
                out.append((p[i - 1], p[i])); i -= 2
            else:
                out.append((p[i],)); i -= 1
    return out


def mont(v, P):
    return v * R % P


def forward(x, W, P, logn, track=None):
    N = 1 << logn
    x = [i32(int(v)) for v in x]
    for step in fwd_steps(logn):
        if len(step) == 1:
            s = step[0]
            ln = N >> (s + 1)
            for t in range(1 << s):
                w = mont(W[(1 << s) + t], P)
                for j in range(2 * t * ln, 2 * t * ln + ln):
                    r = redc(x[j + ln] * w, P)
                    x[j], x[j + ln] = i32(x[j] + r), i32(x[j] - r)
        else:
            s = step[0]
            ln = N >> (s + 2)                    # quarter block
            for t in range(1 << s):
                w1 = W[(1 << s) + t]
                w2, w3 = W[(2 << s) + 2 * t], W[(2 << s) + 2 * t + 1]
                m1, m2, m12, m3, m13 = mont(w1, P), mont(w2, P), mont(w1 * w2, P), mont(w3, P), mont(w1 * w3, P)
                base = 4 * t * ln
                for j in range(base, base + ln):
                    x0, x1, x2, x3 = x[j], x[j + ln], x[j + 2 * ln], x[j + 3 * ln]
                    A = redc(x2 * m1, P)
                    S = redc(x1 * m2 + x3 * m12, P)
                    Sp = redc(x1 * m3 - x3 * m13, P)     # kernel stores P - w1w3 and adds
                    u, v = i32(x0 + A), i32(x0 - A)
                    x[j], x[j + ln], x[j + 2 * ln], x[j + 3 * ln] = i32(u + S), i32(u - S), i32(v + Sp), i32(v - Sp)
        if track is not None:
            track.append(max(abs(v) for v in x) / P)
    return x


def inverse(x, IW, P, logn, renorm, track=None):
    """unscaled inverse; renorm[k] says whether step k renormalises its plain sums"""
    N = 1 << logn
    x = [i32(int(v)) for v in x]
    rmod = mont(1, P)
    for k, step in enumerate(inv_steps(logn)):
        if len(step) == 1:
            s = step[0]
            ln = N >> (s + 1)
            for t in range(1 << s):
                w = mont(IW[(1 << s) + t], P)
                for j in range(2 * t * ln, 2 * t * ln + ln):
                    a, b = x[j], x[j + ln]
                    sm = i32(a + b)
                    x[j] = redc(sm * rmod, P) if renorm[k] else sm
                    x[j + ln] = redc(i32(a - b) * w, P)
        else:
            s = step[0]
            ln = N >> (s + 2)
            for t in range(1 << s):
                w1 = IW[(1 << s) + t]
                w2, w3 = IW[(2 << s) + 2 * t], IW[(2 << s) + 2 * t + 1]
                m1, m2, m3, m12, m13n = mont(w1, P), mont(w2, P), mont(w3, P), mont(w1 * w2, P), mont(P - w1 * w3 % P, P)
                base = 4 * t * ln
                for j in range(base, base + ln):
                    x0, x1, x2, x3 = x[j], x[j + ln], x[j + 2 * ln], x[j + 3 * ln]
                    s0, s1, d0, d1 = i32(x0 + x1), i32(x2 + x3), i32(x0 - x1), i32(x2 - x3)
                    y0 = i32(s0 + s1)
                    if renorm[k]:
                        y0 = redc(y0 * rmod, P)
                    y2 = redc(i32(s0 - s1) * m1, P)
                    y1 = redc(d0 * m2 + d1 * m3, P)
                    y3 = redc(d0 * m12 + d1 * m13n, P)
                    x[j], x[j + ln], x[j + 2 * ln], x[j + 3 * ln] = y0, y1, y2, y3
        if track is not None:
            track.append(max(abs(v) for v in x) / P)
    return x


def fwd_bound(logn, b_in, P):
    """worst-case |value| / P after every forward step (interval arithmetic of the formulas)"""
    q = P / R
    b, out = b_in, []
    for step in fwd_steps(logn):
        if len(step) == 1:
            b = b + (b * q + 0.5)
        else:
            b = b + (b * q + 0.5) + (2 * b * q + 0.5)
        out.append(b)
    return out


def inv_schedule(logn, b_in, P, limit=(1 << 31) - 1):
    """greedy: a step renormalises its sums iff the NEXT step could not take them (every
    difference / sum of four inputs must stay below 2^31), and always at the last step"""
    q = P / R
    steps = inv_steps(logn)
    b, renorm, bounds = b_in, [], []
    for k, step in enumerate(steps):
        fan = 4 if len(step) == 2 else 2
        assert fan * b * P <= limit, f"step {k}: inputs {b:.2f}P too large"
        small = fan * b * q + 0.5                           # every reduced output
        big = fan * b
        last = k == len(steps) - 1
        nxt = 0 if last else (4 if len(steps[k + 1]) == 2 else 2)
        r = last or nxt * max(big, small) * P > limit
        renorm.append(r)
        b = max(small, fan * b * q + 0.5 if r else big)
        bounds.append(b)
    return renorm, bounds


def main():
    rng = np.random.default_rng(1)
    for logn in (10, 11):
        N = 1 << logn
        print(f"N = {N}: forward steps {fwd_steps(logn)}  inverse steps {inv_steps(logn)}")
        for P in (P0, P1):
            W, IW = tables(P, N)
            # forward, digit-sized inputs (and extreme ones), against the radix-2 reference
            for x in (rng.integers(-2048, 2048, N), np.full(N, 2048), np.full(N, -2048),
                      rng.integers(-(P - 1), P, N)):
                tr = []
                got = forward(x, W, P, logn, tr)
                ref = ref_fwd([int(v) % P for v in x], W, P)
                assert [g % P for g in got] == ref, "forward mismatch"
            fb = fwd_bound(logn, (P - 1) / P, P)
            assert tr[-1] <= fb[-1] + 1e-9
            # inverse on inputs at the magnitude the kernels feed (|t| < 4P), extreme signs
            renorm, ib = inv_schedule(logn, 4.0, P)
            ninv = pow(N, P - 2, P)
            for y in (rng.integers(-4 * P + 1, 4 * P, N), np.full(N, 4 * P - 1), np.full(N, -(4 * P - 1)),
                      np.where(rng.integers(0, 2, N) > 0, 4 * P - 1, -(4 * P - 1))):
                tr = []
                back = inverse(y, IW, P, logn, renorm, tr)
                ref = ref_inv([int(v) % P for v in y], IW, P)
                assert [b * ninv % P for b in back] == ref, "inverse mismatch"
                assert all(t <= b + 1e-9 for t, b in zip(tr, ib)), (tr, ib)
            assert max(abs(v) for v in back) < P
            print(f"  P = {P}: forward bound after each step (inputs < P): {[round(b, 2) for b in fb]}")
            print(f"               digits (|x| <= 2^11): {[round(b, 2) for b in fwd_bound(logn, 2048 / P, P)]}")
            print(f"               inverse renormalising steps {[int(r) for r in renorm]}  bounds {[round(b, 2) for b in ib]}")
    print("ok")


if __name__ == "__main__":
    main()
