"""Generic (LOGN = 10 or 11) lane/register model of the wave64 NTT layouts used by
peba1_amd/csrc/ntt_wave.hpp: checks the transposes' address maps and the twiddle-index
formulas of the templated kernel code against a textbook NTT."""
import sys
import numpy as np
from ntt_model import tables, ref_fwd, ref_inv, P0, P1


def params(LOGN):
    RB = LOGN - 6
    REGS = 1 << RB
    LC = LOGN - 2 * RB
    return RB, REGS, LC


def j_of(LOGN, layout, lane, reg):
    RB, REGS, LC = params(LOGN)
    if layout == 0:
        return 64 * reg + lane
    if layout == 1:   # reg = bits [LOGN-RB-1 .. LC], lane = (top RB bits, low LC bits)
        return ((lane >> LC) << (LOGN - RB)) | (reg << LC) | (lane & ((1 << LC) - 1))
    return REGS * lane + reg


def t1_addr(LOGN, lane, reg):      # written from L0, read as rows from L1
    RB, REGS, LC = params(LOGN)
    lane1 = (reg << LC) | (lane & ((1 << LC) - 1))
    reg1 = lane >> LC
    return lane1 * REGS + reg1 + 4 * (lane1 >> LC)


def t2_addr(LOGN, lane, reg):      # written from L1, read as rows from L2
    RB, REGS, LC = params(LOGN)
    j = j_of(LOGN, 1, lane, reg)
    return j + 4 * (j >> (LOGN - RB))


def row_base(LOGN, lane):
    RB, REGS, LC = params(LOGN)
    return lane * REGS + 4 * (lane >> LC)


def transpose_via_lds(LOGN, X, which):
    RB, REGS, LC = params(LOGN)
    scr = {}
    for lane in range(64):
        for reg in range(REGS):
            a = t1_addr(LOGN, lane, reg) if which == 1 else t2_addr(LOGN, lane, reg)
            assert a not in scr and a < (1 << LOGN) + 4 * (64 >> LC)
            scr[a] = X[lane, reg]
    Y = np.zeros_like(X)
    for lane in range(64):
        for reg in range(REGS):
            Y[lane, reg] = scr[row_base(LOGN, lane) + reg]
    return Y


def stage_info(LOGN, s):
    """(layout, register bit paired, twiddle index as a function of lane, reg)"""
    RB, REGS, LC = params(LOGN)
    if s < RB:
        rb = RB - 1 - s
        return 0, rb, lambda lane, reg: (1 << s) + (reg >> (rb + 1))
    if s < 2 * RB:
        rb = 2 * RB - 1 - s
        return 1, rb, lambda lane, reg: (1 << s) + ((lane >> LC) << (s - RB)) + (reg >> (rb + 1))
    rb = LOGN - 1 - s
    return 2, rb, lambda lane, reg: (1 << s) + (lane << (s - 6)) + (reg >> (rb + 1))


def wave_fwd(LOGN, x, W, P):
    RB, REGS, LC = params(LOGN)
    X = np.zeros((64, REGS), dtype=object)
    for lane in range(64):
        for reg in range(REGS):
            X[lane, reg] = int(x[64 * reg + lane])
    layout = 0
    for s in range(LOGN):
        lay, rb, twi = stage_info(LOGN, s)
        if lay != layout:
            X = transpose_via_lds(LOGN, X, lay)
            layout = lay
        for lane in range(64):
            for reg in range(REGS):
                if reg & (1 << rb):
                    continue
                ti = twi(lane, reg)
                j = j_of(LOGN, layout, lane, reg)
                assert ti == (1 << s) + (j >> (LOGN - s))
                assert j_of(LOGN, layout, lane, reg | (1 << rb)) == j + (1 << (LOGN - 1 - s))
                a, b = X[lane, reg], X[lane, reg | (1 << rb)]
                t = b * W[ti] % P
                X[lane, reg], X[lane, reg | (1 << rb)] = (a + t) % P, (a - t) % P
    return X


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for LOGN in (10, 11):
        N = 1 << LOGN
        RB, REGS, LC = params(LOGN)
        for P in (P0, P1):
            W, IW = tables(P, N)
            x = rng.integers(0, P, N)
            ref = ref_fwd(x, W, P)
            Xw = wave_fwd(LOGN, x, W, P)
            got = [Xw[j // REGS, j % REGS] for j in range(N)]
            assert got == ref, (LOGN, P)
        print("LOGN", LOGN, "layouts and twiddle formulas ok")
