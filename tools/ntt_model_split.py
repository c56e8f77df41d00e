#!/usr/bin/env python3
"""Model of the split transforms of blind_rotate_split_kernel (peba1_amd/csrc/kernels.hip): a
Cooley-Tukey negacyclic NTT cut after stage 0 into two independent half-size transforms that run
on sub-tree twiddle tables laid out like a stand-alone N/2-point table (engine.cpp make_twiddles).
Checks, against the textbook transform of tools/ntt_model.py:
  * the wave layouts of WaveNtt<LOGN-1> for LOGN - 1 = 9 (tools/ntt_model_generic.py covers 10, 11),
  * table entry 2^s' + t' of half h = full-size entry (2 + h) 2^s' + t',
  * half h's outputs are slots [h N/2, (h+1) N/2) of the full spectrum in the same order,
  * where a lane of the half transform finds its 16-byte groups in a row of the key image,
  * the inverse: two half inverses, then (a0 + a1, (a0 - a1) W^-1[1]).
"""
import numpy as np
from ntt_model import tables, ref_fwd, ref_inv, P0, P1
import ntt_model_generic as g


def sub_table(W, h, M):
    T = [0] * M
    for e in range(1, M):
        top = 1 << (e.bit_length() - 1)
        T[e] = W[(2 + h) * top + (e - top)]
    return T


def main():
    rng = np.random.default_rng(5)
    for LOGN in (10, 11):
        N, M = 1 << LOGN, 1 << (LOGN - 1)
        RBs, RS, LCs = g.params(LOGN - 1)
        for P in (P0, P1):
            W, IW = tables(P, N)
            x = [int(v) for v in rng.integers(0, P, N)]
            ref = ref_fwd(x, W, P)
            full_spec = list(ref)
            for h in (0, 1):
                sgn = 1 if h == 0 else -1
                xh = [(x[j] + sgn * W[1] * x[j + M]) % P for j in range(M)]
                Th = sub_table(W, h, M)
                # textbook half transform on the sub-table == slots of the full spectrum
                assert ref_fwd(xh, Th, P) == full_spec[h * M:(h + 1) * M], (LOGN, P, h)
                # the wave code path (layouts, twiddle formulas) on the sub-table
                Xw = g.wave_fwd(LOGN - 1, xh, Th, P)
                got = [Xw[j // RS, j % RS] for j in range(M)]
                assert got == full_spec[h * M:(h + 1) * M]
                # key image addressing: word (g*256 + lane'*4 + e) of a row holds register 4g+e of lane' in the
                # full layout L2 (slot j = 2 RS lane' + reg)
                RF = 2 * RS
                img = [None] * N
                for lane_f in range(64):
                    for reg in range(RF):
                        img[(reg >> 2) * 256 + lane_f * 4 + (reg & 3)] = RF * lane_f + reg
                for lane in range(64):
                    lane_off = (RS // 4) * (lane & 1) * 64 + h * 32 + (lane >> 1)
                    for gq in range(RS // 4):
                        for e in range(4):
                            slot = img[(lane_off + gq * 64) * 4 + e]
                            assert slot == h * M + RS * lane + 4 * gq + e, (LOGN, h, lane, gq, e, slot)
            # inverse (unscaled): half inverses then stage 0
            y = [int(v) for v in rng.integers(0, P, N)]
            want = ref_inv(y, IW, P)                      # scaled by N^-1 in ntt_model
            ninv = pow(N, P - 2, P)
            a = []
            for h in (0, 1):
                ITh = sub_table(IW, h, M)
                ah = ref_inv(y[h * M:(h + 1) * M], ITh, P)          # scaled by M^-1
                a.append([v * M % P for v in ah])                    # unscaled
            out = [(a[0][j] + a[1][j]) % P for j in range(M)] + [(a[0][j] - a[1][j]) * IW[1] % P for j in range(M)]
            assert [v * ninv % P for v in out] == want, (LOGN, P)
        print("LOGN", LOGN, "split transforms ok (half =", M, "points)")


if __name__ == "__main__":
    main()
