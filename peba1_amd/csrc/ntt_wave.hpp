// ntt_wave.hpp -- one wave64 computes one N-point negacyclic NTT (N = 1024 or 2048)
// modulo a 27-bit prime: N/64 coefficients per lane in registers, two LDS transposes.
//
// Index j has LOGN bits; RB = LOGN-6 register bits, REGS = 2^RB coefficients per lane,
// LC = LOGN-2*RB (2 for N=1024, 1 for N=2048).  Register/lane layouts (validated by
// tools/ntt_model.py and tools/ntt_model_generic.py):
//   L0: reg = top RB bits,        lane = low 6 bits              (natural: j = 64*reg + lane)
//   L1: reg = next RB bits,       lane = (top RB bits, low LC bits)
//   L2: reg = low RB bits,        lane = top 6 bits              (j = REGS*lane + reg)
// forward (Cooley-Tukey, natural in -> bit-reversed out, psi-merged twiddles):
//   RB stages in L0, transpose, RB stages in L1, transpose, LC stages in L2
//   (4+4+2 for N=1024, 5+5+1 for N=2048).
// inverse (Gentleman-Sande) runs the same path backwards and ends in L0.
// A butterfly always pairs two registers of one lane; the twiddle of stage s is
// W[2^s + (top s bits of j)]: lane-uniform in L0 (scalar loads), per-lane but
// contiguous in L1/L2 (dwordx2/x4 loads).
//
// Signed lazy arithmetic (values are int32 representatives, not canonical):
//   Montgomery product r = b*w/R mod P with |r| < P for ANY |b| < 2^31, w in [0,P)
//   (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32);
//   forward butterfly (a,b) -> (a + r, a - r): 5 instructions, magnitudes grow by
//   P per stage: digits (|d| <= 2^11) end below LOGN*P + 2^11 < 2^31, no reductions;
//   inverse butterfly (a,b) -> (a + b, (a - b)*w): sums double, so the sum is
//   renormalised (times R mod P, 3 more instructions) only at the stages where the
//   next doubling would pass 16P < 2^31 (16*P1 = 2,146,828,304), and at the last:
//   stages 8,4,0 for N=1024 and 9,5,1,0 for N=2048 (inv_renorm_mask below), for any
//   input below 4P.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_field.hpp"

namespace tfhe_hip {

struct PrimeCtx {
    uint32_t P;        // prime
    uint32_t pinv;     // -P^-1 mod 2^32
    uint32_t rmod;     // R mod P: Montgomery multiplication by it is the identity (renormalisation)
    const uint32_t *__restrict__ wf;   // forward twiddles psi^brv(i) * R mod P, [N]
    const uint32_t *__restrict__ wi;   // inverse twiddles psi^-brv(i) * R mod P, [N]
};

// signed Montgomery reduction: T*R^-1 mod P, |result| <= |T|/2^32 + P/2
__device__ __forceinline__ int32_t mont_redc(int64_t T, uint32_t P, uint32_t pinv) {
    const int32_t m = (int32_t)((uint32_t)T * pinv);           // m = -T/P mod 2^32, taken signed
    const int64_t U = T + (int64_t)m * (int64_t)(int32_t)P;    // divisible by 2^32
    return (int32_t)(U >> 32);
}
// b*w*R^-1 mod P with |result| < P for any |b| < 2^31, 0 <= w < P
__device__ __forceinline__ int32_t mont_mul(int32_t b, uint32_t w, uint32_t P, uint32_t pinv) {
    return mont_redc((int64_t)b * (int64_t)(int32_t)w, P, pinv);
}
// unsigned form, operands below P (CRT only): result in [0,2P)
__device__ __forceinline__ uint32_t mont_mul_u(uint32_t b, uint32_t w, uint32_t P, uint32_t pinv) {
    const uint64_t T = (uint64_t)b * (uint64_t)w;
    const uint32_t m = (uint32_t)T * pinv;
    const uint64_t U = T + (uint64_t)m * (uint64_t)P;
    return (uint32_t)(U >> 32);
}
// x in [0,2B) -> [0,B)
__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t B) { return min(x, x - B); }
// representative in (-P,P) -> canonical [0,P)
__device__ __forceinline__ uint32_t canon(int32_t x, uint32_t P) { return min((uint32_t)x, (uint32_t)x + P); }

// LDS is processed in issue order for one wave, so a wave-private transpose
// needs no barrier; this only stops the compiler from reordering across it.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void ct_bfly(int32_t &a, int32_t &b, uint32_t w, const PrimeCtx &c) {
    const int32_t t = mont_mul(b, w, c.P, c.pinv);
    const int32_t a0 = a;
    a = a0 + t;
    b = a0 - t;
}
template <bool RENORM>
__device__ __forceinline__ void gs_bfly(int32_t &a, int32_t &b, uint32_t w, const PrimeCtx &c) {
    const int32_t a0 = a, b0 = b;
    a = RENORM ? mont_mul(a0 + b0, c.rmod, c.P, c.pinv) : a0 + b0;
    b = mont_mul(a0 - b0, w, c.P, c.pinv);
}

// stages of the inverse transform whose sums are renormalised (bit s set), for inputs < 4P
constexpr uint32_t inv_renorm_mask(int logn) {
    uint32_t mask = 0;
    int bound = 4;                       // in units of P
    for (int s = logn - 1; s >= 0; --s) {
        // this stage forms a +- b < 2*bound*P (must stay <= 16P); without renormalisation the
        // next stage would form sums below 4*bound*P
        if (s == 0 || 4 * bound > 16) { mask |= 1u << s; bound = 1; }
        else bound *= 2;
    }
    return mask;
}
static_assert(inv_renorm_mask(10) == ((1u << 8) | (1u << 4) | 1u), "N=1024 renormalisation schedule");
static_assert(inv_renorm_mask(11) == ((1u << 9) | (1u << 5) | (1u << 1) | 1u), "N=2048 renormalisation schedule");

template <int LOGN>
struct WaveNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int RB = LOGN - 6;              // register bits
    static constexpr int REGS = 1 << RB;             // coefficients per lane
    static constexpr int LC = LOGN - 2 * RB;         // stages of the last pass
    static constexpr int SCRATCH_WORDS = N + 4 * (64 >> LC);   // rows of REGS words, padded
    static_assert(LOGN == 10 || LOGN == 11, "wave NTT is laid out for N = 1024 or 2048");

    // ---- transposes through wave-private LDS scratch ---------------------------
    // row of `lane` in the layout being read: REGS contiguous words
    static __device__ __forceinline__ int row_base(int lane) { return lane * REGS + 4 * (lane >> LC); }
    // where (lane, reg) of L0 lands so that L1 reads rows
    static __device__ __forceinline__ int t1_l0_addr(int lane, int reg) {
        const int lane1 = (reg << LC) | (lane & ((1 << LC) - 1));
        return lane1 * REGS + (lane >> LC) + 4 * (lane1 >> LC);
    }
    // where (lane, reg) of L1 lands so that L2 reads rows: natural order j, padded
    static __device__ __forceinline__ int t2_l1_addr(int lane, int reg) {
        const int j = ((lane >> LC) << (LOGN - RB)) | (reg << LC) | (lane & ((1 << LC) - 1));
        return j + 4 * (j >> (LOGN - RB));
    }
    template <typename T>
    static __device__ __forceinline__ void read_row(T (&x)[REGS], const uint32_t *scr, int lane) {
        const uint4 *p = reinterpret_cast<const uint4 *>(scr + row_base(lane));
#pragma unroll
        for (int g = 0; g < REGS / 4; ++g) {
            const uint4 v = p[g];
            x[4 * g] = (T)v.x; x[4 * g + 1] = (T)v.y; x[4 * g + 2] = (T)v.z; x[4 * g + 3] = (T)v.w;
        }
    }
    template <typename T>
    static __device__ __forceinline__ void write_row(const T (&x)[REGS], uint32_t *scr, int lane) {
        uint4 *p = reinterpret_cast<uint4 *>(scr + row_base(lane));
#pragma unroll
        for (int g = 0; g < REGS / 4; ++g)
            p[g] = make_uint4((uint32_t)x[4 * g], (uint32_t)x[4 * g + 1], (uint32_t)x[4 * g + 2], (uint32_t)x[4 * g + 3]);
    }

    // CNT consecutive twiddles starting at w (CNT = 1, 2, 4, 8, 16; w aligned to CNT words)
    template <int CNT>
    static __device__ __forceinline__ void load_tw(uint32_t (&t)[CNT], const uint32_t *__restrict__ w) {
        if constexpr (CNT == 1) {
            t[0] = w[0];
        } else if constexpr (CNT == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(w);
            t[0] = v.x; t[1] = v.y;
        } else {
#pragma unroll
            for (int g = 0; g < CNT / 4; ++g) {
                const uint4 v = reinterpret_cast<const uint4 *>(w)[g];
                t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
            }
        }
    }

    // One stage whose butterflies pair register bit RBIT of a lane; `tw` holds the
    // REGS >> (RBIT+1) twiddles this lane needs, indexed by the register bits above RBIT.
    template <int RBIT>
    static __device__ __forceinline__ void fwd_stage(int32_t (&x)[REGS], const uint32_t (&tw)[REGS >> (RBIT + 1)],
                                                     const PrimeCtx &c) {
        constexpr int h = 1 << RBIT;
#pragma unroll
        for (int r = 0; r < REGS; ++r)
            if (!(r & h)) ct_bfly(x[r], x[r | h], tw[r >> (RBIT + 1)], c);
    }
    template <int RBIT, bool RENORM>
    static __device__ __forceinline__ void inv_stage(int32_t (&x)[REGS], const uint32_t (&tw)[REGS >> (RBIT + 1)],
                                                     const PrimeCtx &c) {
        constexpr int h = 1 << RBIT;
#pragma unroll
        for (int r = 0; r < REGS; ++r)
            if (!(r & h)) gs_bfly<RENORM>(x[r], x[r | h], tw[r >> (RBIT + 1)], c);
    }

    // stage S of the forward transform in the layout it belongs to
    template <int S>
    static __device__ __forceinline__ void fwd(int32_t (&x)[REGS], const PrimeCtx &c, int lane) {
        if constexpr (S < RB) {                                   // L0: lane-uniform twiddles
            constexpr int RBIT = RB - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
#pragma unroll
            for (int e = 0; e < (REGS >> (RBIT + 1)); ++e) tw[e] = c.wf[(1 << S) + e];
            fwd_stage<RBIT>(x, tw, c);
        } else if constexpr (S < 2 * RB) {                        // L1
            constexpr int RBIT = 2 * RB - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
            load_tw<(REGS >> (RBIT + 1))>(tw, c.wf + (1 << S) + ((lane >> LC) << (S - RB)));
            fwd_stage<RBIT>(x, tw, c);
        } else {                                                  // L2
            constexpr int RBIT = LOGN - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
            load_tw<(REGS >> (RBIT + 1))>(tw, c.wf + (1 << S) + (lane << (S - 6)));
            fwd_stage<RBIT>(x, tw, c);
        }
    }
    template <int S>
    static __device__ __forceinline__ void inv(int32_t (&x)[REGS], const PrimeCtx &c, int lane) {
        constexpr bool RN = (inv_renorm_mask(LOGN) >> S) & 1u;
        if constexpr (S < RB) {
            constexpr int RBIT = RB - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
#pragma unroll
            for (int e = 0; e < (REGS >> (RBIT + 1)); ++e) tw[e] = c.wi[(1 << S) + e];
            inv_stage<RBIT, RN>(x, tw, c);
        } else if constexpr (S < 2 * RB) {
            constexpr int RBIT = 2 * RB - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
            load_tw<(REGS >> (RBIT + 1))>(tw, c.wi + (1 << S) + ((lane >> LC) << (S - RB)));
            inv_stage<RBIT, RN>(x, tw, c);
        } else {
            constexpr int RBIT = LOGN - 1 - S;
            uint32_t tw[REGS >> (RBIT + 1)];
            load_tw<(REGS >> (RBIT + 1))>(tw, c.wi + (1 << S) + (lane << (S - 6)));
            inv_stage<RBIT, RN>(x, tw, c);
        }
    }
    template <int S0, int S1>   // stages S0 .. S1-1 ascending
    static __device__ __forceinline__ void fwd_range(int32_t (&x)[REGS], const PrimeCtx &c, int lane) {
        if constexpr (S0 < S1) { fwd<S0>(x, c, lane); fwd_range<S0 + 1, S1>(x, c, lane); }
    }
    template <int S1, int S0>   // stages S1-1 .. S0 descending
    static __device__ __forceinline__ void inv_range(int32_t (&x)[REGS], const PrimeCtx &c, int lane) {
        if constexpr (S1 > S0) { inv<S1 - 1>(x, c, lane); inv_range<S1 - 1, S0>(x, c, lane); }
    }

    // forward NTT: x in L0 (natural order), |x| <= 2^11 -> L2, |x| < LOGN*P + 2^11
    static __device__ __forceinline__ void forward(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane) {
        fwd_range<0, RB>(x, c, lane);
#pragma unroll
        for (int r = 0; r < REGS; ++r) scr[t1_l0_addr(lane, r)] = (uint32_t)x[r];
        wave_lds_fence();
        read_row(x, scr, lane);
        wave_lds_fence();
        fwd_range<RB, 2 * RB>(x, c, lane);
#pragma unroll
        for (int r = 0; r < REGS; ++r) scr[t2_l1_addr(lane, r)] = (uint32_t)x[r];
        wave_lds_fence();
        read_row(x, scr, lane);
        wave_lds_fence();
        fwd_range<2 * RB, LOGN>(x, c, lane);
    }

    // inverse NTT (unscaled: the 1/N is folded into the key image):
    // x in L2, |x| < 4P -> L0 (natural order), |x| < P
    static __device__ __forceinline__ void inverse(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane) {
        inv_range<LOGN, 2 * RB>(x, c, lane);
        write_row(x, scr, lane);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < REGS; ++r) x[r] = (int32_t)scr[t2_l1_addr(lane, r)];
        wave_lds_fence();
        inv_range<2 * RB, RB>(x, c, lane);
        write_row(x, scr, lane);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < REGS; ++r) x[r] = (int32_t)scr[t1_l0_addr(lane, r)];
        wave_lds_fence();
        inv_range<RB, 0>(x, c, lane);
    }
};

// CRT of canonical residues r0 (mod P0) and r1 (mod P1): the centred integer
// they represent, reduced mod 2^32
__device__ __forceinline__ uint32_t crt_to_torus(uint32_t r0, uint32_t r1) {
    uint32_t t = mont_mul_u(r1 + NTT_P1 - r0, CRT_P0INV_MONT, NTT_P1, NTT_PINV1);
    t = csub(t, NTT_P1);
    const uint64_t v = (uint64_t)NTT_P0 * t + r0;
    return (uint32_t)v - (v > CRT_HALF ? CRT_M_LO : 0u);
}

}  // namespace tfhe_hip
