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
// A butterfly always pairs registers of one lane; the twiddle of stage s is
// W[2^s + (top s bits of j)]: lane-uniform in L0 (scalar loads), per-lane but
// contiguous in L1/L2 (vector loads).
//
// Radix-4 steps with merged Montgomery reductions (tools/ntt_model_r4.py is the exact-integer
// model).  Two consecutive stages s, s+1 of one pass act on four registers x0, x1 (partner in
// stage s+1), x2 (partner in stage s), x3.  In a prime field a radix-4 butterfly saves no
// twiddle product over two radix-2 stages (the fourth root of unity is an ordinary constant),
// but it lets two products share ONE reduction -- the 64-bit sum x1*w2 + x3*w1w2 is reduced
// once -- and drops a quarter of the additions:
//   forward   A = redc(x2 w1)   S = redc(x1 w2 + x3 w1w2)   S' = redc(x1 w3 + x3 (P - w1w3))
//             (y0, y1, y2, y3) = ((x0+A)+S, (x0+A)-S, (x0-A)+S', (x0-A)-S')
//   inverse   s0 = x0+x1, s1 = x2+x3, d0 = x0-x1, d1 = x2-x3
//             y0 = s0+s1   y2 = redc((s0-s1) w1)   y1 = redc(d0 w2 + d1 w3)
//             y3 = redc(d0 w1w2 + d1 (P - w1w3))
// 11 multiplier-class + 6 add instructions where two radix-2 stages take 12 + 8 (forward) or
// 12 + 8 + renormalisations (inverse).  The twiddle products come from a second table
// ("quads": {w2, w3, w1w2, P - w1w3} per (s, t), Montgomery form, one 16-byte load).
//
// Signed lazy arithmetic (values are int32 representatives, not canonical):
//   Montgomery reduction r = T/R mod P with |r| <= |T|/2^32 + P/2 for any |T| < 2^63
//   (v_mad_i64_i32 ..., v_mul_lo_u32, v_mad_i64_i32);
//   forward: magnitudes grow by at most P + 3|x|P/2^32 per radix-4 step: digits (|d| <= 2^11) end
//   below 6.1P (N=1024) / 6.7P (N=2048), inputs below P below 8.2P; no reductions;
//   inverse: the plain sum y0 quadruples, so it is renormalised (times R mod P, 3 more
//   instructions) at the steps where the next step could not take it -- every sum or difference
//   of four inputs must stay below 2^31, i.e. inputs below 4P -- and at the last; the
//   schedule (steps 1, 3, 5 of 5 for N=1024) is computed at compile time by make_inv_plan for
//   inputs below 4P and checked there.
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
    const uint4 *__restrict__ qf;      // forward quads [N/2]: entry 2^s + t = {w2, w3, w1 w2, P - w1 w3} of the
    const uint4 *__restrict__ qi;      // radix-4 step on stages (s, s+1), block t; inverse likewise
    const uint32_t *dtab;              // LDS, or null: first-step products of gadget digits, [5][DIGIT_TAB] (forward_digits)
    const void *fw1, *fw2;             // LDS, or null: this lane's twiddles of the forward transforms' second and third
                                       // pass (PassTw::to_image of a WaveNtt::FwdTw1 / FwdTw2), for the LDSTW forms of forward_digits
};

// The first radix-4 step of a forward transform of gadget digits multiplies 7-bit numbers by five
// fixed twiddles (stages 0 and 1 have one block): with at most DIGIT_TAB_BITS bits per digit the eleven
// multiplier-class instructions of a group become five LDS reads of d * w mod P, indexed by the
// digit's two's-complement bit field.  Rows: w1, w2, w1 w2, w3, P - w1 w3.
constexpr int DIGIT_TAB_BITS = 7, DIGIT_TAB = 1 << DIGIT_TAB_BITS;
// Round 4: S and S' both take the products of x1 (rows w2, w3) and of x3 (rows w1 w2, P - w1 w3),
// so the two rows a digit indexes together are stored as ONE 8-byte entry: per group of four coefficients one
// ds_read_b32 (row w1 by x2's digit) and two ds_read_b64 instead of five dword accesses -- the LDS array serves a
// ds_read_b64 in the cycles of one ds_read_b32, and a data-dependent address pays its bank conflicts once, not twice
// (SQ counters: profiles/archive/r04_lds_conflict_attribution*.txt).  Layout of a table, in words:
//   [0, DIGIT_TAB)                 w1 d
//   [DIGIT_TAB, 3 DIGIT_TAB)       {w2 d, w3 d} per digit field
//   [3 DIGIT_TAB, 5 DIGIT_TAB)     {w1 w2 d, (P - w1 w3) d} per digit field
// the lowest digit field must start at this bit or higher for the table index (8-byte entries) to be a plain shift + mask
constexpr int DIGIT_TAB_MIN_SHIFT = 3;
// How many of the first step's register groups (REGS / 4 of them: four at N = 1024) take their products from the table;
// the rest multiply.  The table trades 11 multiplier-class instructions per group for LDS reads with data-dependent
// addresses, i.e. bank conflicts: all-table against no-table measured 38.36 against 38.49 ms per 4,096 rotations with 2.64 G
// against 1.63 G conflict cycles (profiles/archive/r04_lds_conflict_attribution.txt) -- a VALU / LDS balance whose middle points
// round 6 swept once (tools/diag/r6_table_fraction.sh -> profiles/r06_ab_table_fraction.txt).
// A measurement switch: the default is every group by table.
#ifndef BR_TAB_GROUPS
#define BR_TAB_GROUPS 99
#endif

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
// LDS is processed in issue order for one wave, so a wave-private transpose
// needs no barrier; this only stops the compiler from reordering across it.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() drains the wave's vector-memory counter too
// (s_waitcnt vmcnt(0) in front of s_barrier), so a global load cannot be in flight across it; this one waits for the
// wave's LDS operations alone and leaves its outstanding global loads (the next phase's twiddles and key rows,
// requested just before) running while the wave waits for its siblings.  Nothing in the blind-rotate loop stores to
// global memory, so there is nothing else for the barrier to order.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The same barrier inside a branch only SOME waves of a workgroup take (`if (q == 0) ... else ...`), with a comment that
// rides into the assembly listing: tools/isa_mix.py counts the instructions of one blind-rotate step per wave from that
// listing and needs to know which branches exclude each other.  A comment: the code object is byte for byte the same.
#define LDS_BARRIER_ROLE(tag) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier ; isa_mix role " tag ::: "memory")

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
// two products, one reduction
__device__ __forceinline__ int32_t mont_mul2(int32_t a, uint32_t wa, int32_t b, uint32_t wb, const PrimeCtx &c) {
    return mont_redc((int64_t)a * (int64_t)(int32_t)wa + (int64_t)b * (int64_t)(int32_t)wb, c.P, c.pinv);
}
// radix-4 steps (header comment); q = {w2, w3, w1 w2, P - w1 w3}
__device__ __forceinline__ void ct_bfly4(int32_t &x0, int32_t &x1, int32_t &x2, int32_t &x3, uint32_t w1, const uint4 &q,
                                         const PrimeCtx &c) {
    const int32_t A = mont_mul(x2, w1, c.P, c.pinv);
    const int32_t S = mont_mul2(x1, q.x, x3, q.z, c);
    const int32_t T = mont_mul2(x1, q.y, x3, q.w, c);
    const int32_t u = x0 + A, v = x0 - A;
    x0 = u + S; x1 = u - S; x2 = v + T; x3 = v - T;
}
template <bool RENORM>
__device__ __forceinline__ void gs_bfly4(int32_t &x0, int32_t &x1, int32_t &x2, int32_t &x3, uint32_t w1, const uint4 &q,
                                         const PrimeCtx &c) {
    const int32_t s0 = x0 + x1, s1 = x2 + x3, d0 = x0 - x1, d1 = x2 - x3;
    const int32_t sum = s0 + s1;
    x0 = RENORM ? mont_mul(sum, c.rmod, c.P, c.pinv) : sum;
    x2 = mont_mul(s0 - s1, w1, c.P, c.pinv);
    x1 = mont_mul2(d0, q.x, d1, q.y, c);
    x3 = mont_mul2(d0, q.z, d1, q.w, c);
}

// The steps of the inverse transform (descending: radix-4 pairs from the top of each pass, a
// radix-2 stage where a pass has an odd stage left) and which of them renormalise their plain
// sums, for inputs below 4P; magnitudes in units of the larger prime (tools/ntt_model_r4.py
// inv_schedule is the same rule, checked there against the emulated 32-bit arithmetic).
struct InvPlan {
    int nsteps = 0;
    int fan[12] = {};          // 4 = radix-4 step, 2 = radix-2 stage
    bool renorm[12] = {};
    double out_bound = 0;      // |outputs| / P
};
constexpr InvPlan make_inv_plan(int logn) {
    InvPlan pl;
    const int rb = logn - 6;
    const int lo[3] = {2 * rb, rb, 0}, hi[3] = {logn, 2 * rb, rb};
    for (int p = 0; p < 3; ++p) {
        int cnt = hi[p] - lo[p];
        while (cnt >= 2) { pl.fan[pl.nsteps++] = 4; cnt -= 2; }
        if (cnt) pl.fan[pl.nsteps++] = 2;
    }
    const double P = 134176769.0, q = P / 4294967296.0, limit = 2147483647.0 / P;
    double b = 4.0;
    for (int k = 0; k < pl.nsteps; ++k) {
        const double small = pl.fan[k] * b * q + 0.5, big = pl.fan[k] * b;
        if (big > limit) { pl.out_bound = 1e9; return pl; }            // inputs too large: caught below
        const bool last = k == pl.nsteps - 1;
        const double next_in = big > small ? big : small;
        const bool r = last || pl.fan[k + 1] * next_in > limit;
        pl.renorm[k] = r;
        b = r ? small : next_in;
    }
    pl.out_bound = b;
    return pl;
}
static_assert(make_inv_plan(10).nsteps == 5 && make_inv_plan(10).renorm[0] && !make_inv_plan(10).renorm[1] &&
              make_inv_plan(10).renorm[2] && !make_inv_plan(10).renorm[3] && make_inv_plan(10).renorm[4],
              "N=1024 renormalisation schedule");
static_assert(make_inv_plan(11).nsteps == 7 && make_inv_plan(10).out_bound < 1.0 && make_inv_plan(11).out_bound < 1.0 &&
              make_inv_plan(9).nsteps == 6 && make_inv_plan(9).out_bound < 1.0,
              "inverse NTT outputs must end below P");

template <int LOGN>
struct WaveNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int RB = LOGN - 6;              // register bits
    static constexpr int REGS = 1 << RB;             // coefficients per lane
    static constexpr int LC = LOGN - 2 * RB;         // stages of the last pass
    static constexpr int ROW_WORDS = N + 4 * (64 >> LC);       // layout R: 64 rows of REGS words, a 16-byte pad every 2^LC rows
    // Layout H, the inverse transform's transposes (round 4).  The inverse writes rows with ds_write_b128 and reads
    // them back word by word, scattered -- it never reads a row with ds_read_b128, whose odd lane groups are what
    // forces layout R's shape.  In layout R the row stores (8 contiguous lanes, banks mod 32, rows 16 words apart) and
    // the L0 gather (four rows 16 words apart) are both two-way bank conflicts: at N = 1024, 96 of the LDS array's
    // ~1,320 cycles per wave and blind-rotate step.  Layout H cuts a row into pieces of 8 words; piece c of all 64 rows
    // is an array of its own with a 16-byte pad every 4 rows:
    //     word (row, col)  ->  (col / 8) * HPIECE + 8 * row + 4 * (row / 4) + col % 8.
    // Rows 8 words apart put the eight lanes of a b128 store group on eight different 16-byte slots and the four rows
    // of a gather on four different 8-bank windows; the pads keep the second gather (rows 4 apart) conflict-free too
    // (tools/lds_bank_model.py checks all of it against the bank rules of MI355X_MICROARCH.md).
    static constexpr int HROW = REGS < 8 ? REGS : 8;
    static constexpr int HPIECES = REGS / HROW;
    // piece arrays 16 words off a multiple of 32 where a gather spans two pieces (N = 2048: 16 columns per half wave)
    static constexpr int HPIECE = 64 * HROW + 4 * 16 + (HPIECES > 2 ? 16 : 0);
    static constexpr int INV_WORDS = HPIECES * HPIECE;
    static constexpr int SCRATCH_WORDS = INV_WORDS > ROW_WORDS ? INV_WORDS : ROW_WORDS;   // a wave's transpose scratch
    static_assert(LOGN >= 9 && LOGN <= 11, "wave NTT is laid out for N = 512 (half of a split 1024-point transform), 1024 or 2048");

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
    // layout H (inverse transform only)
    static __device__ __forceinline__ int h_addr(int row, int col) {
        return (col / HROW) * HPIECE + HROW * row + 4 * (row >> 2) + (col % HROW);
    }
    static __device__ __forceinline__ void write_row_h(const int32_t (&x)[REGS], uint32_t *scr, int lane) {
        uint32_t *base = scr + HROW * lane + 4 * (lane >> 2);
#pragma unroll
        for (int g = 0; g < REGS / 4; ++g)
            *reinterpret_cast<uint4 *>(base + ((4 * g) / HROW) * HPIECE + (4 * g) % HROW) =
                make_uint4((uint32_t)x[4 * g], (uint32_t)x[4 * g + 1], (uint32_t)x[4 * g + 2], (uint32_t)x[4 * g + 3]);
    }
    // (lane, reg) of the layout being entered -> where that word sits in layout H
    static __device__ __forceinline__ int h_t2_addr(int lane, int reg) {      // L2 rows -> L1
        const int j = ((lane >> LC) << (LOGN - RB)) | (reg << LC) | (lane & ((1 << LC) - 1));
        return h_addr(j >> RB, j & (REGS - 1));
    }
    static __device__ __forceinline__ int h_t1_addr(int lane, int reg) {      // L1 rows -> L0
        return h_addr((reg << LC) | (lane & ((1 << LC) - 1)), lane >> LC);
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

    // ---- steps ---------------------------------------------------------------------
    // Register bit paired by stage S, and the twiddle-table index of this lane's first block
    // of stage S (the lane needs REGS >> (rbit + 1) consecutive entries from there, selected by
    // the register bits above rbit).
    static constexpr int rbit_of(int S) { return S < RB ? RB - 1 - S : S < 2 * RB ? 2 * RB - 1 - S : LOGN - 1 - S; }
    static constexpr int pass_end(int S) { return S < RB ? RB : S < 2 * RB ? 2 * RB : LOGN; }
    static constexpr int pass_begin(int S) { return S < RB ? 0 : S < 2 * RB ? RB : 2 * RB; }
    template <int S>
    static __device__ __forceinline__ int tw_base(int lane) {
        if constexpr (S < RB) return 1 << S;                                           // L0: lane-uniform
        else if constexpr (S < 2 * RB) return (1 << S) + ((lane >> LC) << (S - RB));   // L1
        else return (1 << S) + (lane << (S - 6));                                       // L2
    }
    template <int CNT>
    static __device__ __forceinline__ void load_quads(uint4 (&q)[CNT], const uint4 *__restrict__ tab) {
#pragma unroll
        for (int e = 0; e < CNT; ++e) q[e] = tab[e];
    }

    // The twiddles of one pass, held in registers: loaded as early as the caller can afford --
    // a pass's loads are issued BEFORE the LDS transpose that precedes it (the compiler cannot
    // do that itself: the transposes are fenced against memory reordering), so their latency
    // overlaps the transpose instead of stalling the first butterflies; the first pass's loads are
    // issued by the caller ahead of whatever it does before the transform.
    // Forward passes take radix-4 steps from their first stage (S, S+2, ...) and a radix-2 stage if
    // one is left; inverse passes take radix-4 steps from their top (TOP-2, TOP-4, ...) likewise.
    template <int S, int END, bool FWD>
    struct PassTw {
        static constexpr bool PAIR = FWD ? (S + 1 < END) : (END - 2 >= S);     // inverse: [S, END) descending from END
        static constexpr int STAGE = FWD ? S : (PAIR ? END - 2 : END - 1);    // the (lower) stage this step starts at
        static constexpr int CNT = REGS >> (rbit_of(STAGE) + 1);
        uint32_t tw[CNT];
        uint4 q[PAIR ? CNT : 1];
        PassTw<FWD ? S + (PAIR ? 2 : 1) : S, FWD ? END : END - (PAIR ? 2 : 1), FWD> rest;
        __device__ __forceinline__ void load(const PrimeCtx &c, int lane) {
            const int base = tw_base<STAGE>(lane);
            load_tw<CNT>(tw, (FWD ? c.wf : c.wi) + base);
            if constexpr (PAIR) load_quads<CNT>(q, (FWD ? c.qf : c.qi) + base);
            rest.load(c, lane);
        }
        // packed image for an LDS copy (16-byte units: the twiddles, then the quads, then the later steps)
        static constexpr int TW16 = (CNT + 3) / 4;
        static constexpr int IMAGE16 = TW16 + (PAIR ? CNT : 0) + decltype(rest)::IMAGE16;
        __device__ __forceinline__ void to_image(uint4 *img) const {
#pragma unroll
            for (int g = 0; g < TW16; ++g)
                img[g] = make_uint4(tw[4 * g], 4 * g + 1 < CNT ? tw[4 * g + 1] : 0u, 4 * g + 2 < CNT ? tw[4 * g + 2] : 0u,
                                    4 * g + 3 < CNT ? tw[4 * g + 3] : 0u);
            if constexpr (PAIR) {
#pragma unroll
                for (int e = 0; e < CNT; ++e) img[TW16 + e] = q[e];
            }
            rest.to_image(img + TW16 + (PAIR ? CNT : 0));
        }
        __device__ __forceinline__ void from_image(const uint4 *img) {
#pragma unroll
            for (int g = 0; g < TW16; ++g) {
                const uint4 v = img[g];
                tw[4 * g] = v.x;
                if constexpr (CNT > 1) { if (4 * g + 1 < CNT) tw[4 * g + 1] = v.y; if (4 * g + 2 < CNT) tw[4 * g + 2] = v.z; if (4 * g + 3 < CNT) tw[4 * g + 3] = v.w; }
            }
            if constexpr (PAIR) {
#pragma unroll
                for (int e = 0; e < CNT; ++e) q[e] = img[TW16 + e];
            }
            rest.from_image(img + TW16 + (PAIR ? CNT : 0));
        }
    };
    template <int S, bool FWD>
    struct PassTw<S, S, FWD> {
        static constexpr int IMAGE16 = 0;
        __device__ __forceinline__ void load(const PrimeCtx &, int) {}
        __device__ __forceinline__ void to_image(uint4 *) const {}
        __device__ __forceinline__ void from_image(const uint4 *) {}
    };
    using FwdTw0 = PassTw<0, RB, true>;
    using FwdTw1 = PassTw<RB, 2 * RB, true>;
    using FwdTw2 = PassTw<2 * RB, LOGN, true>;
    using InvTw2 = PassTw<2 * RB, LOGN, false>;
    using InvTw1 = PassTw<RB, 2 * RB, false>;
    using InvTw0 = PassTw<0, RB, false>;

    // one step with its twiddles in registers
    template <int STAGE, bool PAIR, bool FWD, bool RENORM, int CNT>
    static __device__ __forceinline__ void step(int32_t (&x)[REGS], const uint32_t (&tw)[CNT], const uint4 *q,
                                                const PrimeCtx &c) {
        constexpr int RBIT = rbit_of(STAGE), h = 1 << RBIT, l = h >> 1;
        static_assert(CNT == (REGS >> (RBIT + 1)), "twiddle count of the step");
        if constexpr (PAIR) {
            static_assert(pass_end(STAGE) == pass_end(STAGE + 1), "a radix-4 step stays inside one pass");
#pragma unroll
            for (int r = 0; r < REGS; ++r)
                if (!(r & (h | l))) {
                    const int e = r >> (RBIT + 1);
                    if constexpr (FWD) ct_bfly4(x[r], x[r | l], x[r | h], x[r | h | l], tw[e], q[e], c);
                    else gs_bfly4<RENORM>(x[r], x[r | l], x[r | h], x[r | h | l], tw[e], q[e], c);
                }
        } else {
#pragma unroll
            for (int r = 0; r < REGS; ++r)
                if (!(r & h)) {
                    if constexpr (FWD) ct_bfly(x[r], x[r | h], tw[r >> (RBIT + 1)], c);
                    else gs_bfly<RENORM>(x[r], x[r | h], tw[r >> (RBIT + 1)], c);
                }
        }
    }
    // all steps of a pass; K = index of the step in the whole inverse transform (InvPlan)
    template <int S, int END>
    static __device__ __forceinline__ void fwd_pass(int32_t (&x)[REGS], const PrimeCtx &c, const PassTw<S, END, true> &t) {
        using T = PassTw<S, END, true>;
        step<T::STAGE, T::PAIR, true, false, T::CNT>(x, t.tw, t.q, c);
        if constexpr (S + (T::PAIR ? 2 : 1) < END) fwd_pass(x, c, t.rest);
    }
    template <int K, int S, int END>
    static __device__ __forceinline__ void inv_pass(int32_t (&x)[REGS], const PrimeCtx &c, const PassTw<S, END, false> &t) {
        using T = PassTw<S, END, false>;
        constexpr InvPlan plan = make_inv_plan(LOGN);
        static_assert(plan.fan[K] == (T::PAIR ? 4 : 2), "plan and pass structure agree");
        step<T::STAGE, T::PAIR, false, plan.renorm[K], T::CNT>(x, t.tw, t.q, c);
        if constexpr (END - (T::PAIR ? 2 : 1) > S) inv_pass<K + 1>(x, c, t.rest);
    }
    static constexpr int steps_in(int stages) { return (stages + 1) / 2; }

    // forward NTT: x in L0 (natural order) -> L2; |x| <= 2^11 ends below 6.7P, |x| < P below 8.2P.
    // t0: the first pass's twiddles, loaded by the caller (FwdTw0 t0; t0.load(c, lane);)
    template <bool LDSTW = false>
    static __device__ __forceinline__ void forward(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane,
                                                   const FwdTw0 &t0) {
        fwd_pass(x, c, t0);
        forward_tail<LDSTW>(x, c, scr, lane);
    }
    static __device__ __forceinline__ void forward(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane) {
        FwdTw0 t0;
        t0.load(c, lane);
        forward<false>(x, c, scr, lane, t0);
    }

    // Forward NTT of the gadget digit at bit position `shift`, `width` bits wide, of every D[r] (digit
    // fields in two's complement, i.e. after the (D + offset) ^ offset of kernels.hip).  TABLE: the
    // first radix-4 step reads its products from c.dtab (width <= DIGIT_TAB_BITS) -- per group 5 LDS
    // reads, 3 shift+mask index computations and 8 additions instead of 4 bit-field extractions,
    // 11 multiplier-class instructions and 6 additions; magnitudes after the step are below
    // 1.5P + 2^6 instead of P, after the transform below 6.8P (N=1024) / 7.4P (N=2048).
    // t0: the first pass's twiddles (lane-uniform and the same for every transform: a caller may load them once)
    // LDSTW: the per-lane twiddles of the second and third pass come from the workgroup's LDS copy (c.fw1, c.fw2)
    // instead of global memory.  They are the same for every row and step; from LDS they cost 13 ds_read_b128 per
    // transform where the global form issues 16 loads through the vector-memory pipe -- and, more to the point, they
    // leave the vector-memory counter to the key rows alone: a wait for twiddles no longer waits for the (older,
    // slower) key loads, whose latency then hides under the whole transform (round 3).
    template <bool TABLE, bool LDSTW = false>
    static __device__ __forceinline__ void forward_digits(int32_t (&x)[REGS], const uint32_t (&D)[REGS], int shift, int width,
                                                          const PrimeCtx &c, uint32_t *scr, int lane, const FwdTw0 &t0) {
        if constexpr (!TABLE) {
#pragma unroll
            for (int r = 0; r < REGS; ++r) x[r] = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width);
            fwd_pass(x, c, t0);
            forward_tail<LDSTW>(x, c, scr, lane);
        } else {
            static_assert(FwdTw0::PAIR && FwdTw0::CNT == 1, "the first step is a radix-4 step with one block");
            constexpr int RBIT = rbit_of(0), h = 1 << RBIT, l = h >> 1;
            const uint32_t mask4 = ((1u << width) - 1u) << 2;            // byte offset of a table word
            const int sh = shift - 2;
            const char *tab = reinterpret_cast<const char *>(c.dtab);
            const uint32_t mask8 = mask4 << 1;                           // byte offset of a pair
            const int sh8 = shift - 3;
            const char *tab1 = tab + DIGIT_TAB * 4, *tab3 = tab + 3 * DIGIT_TAB * 4;
#pragma unroll
            for (int r = 0; r < REGS; ++r)
                if (!(r & (h | l))) {
                    // groups in register order: the first BR_TAB_GROUPS by table, the others by the radix-4 butterfly on
                    // their extracted digits (|outputs| < P + 2^6: inside the table groups' 1.5P + 2^6)
                    if ((r & (l - 1)) + ((r >> (RBIT + 1)) * l) >= BR_TAB_GROUPS) {
                        x[r] = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width);
                        x[r | l] = __builtin_amdgcn_sbfe((int32_t)D[r | l], shift, width);
                        x[r | h] = __builtin_amdgcn_sbfe((int32_t)D[r | h], shift, width);
                        x[r | h | l] = __builtin_amdgcn_sbfe((int32_t)D[r | h | l], shift, width);
                        ct_bfly4(x[r], x[r | l], x[r | h], x[r | h | l], t0.tw[0], t0.q[0], c);
                        continue;
                    }
                    const int32_t x0 = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width);
                    const uint32_t i1 = (D[r | l] >> sh8) & mask8, i2 = (D[r | h] >> sh) & mask4, i3 = (D[r | h | l] >> sh8) & mask8;
                    const int32_t A = (int32_t)*reinterpret_cast<const uint32_t *>(tab + i2);
                    const uint2 p1 = *reinterpret_cast<const uint2 *>(tab1 + i1), p3 = *reinterpret_cast<const uint2 *>(tab3 + i3);
                    const int32_t S = (int32_t)p1.x + (int32_t)p3.x;
                    const int32_t T = (int32_t)p1.y + (int32_t)p3.y;
                    const int32_t u = x0 + A, v = x0 - A;
                    x[r] = u + S; x[r | l] = u - S; x[r | h] = v + T; x[r | h | l] = v - T;
                }
            forward_rest<LDSTW>(x, c, scr, lane, t0);
        }
    }
    // everything after the first radix-4 step of a forward transform (for callers that produce that
    // step's outputs themselves, from tables)
    template <bool LDSTW = false>
    static __device__ __forceinline__ void forward_rest(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane,
                                                        const FwdTw0 &t0) {
        static_assert(FwdTw0::PAIR, "the first step is a radix-4 step");
        if constexpr (RB > 2) fwd_pass(x, c, t0.rest);               // the other steps of the first pass
        forward_tail<LDSTW>(x, c, scr, lane);
    }
    // second and third pass of a forward transform, with the two transposes in front of them; a pass's twiddles are
    // requested BEFORE the transpose in front of it (their latency overlaps the transpose)
    template <bool LDSTW>
    static __device__ __forceinline__ void forward_tail(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane) {
        FwdTw1 t1;
        if constexpr (LDSTW) t1.from_image(static_cast<const uint4 *>(c.fw1)); else t1.load(c, lane);
#pragma unroll
        for (int r = 0; r < REGS; ++r) scr[t1_l0_addr(lane, r)] = (uint32_t)x[r];
        wave_lds_fence();
        read_row(x, scr, lane);
        wave_lds_fence();
        fwd_pass(x, c, t1);
        FwdTw2 t2;
        if constexpr (LDSTW) t2.from_image(static_cast<const uint4 *>(c.fw2)); else t2.load(c, lane);
#pragma unroll
        for (int r = 0; r < REGS; ++r) scr[t2_l1_addr(lane, r)] = (uint32_t)x[r];
        wave_lds_fence();
        read_row(x, scr, lane);
        wave_lds_fence();
        fwd_pass(x, c, t2);
    }
    // fills this prime's digit table for digits of `width` bits (threads tid, tid + nthreads, ... of
    // the workgroup); entry f of a row is for the digit whose two's-complement bit field is f
    static __device__ __forceinline__ void build_digit_table(uint32_t *tab, const PrimeCtx &c, int width, int tid, int nthreads) {
        const uint4 q = c.qf[1];
        const uint32_t w[5] = {c.wf[1], q.x, q.z, q.y, q.w};                // w1, w2, w1 w2, w3, P - w1 w3
        const int fields = 1 << width;
        for (int e = tid; e < 5 * fields; e += nthreads) {
            const int k = e >> width, f = e & (fields - 1);
            const int32_t d = f < fields / 2 ? f : f - fields;
            // k = 0: the word row; k = 1, 3 (w2, w3): halves of the pair x1's digit reads; k = 2, 4: of x3's
            const int pos = k == 0 ? f : (k & 1 ? DIGIT_TAB : 3 * DIGIT_TAB) + 2 * f + (k > 2 ? 1 : 0);
            tab[pos] = (uint32_t)mont_mul(d, w[k], c.P, c.pinv);
        }
    }

    // inverse NTT (unscaled: the 1/N is folded into the key image):
    // x in L2, |x| < 4P -> L0 (natural order), |x| < P.  t2: the first pass's twiddles (InvTw2), as above
    // LAYOUT_H (default): the transposes go through layout H (above; scr must hold INV_WORDS words); false: layout R
    // (the 8-wave form's half transforms: with H its schedule came out 0.13 ms per rotation slower, profiles/archive/r04_ab_*.txt)
    template <bool LAYOUT_H = true>
    static __device__ __forceinline__ void inverse(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane,
                                                   const InvTw2 &t2) {
        inv_pass<0>(x, c, t2);
        InvTw1 t1;
        t1.load(c, lane);
        if constexpr (LAYOUT_H) write_row_h(x, scr, lane); else write_row(x, scr, lane);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < REGS; ++r) x[r] = (int32_t)scr[LAYOUT_H ? h_t2_addr(lane, r) : t2_l1_addr(lane, r)];
        wave_lds_fence();
        inv_pass<steps_in(LC)>(x, c, t1);
        InvTw0 t0;
        t0.load(c, lane);
        if constexpr (LAYOUT_H) write_row_h(x, scr, lane); else write_row(x, scr, lane);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < REGS; ++r) x[r] = (int32_t)scr[LAYOUT_H ? h_t1_addr(lane, r) : t1_l0_addr(lane, r)];
        wave_lds_fence();
        inv_pass<steps_in(LC) + steps_in(RB)>(x, c, t0);
    }
    static __device__ __forceinline__ void inverse(int32_t (&x)[REGS], const PrimeCtx &c, uint32_t *scr, int lane) {
        InvTw2 t2;
        t2.load(c, lane);
        inverse<true>(x, c, scr, lane, t2);
    }
};

// CRT of SIGNED representatives r0 (modulo P0) and r1 (modulo P1), |r0|, |r1| < 2P -- what the inverse transforms
// leave, no canonicalisation.  With t = (r1 - r0) P0^-1 mod P1 taken as the signed Montgomery output (|t| <=
// 4P P1 / 2^32 + P1 / 2 < 0.63 P1), x = r0 + P0 t is congruent to the value modulo both primes and |x| < 0.63 M + 2P.
// The true centred value v satisfies |v| < CRT_EXACT_LIMIT = 0.36 M (checked at key upload), so x - v, a multiple of
// M below M in magnitude, is 0: x IS the centred integer and its low 32 bits are r0 + P0 t in wrapping arithmetic.
// 4 multiplier-class + 2 add instructions, where the round-2 form on canonical residues (compare against M/2, conditional
// subtraction) took 4 + 9 and needed canonical inputs (2 more per residue): 184 VALU instructions less per wave and
// blind-rotate step by static count (round 3).
__device__ __forceinline__ uint32_t crt_signed_to_torus(int32_t r0, int32_t r1) {
    const int32_t t = mont_mul(r1 - r0, CRT_P0INV_MONT, NTT_P1, NTT_PINV1);
    return (uint32_t)r0 + NTT_P0 * (uint32_t)t;
}

}  // namespace tfhe_hip
