// ntt_wave.hpp -- one wave64 computes one 1024-point negacyclic NTT mod a
// 27-bit prime, 16 coefficients per lane in registers, two LDS transposes.
//
// Index j = b9..b0.  Register/lane layouts (validated by tools/ntt_model.py):
//   L0: reg = b9..b6, lane = b5..b0            (natural: j = 64*reg + lane)
//   L1: reg = b5..b2, lane = (b9..b6, b1, b0)
//   L2: reg = b3..b0, lane = b9..b4            (j = 16*lane + reg)
// forward (Cooley-Tukey, natural in -> bit-reversed out, psi-merged twiddles):
//   stages 0-3 in L0, transpose, 4-7 in L1, transpose, 8-9 in L2.
// inverse (Gentleman-Sande) runs the same path backwards and ends in L0.
// A butterfly always pairs two registers of one lane; the twiddle of stage s is
// W[2^s + (top s bits of j)]: lane-uniform in L0 (scalar loads), per-lane but
// contiguous in L1/L2 (dwordx2/x4 loads).
//
// Signed lazy arithmetic (values are int32 representatives, not canonical):
//   Montgomery product r = b*w/R mod P with |r| < P for ANY |b| < 2^31, w in [0,P)
//   (v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32);
//   forward butterfly (a,b) -> (a + r, a - r): 5 instructions, magnitudes grow by
//   P per stage: digits (|d| <= 2^11) end below 10P + 2^11 < 2^31, no reductions;
//   inverse butterfly (a,b) -> (a + b, (a - b)*w): sums double, so the sum is
//   renormalised (times R mod P, 3 more instructions) at stages 8, 4 and 0 only:
//   |in| < 3P -> 6P -> [8] P -> 2P,4P,8P -> [4] P -> 2P,4P,8P -> [0] P, and the
//   largest intermediate, a +- b at the renormalising stages, is < 16P < 2^31
//   (16*P1 = 2,146,828,304).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_field.hpp"

namespace tfhe_hip {

constexpr int NTT_SCRATCH_WORDS = 1088;   // 16 rows of 64 words, each padded by 4

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

// ---- transposes through wave-private LDS scratch -------------------------
// T1 address space: word(lane', reg') = lane'*16 + reg' + 4*(lane'>>2) in L1 terms
__device__ __forceinline__ int t1_l0_addr(int lane, int reg) { return reg * 68 + (lane & 3) * 16 + (lane >> 2); }
// T2 address space: natural order padded, word(j) = j + 4*(j>>6)
__device__ __forceinline__ int t2_l1_addr(int lane, int reg) { return reg * 4 + (lane >> 2) * 68 + (lane & 3); }
__device__ __forceinline__ int row16_base(int lane) { return lane * 16 + 4 * (lane >> 2); }

template <typename T>
__device__ __forceinline__ void read_row16(T (&x)[16], const uint32_t *scr, int lane) {
    const uint4 *p = reinterpret_cast<const uint4 *>(scr + row16_base(lane));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint4 v = p[g];
        x[4 * g] = (T)v.x; x[4 * g + 1] = (T)v.y; x[4 * g + 2] = (T)v.z; x[4 * g + 3] = (T)v.w;
    }
}
template <typename T>
__device__ __forceinline__ void write_row16(const T (&x)[16], uint32_t *scr, int lane) {
    uint4 *p = reinterpret_cast<uint4 *>(scr + row16_base(lane));
#pragma unroll
    for (int g = 0; g < 4; ++g)
        p[g] = make_uint4((uint32_t)x[4 * g], (uint32_t)x[4 * g + 1], (uint32_t)x[4 * g + 2], (uint32_t)x[4 * g + 3]);
}

// forward NTT: x in L0 (natural order), |x| <= 2^11 -> L2, |x| < 10P + 2^11
__device__ __forceinline__ void ntt_fwd_1024(int32_t (&x)[16], const PrimeCtx &c, uint32_t *scr, int lane) {
    // pass A: stages 0..3, pairs differ in reg bit 3-s, twiddles lane-uniform
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int h = 1 << (3 - s);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (!(r & h)) ct_bfly(x[r], x[r | h], c.wf[(1 << s) + (r >> (4 - s))], c);
    }
    // L0 -> L1
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[t1_l0_addr(lane, r)] = (uint32_t)x[r];
    wave_lds_fence();
    read_row16(x, scr, lane);
    wave_lds_fence();
    {   // pass B: stages 4..7, twiddle index 2^s + (a << (s-4)) + (reg >> (8-s)), a = lane>>2
        const int a = lane >> 2;
        const uint32_t w4 = c.wf[16 + a];
        const uint2 w5 = *reinterpret_cast<const uint2 *>(c.wf + 32 + 2 * a);
        const uint4 w6 = *reinterpret_cast<const uint4 *>(c.wf + 64 + 4 * a);
        const uint4 w7a = *reinterpret_cast<const uint4 *>(c.wf + 128 + 8 * a);
        const uint4 w7b = *reinterpret_cast<const uint4 *>(c.wf + 128 + 8 * a + 4);
        const uint32_t t5[2] = {w5.x, w5.y};
        const uint32_t t6[4] = {w6.x, w6.y, w6.z, w6.w};
        const uint32_t t7[8] = {w7a.x, w7a.y, w7a.z, w7a.w, w7b.x, w7b.y, w7b.z, w7b.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) ct_bfly(x[r], x[r | 8], w4, c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 4)) ct_bfly(x[r], x[r | 4], t5[r >> 3], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 2)) ct_bfly(x[r], x[r | 2], t6[r >> 2], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 1)) ct_bfly(x[r], x[r | 1], t7[r >> 1], c);
    }
    // L1 -> L2
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[t2_l1_addr(lane, r)] = (uint32_t)x[r];
    wave_lds_fence();
    read_row16(x, scr, lane);
    wave_lds_fence();
    {   // pass C: stages 8,9, twiddle index 2^s + (lane << (s-6)) + (reg >> (10-s))
        const uint4 w8 = *reinterpret_cast<const uint4 *>(c.wf + 256 + 4 * lane);
        const uint4 w9a = *reinterpret_cast<const uint4 *>(c.wf + 512 + 8 * lane);
        const uint4 w9b = *reinterpret_cast<const uint4 *>(c.wf + 512 + 8 * lane + 4);
        const uint32_t t8[4] = {w8.x, w8.y, w8.z, w8.w};
        const uint32_t t9[8] = {w9a.x, w9a.y, w9a.z, w9a.w, w9b.x, w9b.y, w9b.z, w9b.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 2)) ct_bfly(x[r], x[r | 2], t8[r >> 2], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 1)) ct_bfly(x[r], x[r | 1], t9[r >> 1], c);
    }
}

// inverse NTT (unscaled: the 1/N is folded into the key image):
// x in L2, |x| < 3P -> L0 (natural order), |x| < P
__device__ __forceinline__ void ntt_inv_1024(int32_t (&x)[16], const PrimeCtx &c, uint32_t *scr, int lane) {
    {
        const uint4 w8 = *reinterpret_cast<const uint4 *>(c.wi + 256 + 4 * lane);
        const uint4 w9a = *reinterpret_cast<const uint4 *>(c.wi + 512 + 8 * lane);
        const uint4 w9b = *reinterpret_cast<const uint4 *>(c.wi + 512 + 8 * lane + 4);
        const uint32_t t8[4] = {w8.x, w8.y, w8.z, w8.w};
        const uint32_t t9[8] = {w9a.x, w9a.y, w9a.z, w9a.w, w9b.x, w9b.y, w9b.z, w9b.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 1)) gs_bfly<false>(x[r], x[r | 1], t9[r >> 1], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 2)) gs_bfly<true>(x[r], x[r | 2], t8[r >> 2], c);
    }
    // L2 -> L1
    write_row16(x, scr, lane);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (int32_t)scr[t2_l1_addr(lane, r)];
    wave_lds_fence();
    {
        const int a = lane >> 2;
        const uint32_t w4 = c.wi[16 + a];
        const uint2 w5 = *reinterpret_cast<const uint2 *>(c.wi + 32 + 2 * a);
        const uint4 w6 = *reinterpret_cast<const uint4 *>(c.wi + 64 + 4 * a);
        const uint4 w7a = *reinterpret_cast<const uint4 *>(c.wi + 128 + 8 * a);
        const uint4 w7b = *reinterpret_cast<const uint4 *>(c.wi + 128 + 8 * a + 4);
        const uint32_t t5[2] = {w5.x, w5.y};
        const uint32_t t6[4] = {w6.x, w6.y, w6.z, w6.w};
        const uint32_t t7[8] = {w7a.x, w7a.y, w7a.z, w7a.w, w7b.x, w7b.y, w7b.z, w7b.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 1)) gs_bfly<false>(x[r], x[r | 1], t7[r >> 1], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 2)) gs_bfly<false>(x[r], x[r | 2], t6[r >> 2], c);
#pragma unroll
        for (int r = 0; r < 16; ++r) if (!(r & 4)) gs_bfly<false>(x[r], x[r | 4], t5[r >> 3], c);
#pragma unroll
        for (int r = 0; r < 8; ++r) gs_bfly<true>(x[r], x[r | 8], w4, c);
    }
    // L1 -> L0
    write_row16(x, scr, lane);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (int32_t)scr[t1_l0_addr(lane, r)];
    wave_lds_fence();
#pragma unroll
    for (int s = 3; s >= 1; --s) {
        const int h = 1 << (3 - s);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (!(r & h)) gs_bfly<false>(x[r], x[r | h], c.wi[(1 << s) + (r >> (4 - s))], c);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) gs_bfly<true>(x[r], x[r | 8], c.wi[1], c);
}

// CRT of canonical residues r0 (mod P0) and r1 (mod P1): the centred integer
// they represent, reduced mod 2^32
__device__ __forceinline__ uint32_t crt_to_torus(uint32_t r0, uint32_t r1) {
    uint32_t t = mont_mul_u(r1 + NTT_P1 - r0, CRT_P0INV_MONT, NTT_P1, NTT_PINV1);
    t = csub(t, NTT_P1);
    const uint64_t v = (uint64_t)NTT_P0 * t + r0;
    return (uint32_t)v - (v > CRT_HALF ? CRT_M_LO : 0u);
}

}  // namespace tfhe_hip
