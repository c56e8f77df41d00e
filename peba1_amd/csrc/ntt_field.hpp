// ntt_field.hpp -- the two NTT primes and their Montgomery constants, shared by
// host (table generation) and device (kernels).
//
// Why two 27-bit primes: measured on gfx950 (profiles/archive/r01_valu_rates.txt),
// v_mul_lo/hi_u32, v_mad_u64_u32 and v_fma_f64 all issue at the same rate
// (~4.5 cycles per wave64 instruction at >=2 waves/SIMD), half the rate of
// v_add_u32.  A Montgomery product is then 3 multiplier ops, and with
// P < 2^27.6 every value of a 10- or 11-stage lazy NTT stays below 2^32
// ((1+2*stages)*P), so butterflies need no reduction at all.  The exact
// negacyclic product of the external product is bounded by
// (k+1) l N (Bg/2) 2^31 <= 2^49.6; P0*P1 ~ 2^54
// covers the centred range, and the result is recombined by CRT and reduced
// mod 2^32.
#pragma once
#include <stdint.h>

namespace tfhe_hip {

constexpr uint32_t NTT_P[2] = {134111233u /*0x7fe6001*/, 134176769u /*0x7ff6001*/};
constexpr int NTT_MAX_LOGN = 11;   // 4096 | P-1 for both primes

constexpr uint32_t neg_inv32(uint32_t p) {
    // Newton iteration for p^-1 mod 2^32, then negate
    uint32_t x = p;                 // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2u - p * x;
    return 0u - x;
}
constexpr uint32_t NTT_PINV_NEG[2] = {neg_inv32(NTT_P[0]), neg_inv32(NTT_P[1])};

constexpr uint64_t mulmod_c(uint64_t a, uint64_t b, uint64_t p) { return a * b % p; }
constexpr uint64_t powmod_c(uint64_t a, uint64_t e, uint64_t p) {
    uint64_t r = 1;
    a %= p;
    while (e) { if (e & 1) r = mulmod_c(r, a, p); a = mulmod_c(a, a, p); e >>= 1; }
    return r;
}
// R = 2^32 mod P
constexpr uint32_t NTT_R[2] = {(uint32_t)((1ull << 32) % NTT_P[0]), (uint32_t)((1ull << 32) % NTT_P[1])};
// generators of (Z/P)^* (sympy.primitive_root)
constexpr uint32_t NTT_GEN[2] = {10u, 3u};

// CRT: x = r0 + P0 * ((r1 - r0) * P0^{-1} mod P1); constant in Montgomery form mod P1
constexpr uint32_t CRT_P0INV_MONT = (uint32_t)mulmod_c(powmod_c(NTT_P[0], NTT_P[1] - 2, NTT_P[1]), NTT_R[1], NTT_P[1]);
// scalar aliases (device code uses these, never the arrays)
constexpr uint32_t NTT_P0 = NTT_P[0], NTT_P1 = NTT_P[1];
constexpr uint32_t NTT_PINV0 = NTT_PINV_NEG[0], NTT_PINV1 = NTT_PINV_NEG[1];
constexpr uint64_t CRT_M = (uint64_t)NTT_P[0] * NTT_P[1];
// The kernels recombine SIGNED lazy residues (ntt_wave.hpp crt_signed_to_torus): the integer they form is
// r0 + P0 t with |r0| < 2P and |t| < 0.63 P1, i.e. below 0.63 M + 2P in magnitude.  It equals the true centred
// value -- not merely modulo M -- as long as that value is below M - (0.63 M + 2P); a key is accepted only if the
// bound of its external product, (k+1) l N (Bg/2) 2^31, stays below this limit (0.36 M = 2^52.5; TFHE's sets reach
// 2^49.6, the legacy Bg = 2^10 set 2^52).
constexpr uint64_t CRT_EXACT_LIMIT = CRT_M / 100 * 36;

}  // namespace tfhe_hip
