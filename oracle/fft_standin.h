/*
 * fft_standin.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE, AND NOT THE PARITY ORACLE.
 * The AVX2 + FMA fp64-FFT evaluator behind use_ntt = 4 (see fft_standin.c): a cost-faithful
 * stand-in for upstream TFHE's CPU path in bench.py's cpu_baseline, approximate like upstream.
 */
#ifndef ORC_FFT_STANDIN_H
#define ORC_FFT_STANDIN_H
#include "tfhe_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
/* 1 when the host CPU has AVX2 and FMA (use_ntt = 4 falls back to 3 otherwise) */
int  orc_fft4_available(void);
/* the n CMUX steps of a blind rotation on an accumulator the caller initialised */
void orc_fft4_blind_rotate_steps(const OrcKeySet *ks, const int32_t *bara, Torus32 *acc);
/* res = ip * tp mod (X^N+1) mod 2^32 through this evaluator (tests bound its error) */
void orc_fft4_negacyclic(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N);
#ifdef __cplusplus
}
#endif
#endif
