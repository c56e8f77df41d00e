/*
 * peba1_circuits.h -- C ABI of libpeba1-circuits: the encrypted integer circuits
 * and the protocol function f of lab-incert/peba1, re-expressed over the boots*
 * gate API so that they issue the reference's gate sequence gate for gate.
 *
 * Each entry cites the reference function it mirrors (/root/reference/src/Math.cpp
 * and include/Math.h).  The reference passes templates as std::vector<LweSample*>;
 * here they are pointer arrays plus a slot count.  The library only calls the
 * public tfhe API, so it runs over libtfhe-hip (GPU) or any other provider of
 * those symbols.  With tfhe_hip_set_deferred(1) a whole circuit is recorded and
 * executed as levelised batches at the next flush/decrypt.
 */
#ifndef PEBA1_CIRCUITS_H
#define PEBA1_CIRCUITS_H

#include "tfhe/tfhe.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Math.cpp:27-50   bootsADD1bit: full adder, 7 bootstraps, carry updated in place */
void peba1_add_1bit(LweSample *result, LweSample *a, LweSample *b, LweSample *carry,
                    const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:54-67   bootsADDNbit: ripple-carry adder, carry-out in carry[0] */
void peba1_add_nbit(LweSample *result, LweSample *a, LweSample *b, LweSample *carry, int bitsize,
                    const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:71-93   bootsTwoSComplement */
void peba1_twos_complement(LweSample *result, LweSample *a, int bitsize, const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:97-119  bootsABS */
void peba1_abs(LweSample *result, LweSample *a, int bitsize, const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:123-180 bootsSUBNbit: result (bitsize+1 samples) = |a - b| */
void peba1_sub_nbit(LweSample *result, LweSample *a, LweSample *b, int bitsize,
                    const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:183-211 bootsShiftLeft / bootsShiftRight / bootsShiftLeftNR */
void peba1_shift_left(LweSample *result, LweSample *a, int bitsize, int n, const TFheGateBootstrappingCloudKeySet *ck);
void peba1_shift_right(LweSample *result, LweSample *a, int bitsize, int n, const TFheGateBootstrappingCloudKeySet *ck);
void peba1_shift_left_inplace(LweSample *a, int bitsize, int n, const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:214-250 bootsMultiply: shift-and-add, result is 23 samples (the
 * reference hard-codes length 23) */
void peba1_multiply(LweSample *result, LweSample *a, LweSample *b, int bitsize,
                    const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:259-262 compare_bit */
void peba1_compare_bit(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *lsb_carry,
                       LweSample *tmp, const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:265-286 minimum: result = min(a,b), bit[0] = (a > b) */
void peba1_minimum(LweSample *result, LweSample *bit, const LweSample *a, const LweSample *b, int nb_bits,
                   const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:333-369 HE_EuclideanDistance: result (24 samples) += sum_i (b_i - a_i)^2 */
void peba1_euclidean_distance(LweSample *result, LweSample *const *a, LweSample *const *b, int nslots, int bitsize,
                              const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:379-387 Function_f: result_b[0] = (distance > bound); result_b has 3*bitsize samples */
void peba1_function_f(LweSample *result_b, LweSample *const *a, LweSample *const *b, int nslots,
                      LweSample *bound_match, int bitsize, const TFheGateBootstrappingCloudKeySet *ck);
/* Math.cpp:390-417 Function_g with the reference's heap overflow fixed:
 * result (bitsize samples) = result_b ? r1 : r0, computed as (1-b)*r0 + b*r1 */
void peba1_function_g(LweSample *result, LweSample *result_b, LweSample *r0, LweSample *r1, int bitsize,
                      const TFheGateBootstrappingCloudKeySet *ck);

/* Optimised variants -- NOT the reference's gate sequence: the same inputs
 * and the same decrypted outputs through a different, much smaller and shallower DAG
 * (|a-b| by borrow chain, squarer partial products, one carry-save column compressor for all
 * slots, parallel-prefix adder and comparator; circuits_fast.cpp).  128 slots x 8 bit: about
 * 27,000 bootstraps at depth about 75 instead of 215,544 at depth 377.  result / result_b
 * have 3*bitsize samples; arithmetic modulo 2^(3*bitsize) as in the reference. */
void peba1_euclidean_distance_fast(LweSample *result, LweSample *const *a, LweSample *const *b, int nslots,
                                   int bitsize, const TFheGateBootstrappingCloudKeySet *ck);
void peba1_function_f_fast(LweSample *result_b, LweSample *const *a, LweSample *const *b, int nslots,
                           LweSample *bound_match, int bitsize, const TFheGateBootstrappingCloudKeySet *ck);

/* Slot-sharded variant of the distance for multi-GPU runs: the
 * partial sum of squares over slots [0, nslots) of this rank, 24 samples, with
 * the accumulator explicitly zeroed first. */
void peba1_partial_distance(LweSample *partial, LweSample *const *a, LweSample *const *b, int nslots, int bitsize,
                            const TFheGateBootstrappingCloudKeySet *ck);
/* rank-0 tail: distance = sum of `nparts` partial sums (each 24 samples, pairwise
 * tree of 23-bit adders), then minimum against bound; result_b has 24 samples */
void peba1_combine_and_compare(LweSample *result_b, LweSample *const *partials, int nparts, LweSample *bound_match,
                               const TFheGateBootstrappingCloudKeySet *ck);
/* the same tail through a carry-save compressor, a prefix adder and a prefix comparator
 * (circuits_fast.cpp): depth ~20 instead of ~290 for 8 partial sums; same decrypted result_b[0] */
void peba1_combine_and_compare_fast(LweSample *result_b, LweSample *const *partials, int nparts, LweSample *bound_match,
                                    const TFheGateBootstrappingCloudKeySet *ck);

/* ---- Hamming distance + threshold (not in the reference; BASELINE.json's wording of the
 * workload).  Built from the reference's own blocks: XOR per bit, a pairwise
 * tree of its ripple adders (Math.cpp:54-67) as population count, its comparator
 * (Math.cpp:265-286).  Results are pinned by plaintext arithmetic and per-gate parity. ---- */
/* number of samples of a population count of nbits bits: floor(log2 nbits) + 1 */
int peba1_hamming_count_bits(int nbits);
/* count (peba1_hamming_count_bits(nbits) samples) = popcount(a XOR b); a, b: nbits samples */
void peba1_hamming_distance(LweSample *count, LweSample *a, LweSample *b, int nbits,
                            const TFheGateBootstrappingCloudKeySet *ck);
/* result_b[0] = (hamming(a,b) > bound), same polarity as Function_f; result_b and bound have
 * peba1_hamming_count_bits(nbits) samples */
void peba1_hamming_match(LweSample *result_b, LweSample *a, LweSample *b, int nbits, LweSample *bound_match,
                         const TFheGateBootstrappingCloudKeySet *ck);

#ifdef __cplusplus
}
#endif
#endif
