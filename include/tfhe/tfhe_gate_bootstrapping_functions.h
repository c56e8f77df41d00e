/*
 * tfhe/tfhe_gate_bootstrapping_functions.h -- the C API the reference binds
 * (/root/reference/src/Math.cpp:5, include/Math.h:4-7).  The 16 unmangled symbols
 * that `nm -u` shows for the reference's objects come first;
 * each cites the reference call site it serves.
 */
#ifndef TFHE_HIP_GATE_BOOTSTRAPPING_FUNCTIONS_H
#define TFHE_HIP_GATE_BOOTSTRAPPING_FUNCTIONS_H

#include "tfhe_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- used by Math.o (10) ---- */
/* Math.cpp:28-30 and 33 more sites.  Defined behaviour where upstream has none:
 * a new sample is the trivial encryption of bit 0, (a = 0, b = -1/8), exactly
 * what bootsCONSTANT(.., 0) writes (upstream leaves a[] uninitialised).  That
 * makes Function_f's never-initialised accumulator (Math.cpp:381-383)
 * act as the number 0 instead of sitting on every gate's decision boundary. */
LweSample *new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet *params);
/* Math.cpp:47-49 and 33 more sites */
void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample *samples);
/* Math.cpp:58,75,77 ... : trivial sample (0, value ? 1/8 : -1/8), no bootstrap */
void bootsCONSTANT(LweSample *result, int32_t value, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:153 : negate, no bootstrap */
void bootsNOT(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:31,39,44,63 ... : copy, no bootstrap */
void bootsCOPY(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:38,40,42,158,165,233 */
void bootsAND(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:170 */
void bootsOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:34,35,41,43,84,112 */
void bootsXOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:260 */
void bootsXNOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
/* Math.cpp:261,277 : result = a ? b : c ; two blind rotates, one key switch */
void bootsMUX(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c,
              const TFheGateBootstrappingCloudKeySet *bk);

/* ---- additionally used by main.o (6) ---- */
/* main.cpp:21 */
TFheGateBootstrappingParameterSet *new_default_gate_bootstrapping_parameters(int32_t minimum_lambda);
/* main.cpp:22 : keys + device-resident evaluation keys */
TFheGateBootstrappingSecretKeySet *new_random_gate_bootstrapping_secret_keyset(const TFheGateBootstrappingParameterSet *params);
/* main.cpp:602 */
void delete_gate_bootstrapping_parameters(TFheGateBootstrappingParameterSet *params);
/* main.cpp:600 */
void delete_gate_bootstrapping_secret_keyset(TFheGateBootstrappingSecretKeySet *keyset);
/* main.cpp:63-69 */
void bootsSymEncrypt(LweSample *result, int32_t message, const TFheGateBootstrappingSecretKeySet *key);
/* main.cpp:78-83 : flushes pending deferred gates that feed `sample` */
int32_t bootsSymDecrypt(const LweSample *sample, const TFheGateBootstrappingSecretKeySet *key);

/* ---- rest of upstream's gate API (not called by the reference) ---- */
void bootsNAND(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsNOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDNY(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDYN(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsORNY(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsORYN(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
LweSample *new_gate_bootstrapping_ciphertext(const TFheGateBootstrappingParameterSet *params);
void delete_gate_bootstrapping_ciphertext(LweSample *sample);
void delete_gate_bootstrapping_cloud_keyset(TFheGateBootstrappingCloudKeySet *keyset); /* main.cpp:601 (comment) */

#ifdef __cplusplus
}
#endif
#endif
