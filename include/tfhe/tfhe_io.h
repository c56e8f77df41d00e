/* tfhe/tfhe_io.h -- included by the reference (include/Math.h:5, include/Client.h:5); it never
 * calls an export_/import_ function, but a deployment of the PEBA1
 * protocol needs them: the client keeps the secret keyset, the server loads only the cloud
 * keyset and ciphertexts.  The entry points carry upstream's names and argument meaning.
 * The byte format is this library's own container ("TFHP", version 1, little endian):
 * upstream's layout is not in /root/reference and could not be verified, so files are NOT
 * interchangeable with upstream's.  A malformed, truncated or foreign (upstream) file is not
 * fatal: loaders return NULL, import leaves the sample untouched, and tfhe_hip_last_error()
 * says why.
 *
 * Client.h uses std::vector while including only tfhe headers, so in C++ this
 * header must pull in <vector> and <iostream> as upstream's does. */
#ifndef TFHE_HIP_TFHE_IO_H
#define TFHE_HIP_TFHE_IO_H
#include "tfhe_core.h"
#ifdef __cplusplus
#include <iostream>
#include <vector>
#endif
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* parameter set */
void export_tfheGateBootstrappingParameterSet_toFile(FILE *F, const TFheGateBootstrappingParameterSet *params);
TFheGateBootstrappingParameterSet *new_tfheGateBootstrappingParameterSet_fromFile(FILE *F);

/* cloud keyset (bootstrapping key + key-switching key + parameters); the loaded keyset owns
 * its parameters and is released with delete_gate_bootstrapping_cloud_keyset.  The device
 * image is built at the first gate evaluated with it. */
void export_tfheGateBootstrappingCloudKeySet_toFile(FILE *F, const TFheGateBootstrappingCloudKeySet *keyset);
TFheGateBootstrappingCloudKeySet *new_tfheGateBootstrappingCloudKeySet_fromFile(FILE *F);

/* secret keyset (LWE key, TLWE key, and the cloud keyset) */
void export_tfheGateBootstrappingSecretKeySet_toFile(FILE *F, const TFheGateBootstrappingSecretKeySet *keyset);
TFheGateBootstrappingSecretKeySet *new_tfheGateBootstrappingSecretKeySet_fromFile(FILE *F);

/* one ciphertext (pending deferred gates feeding it are flushed first) */
void export_gate_bootstrapping_ciphertext_toFile(FILE *F, const LweSample *sample,
                                                 const TFheGateBootstrappingParameterSet *params);
void import_gate_bootstrapping_ciphertext_fromFile(FILE *F, LweSample *sample,
                                                   const TFheGateBootstrappingParameterSet *params);

#ifdef __cplusplus
}
#endif
#endif
