/*
 * tfhe/tfhe_core.h -- drop-in replacement for the tfhe/tfhe header of the same
 * name, as included by the reference at /root/reference/include/Math.h:7.
 *
 * Types only.  The reference's callers (src/Math.cpp, src/main.cpp) never read a
 * field of LweSample; they index arrays of it (Math.cpp:61,84,158) and use
 *   cloud_key->params      (Math.cpp:28 and 35 more sites)
 *   &key->cloud            (main.cpp:23)
 * so those member names are kept.  The rest of each struct belongs to this
 * library (libtfhe-hip), which is the only code that reads it.
 */
#ifndef TFHE_HIP_TFHE_CORE_H
#define TFHE_HIP_TFHE_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Torus32 = fixed-point element of R/Z (x / 2^32).  Must support
 * abs(a - b) < 10 as at Math.cpp:253. */
typedef int32_t Torus32;

/* LWE ciphertext handle.  Same size and field offsets as upstream
 * (a @0, b @8, current_variance @16); `slot` sits in upstream's padding.
 * When slot >= 0 the authoritative value lives in device memory and the host
 * mirror (a, b) is refreshed by bootsSymDecrypt / tfhe_hip_sync_sample. */
typedef struct LweSample {
    Torus32 *a;               /* n mask words (host mirror)                    */
    Torus32  b;               /* body (host mirror)                            */
    int32_t  slot;            /* device slot holding the value, -1 = host only */
    double   current_variance;/* kept for source compatibility; not maintained */
} LweSample;

typedef struct LweParams {
    int32_t n;
    double  alpha_min;
    double  alpha_max;
} LweParams;

typedef struct TLweParams {
    int32_t N;
    int32_t k;
    double  alpha_min;
    double  alpha_max;
} TLweParams;

typedef struct TGswParams {
    int32_t l;
    int32_t Bgbit;
    int32_t Bg;
    int32_t halfBg;
    uint32_t maskMod;
    const TLweParams *tlwe_params;
    int32_t kpl;
    uint32_t offset;
} TGswParams;

typedef struct TFheGateBootstrappingParameterSet {
    int32_t ks_t;
    int32_t ks_basebit;
    const LweParams  *in_out_params;
    const TGswParams *tgsw_params;
} TFheGateBootstrappingParameterSet;

struct TfheHipCloudKey;   /* engine-side key material (host + device) */
struct TfheHipSecretKey;

typedef struct TFheGateBootstrappingCloudKeySet {
    const TFheGateBootstrappingParameterSet *params;
    struct TfheHipCloudKey *bk;      /* upstream: LweBootstrappingKey*    */
    struct TfheHipCloudKey *bkFFT;   /* upstream: LweBootstrappingKeyFFT* */
} TFheGateBootstrappingCloudKeySet;

typedef struct TFheGateBootstrappingSecretKeySet {
    const TFheGateBootstrappingParameterSet *params;
    struct TfheHipSecretKey *lwe_key;   /* upstream: LweKey*  */
    struct TfheHipSecretKey *tgsw_key;  /* upstream: TGswKey* */
    TFheGateBootstrappingCloudKeySet cloud;  /* embedded by value, as upstream */
} TFheGateBootstrappingSecretKeySet;

/* tfhe: modSwitchFromTorus32 / modSwitchToTorus32 (numeric helpers) */
int32_t modSwitchFromTorus32(Torus32 phase, int32_t Msize);
Torus32 modSwitchToTorus32(int32_t mu, int32_t Msize);

#ifdef __cplusplus
}
#endif
#endif
