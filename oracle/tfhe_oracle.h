/*
 * tfhe_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, exact-integer CPU restatement of the TFHE gate-bootstrapping
 * arithmetic that lab-incert/peba1 reaches through the boots* C API
 * (call sites: /root/reference/src/Math.cpp:27-417).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY STATUS: "parity unpinned" at the ciphertext level.  The arithmetic
 * lives in the third-party library tfhe/tfhe (github.com/tfhe/tfhe, linked as
 * /usr/local/lib/libtfhe-nayuki-portable.so by /root/reference/CMakeLists.txt:9-15,
 * version not pinned, absent from /root/reference and from this image), and
 * the reference holds no golden ciphertext, key or intermediate
 * (SURVEY.md section 8c).  What IS pinned: decrypted-bit semantics of every
 * gate (truth tables) and the circuit-level known answers of SURVEY.md 8c,
 * see tests/golden/.  This file restates the published algorithm (CGGI16 /
 * tfhe v1.1 sources as recalled in SURVEY.md Appendix A) with the negacyclic
 * product computed EXACTLY (mod 2^32), where tfhe approximates it with an
 * fp64 FFT.
 *
 * Ciphertext layout everywhere in this repo: int32 words a[0..n-1], then b.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t Torus32;

typedef struct OrcParams {
    int32_t n;          /* LWE dimension                                   */
    int32_t N;          /* ring degree                                     */
    int32_t k;          /* TLWE mask polynomials (1 for every built-in set)*/
    int32_t l;          /* gadget length  (bk_l)                           */
    int32_t Bgbit;      /* log2 gadget base                                */
    int32_t ks_t;       /* key-switch digits                               */
    int32_t ks_basebit; /* log2 key-switch base                            */
    double  ks_stdev;   /* LWE / key-switch noise (alpha_min)              */
    double  bk_stdev;   /* bootstrapping-key noise                         */
    double  max_stdev;
} OrcParams;

/* tfhe v1.1 new_default_gate_bootstrapping_parameters(lambda):
 * lambda in (80,128] -> n=630,N=1024,k=1,l=3,Bgbit=7,ks 8x2bit,
 * lambda <= 80       -> n=500,l=2,Bgbit=10.       */
int orc_params_default(OrcParams *out, int32_t minimum_lambda);
/* BASELINE.json configs[4]: N=2048, Bg=2^6, l=3; the rest fixed by this
 * repo (n=1024, ks 8x2bit, noise as P128) -- documented in DESIGN.md.      */
int orc_params_p2048(OrcParams *out);

/* ---- deterministic PRNG shared (by specification) with the product ---- */
typedef struct OrcRng { uint64_t s[4]; } OrcRng;
void     orc_rng_seed(OrcRng *r, uint64_t seed);
uint64_t orc_rng_next(OrcRng *r);
Torus32  orc_rng_torus(OrcRng *r);
double   orc_rng_gauss(OrcRng *r, double sigma);
Torus32  orc_dtot32(double d);

/* ---- key material ---- */
typedef struct OrcKeySet {
    OrcParams p;
    int32_t  *lwe_key;   /* [n]           bits                               */
    int32_t  *tlwe_key;  /* [k][N]        bits                               */
    Torus32  *bk;        /* [n][(k+1)l][k+1][N]  TGSW(s_i), torus domain     */
    Torus32  *ksk;       /* [kN][t][base][n+1]   row 0 of each digit is zero */
    uint64_t *bk_ntt;    /* [n][(k+1)l][k+1][N]  Goldilocks NTT image of bk  */
    uint32_t *bk_fast;   /* [n][(k+1)l][k+1][2][N] two-prime Montgomery image  */
    double   *bk_fft;    /* [n][(k+1)l][k+1][re N/2 | im N/2] fp64 FFT image (use_ntt = 3 only) */
    double   *bk_fft4;   /* the same image in the order of fft_standin.c's AVX2 transform, times 2/N; built on the first use of use_ntt = 4 */
} OrcKeySet;

OrcKeySet *orc_keygen(const OrcParams *p, uint64_t seed);
void       orc_keyset_free(OrcKeySet *ks);
size_t     orc_bk_words(const OrcParams *p);   /* number of Torus32 in bk  */
size_t     orc_ksk_words(const OrcParams *p);  /* number of Torus32 in ksk */

/* ---- encrypt / decrypt (bootsSymEncrypt / bootsSymDecrypt) ---- */
void    orc_encrypt_bit(const OrcKeySet *ks, OrcRng *rng, int32_t message, Torus32 *ct);
Torus32 orc_phase(const OrcKeySet *ks, const Torus32 *ct);
int32_t orc_decrypt_bit(const OrcKeySet *ks, const Torus32 *ct);

/* ---- pieces of one bootstrapped gate ---- */
int32_t orc_modswitch(Torus32 x, int32_t Msize);              /* step 2 */
Torus32 orc_modswitch_to_torus(int32_t mu, int32_t Msize);
/* use_ntt arguments below: 0 = schoolbook, 1 = Goldilocks NTT, 2 = two 27-bit primes + CRT
 * (vectorisable; used by the cpu_baseline).  All three give the same words.
 * 3 = fp64 FFT, the way upstream TFHE multiplies (folded N/2-point complex transform):
 * APPROXIMATE, NOT the oracle -- low-order noise bits differ from the exact modes; it exists
 * only as a cost-faithful stand-in for upstream's CPU path in bench.py's cpu_baseline note
 * and is checked at decrypt level.
 * 4 = the same fp64 FFT in AVX2 + FMA (fft_standin.c; what upstream's spqlios-fma flavour
 * costs): APPROXIMATE, NOT the oracle; on a host without AVX2/FMA it runs as 3. */
/* exact negacyclic product res = ip * tp mod (X^N+1) mod 2^32, schoolbook */
void orc_negacyclic_schoolbook(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N);
/* same product through the Goldilocks NTT (must equal the schoolbook)      */
void orc_negacyclic_ntt(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N);
/* gadget decomposition of one torus polynomial into l digit polynomials    */
void orc_decompose(int32_t *digits /*[l][N]*/, const Torus32 *poly, const OrcParams *p);
/* acc <- acc + BK_i (.) ((X^barai - 1) * acc);  acc is (k+1) x N           */
void orc_cmux_rotate(const OrcKeySet *ks, int32_t i, int32_t barai, Torus32 *acc, int use_ntt);
/* full blind rotate: acc = (0, X^{-barb} * mu*(1+X+..)), then n CMUX steps */
void orc_blind_rotate(const OrcKeySet *ks, const int32_t *bara, int32_t barb,
                      Torus32 mu, Torus32 *acc, int use_ntt);
/* sample extract at index 0: u is kN+1 words                               */
void orc_sample_extract(const OrcParams *p, const Torus32 *acc, Torus32 *u);
/* key switch kN -> n                                                        */
void orc_keyswitch(const OrcKeySet *ks, const Torus32 *u, Torus32 *ct);
/* prelude + modswitch + blind rotate + extract (tfhe_bootstrap_woKS)       */
void orc_bootstrap_woks(const OrcKeySet *ks, const Torus32 *lin, Torus32 mu, Torus32 *u, int use_ntt);

/* ---- gates ---- */
enum OrcGate {
    ORC_NAND = 0, ORC_OR, ORC_AND, ORC_NOR, ORC_XOR, ORC_XNOR,
    ORC_ANDNY, ORC_ANDYN, ORC_ORNY, ORC_ORYN,
    ORC_NGATES2,
    ORC_MUX = 16, ORC_NOT, ORC_COPY, ORC_CONST0, ORC_CONST1
};
/* the linear prelude of a 2-input gate: t = (0,c0) + sa*ca + sb*cb         */
void orc_gate_prelude(const OrcParams *p, int gate, const Torus32 *ca, const Torus32 *cb, Torus32 *t);
void orc_gate2(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb, int use_ntt);
void orc_mux(const OrcKeySet *ks, Torus32 *out, const Torus32 *a, const Torus32 *b, const Torus32 *c, int use_ntt);
void orc_not(const OrcParams *p, Torus32 *out, const Torus32 *a);
void orc_constant(const OrcParams *p, Torus32 *out, int32_t value);

/* many independent 2-input gates on `nthreads` host threads (cpu_baseline) */
void orc_gate2_batch(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb,
                     int32_t count, int32_t nthreads);
/* same with an explicit evaluator (use_ntt as above; 3 = the approximate fp64-FFT stand-in) */
void orc_gate2_batch_mode(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb,
                          int32_t count, int32_t nthreads, int use_ntt);

#ifdef __cplusplus
}
#endif
#endif
