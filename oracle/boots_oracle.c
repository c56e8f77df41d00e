/*
 * boots_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The subset of the tfhe gate API that the circuits use (the headers under include/tfhe), served by
 * the CPU oracle (tfhe_oracle.c), so that libpeba1-circuits -- and, where present, the
 * reference's own Math.cpp -- can run whole circuits on the CPU with real ciphertexts.
 * Used only to generate golden digests (tests/golden/make_function_f_digest.py) and by
 * CPU tests; it is far too slow to be anything else (~0.2 s per gate per core).
 * Semantics mirror libtfhe-hip's shim: a fresh sample is the trivial encryption of 0.
 */
#include <stdlib.h>
#include <string.h>

#include "tfhe_oracle.h"
#include "tfhe/tfhe.h"

static OrcKeySet *g_ks = NULL;
static OrcRng g_rng;
static TFheGateBootstrappingParameterSet g_params;
static LweParams g_lwe;
static TFheGateBootstrappingCloudKeySet g_cloud;   /* circuits read cloud_key->params */
static long long g_gates = 0;

void orc_boots_bind(OrcKeySet *ks, uint64_t encrypt_seed) {
    g_ks = ks;
    orc_rng_seed(&g_rng, encrypt_seed);
    g_lwe.n = ks->p.n; g_lwe.alpha_min = ks->p.ks_stdev; g_lwe.alpha_max = ks->p.max_stdev;
    g_params.ks_t = ks->p.ks_t; g_params.ks_basebit = ks->p.ks_basebit;
    g_params.in_out_params = &g_lwe; g_params.tgsw_params = NULL;
    g_cloud.params = &g_params; g_cloud.bk = NULL; g_cloud.bkFFT = NULL;
    g_gates = 0;
}
long long orc_boots_gate_count(void) { return g_gates; }
const TFheGateBootstrappingParameterSet *orc_boots_params(void) { return &g_params; }
const TFheGateBootstrappingCloudKeySet *orc_boots_cloud(void) { return &g_cloud; }

/* flat words of a sample (n mask words then body) and back */
static void to_words(const LweSample *s, Torus32 *w) { memcpy(w, s->a, sizeof(Torus32) * g_ks->p.n); w[g_ks->p.n] = s->b; }
static void from_words(LweSample *s, const Torus32 *w) { memcpy(s->a, w, sizeof(Torus32) * g_ks->p.n); s->b = w[g_ks->p.n]; }
void orc_boots_export(const LweSample *s, int32_t count, Torus32 *out) {
    for (int32_t i = 0; i < count; ++i) to_words(&s[i], out + (size_t)i * (g_ks->p.n + 1));
}

LweSample *new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet *params) {
    const int32_t n = params->in_out_params->n;
    char *mem = (char *)calloc(1, (size_t)nbelems * (sizeof(LweSample) + sizeof(Torus32) * n) + 16);
    LweSample *s = (LweSample *)mem;
    Torus32 *words = (Torus32 *)(mem + (size_t)nbelems * sizeof(LweSample));
    for (int32_t i = 0; i < nbelems; ++i) {
        s[i].a = words + (size_t)i * n;
        s[i].b = -(1 << 29);                 /* trivial encryption of 0 */
        s[i].slot = -1;
    }
    return s;
}
void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample *samples) { (void)nbelems; free(samples); }

void bootsSymEncrypt(LweSample *r, int32_t m, const TFheGateBootstrappingSecretKeySet *key) {
    (void)key;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * (g_ks->p.n + 1));
    orc_encrypt_bit(g_ks, &g_rng, m, w);
    from_words(r, w);
    free(w);
}
int32_t bootsSymDecrypt(const LweSample *s, const TFheGateBootstrappingSecretKeySet *key) {
    (void)key;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * (g_ks->p.n + 1));
    to_words(s, w);
    const int32_t bit = orc_decrypt_bit(g_ks, w);
    free(w);
    return bit;
}

static void gate2(int g, LweSample *r, const LweSample *a, const LweSample *b) {
    const size_t nw = (size_t)g_ks->p.n + 1;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * nw * 3);
    to_words(a, w); to_words(b, w + nw);
    orc_gate2(g_ks, g, w + 2 * nw, w, w + nw, 2);
    from_words(r, w + 2 * nw);
    free(w);
    ++g_gates;
}
void bootsAND(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_AND, r, a, b); }
void bootsOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_OR, r, a, b); }
void bootsXOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_XOR, r, a, b); }
void bootsXNOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_XNOR, r, a, b); }
void bootsNAND(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_NAND, r, a, b); }
void bootsNOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_NOR, r, a, b); }
void bootsANDNY(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_ANDNY, r, a, b); }
void bootsANDYN(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_ANDYN, r, a, b); }
void bootsORNY(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_ORNY, r, a, b); }
void bootsORYN(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *k) { (void)k; gate2(ORC_ORYN, r, a, b); }
void bootsMUX(LweSample *r, const LweSample *a, const LweSample *b, const LweSample *c, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    const size_t nw = (size_t)g_ks->p.n + 1;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * nw * 4);
    to_words(a, w); to_words(b, w + nw); to_words(c, w + 2 * nw);
    orc_mux(g_ks, w + 3 * nw, w, w + nw, w + 2 * nw, 2);
    from_words(r, w + 3 * nw);
    free(w);
    g_gates += 2;
}
void bootsNOT(LweSample *r, const LweSample *a, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    for (int32_t i = 0; i < g_ks->p.n; ++i) r->a[i] = (Torus32)(0u - (uint32_t)a->a[i]);
    r->b = (Torus32)(0u - (uint32_t)a->b);
}
void bootsCOPY(LweSample *r, const LweSample *a, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    if (r == a) return;
    memmove(r->a, a->a, sizeof(Torus32) * g_ks->p.n);
    r->b = a->b;
}
void bootsCONSTANT(LweSample *r, int32_t v, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    memset(r->a, 0, sizeof(Torus32) * g_ks->p.n);
    r->b = v ? (1 << 29) : -(1 << 29);
}
