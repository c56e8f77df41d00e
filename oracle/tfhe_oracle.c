/*
 * tfhe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see tfhe_oracle.h).
 *
 * Plain-C exact-integer restatement of TFHE gate bootstrapping as used by
 * lab-incert/peba1 through bootsAND/OR/XOR/XNOR/MUX/NOT/COPY/CONSTANT
 * (/root/reference/src/Math.cpp:27-417).  "parity unpinned" at ciphertext
 * level -- the arithmetic is in tfhe/tfhe (un-vendored, un-pinned; linked at
 * /root/reference/CMakeLists.txt:9-15); the algorithm follows SURVEY.md
 * Appendix A.  Each function names the upstream routine it restates.
 *
 * The negacyclic product is exact mod 2^32.  Three interchangeable evaluators:
 *   - schoolbook in wrapping 32-bit arithmetic (the definition),
 *   - a 64-bit Goldilocks (p = 2^64-2^32+1) NTT, exact because
 *     |sum| < (k+1) l N (Bg/2) 2^31 <= 2^52 < p/2  (SURVEY.md Appendix A.5), and
 *   - two 27-bit primes + CRT in plain vectorisable loops (the arithmetic the GPU
 *     uses; here for a CPU baseline that is not handicapped by 128-bit products).
 * tests/test_host_cpu.py checks that all three give the same words.
 */
#include "tfhe_oracle.h"
#include "fft_standin.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* parameters (tfhe: new_default_gate_bootstrapping_parameters)        */
/* ------------------------------------------------------------------ */
int orc_params_default(OrcParams *o, int32_t lambda) {
    if (lambda <= 0 || lambda > 128) return -1;
    o->N = 1024; o->k = 1; o->ks_t = 8; o->ks_basebit = 2; o->max_stdev = 0.012467;
    if (lambda > 80) {            /* v1.1 128-bit set */
        o->n = 630; o->l = 3; o->Bgbit = 7;
        o->ks_stdev = 1.0 / 32768.0;          /* 2^-15 */
        o->bk_stdev = 1.0 / 33554432.0;       /* 2^-25 */
    } else {                      /* 80-bit set */
        o->n = 500; o->l = 2; o->Bgbit = 10;
        o->ks_stdev = 2.44e-5; o->bk_stdev = 7.18e-9;
    }
    return 0;
}

int orc_params_p2048(OrcParams *o) {
    o->N = 2048; o->k = 1; o->n = 1024; o->l = 3; o->Bgbit = 6;
    o->ks_t = 8; o->ks_basebit = 2;
    o->ks_stdev = 1.0 / 32768.0;
    o->bk_stdev = 1.0 / 33554432.0;
    o->max_stdev = 0.012467;
    return 0;
}

size_t orc_bk_words(const OrcParams *p) {
    return (size_t)p->n * (size_t)((p->k + 1) * p->l) * (size_t)(p->k + 1) * (size_t)p->N;
}
size_t orc_ksk_words(const OrcParams *p) {
    return (size_t)p->k * p->N * (size_t)p->ks_t * (size_t)(1 << p->ks_basebit) * (size_t)(p->n + 1);
}

/* ------------------------------------------------------------------ */
/* PRNG: xoshiro256** seeded by splitmix64.  The product implements the */
/* same generator from this specification ("key derivation") */
/* ------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_rng_seed(OrcRng *r, uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; ++i) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t t = z;
        t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ULL;
        t = (t ^ (t >> 27)) * 0x94D049BB133111EBULL;
        r->s[i] = t ^ (t >> 31);
    }
}

uint64_t orc_rng_next(OrcRng *r) {
    uint64_t *s = r->s;
    const uint64_t result = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

Torus32 orc_rng_torus(OrcRng *r) { return (Torus32)(uint32_t)(orc_rng_next(r) >> 32); }

/* tfhe dtot32: double in R/Z -> Torus32 */
Torus32 orc_dtot32(double d) {
    return (Torus32)(int64_t)((d - (double)(int64_t)d) * 4294967296.0);
}

/* Box-Muller, one normal per two draws (tfhe uses std::normal_distribution;
 * the sampler is a keygen detail, bootstrapping itself draws nothing).       */
double orc_rng_gauss(OrcRng *r, double sigma) {
    const double u1 = ((double)(orc_rng_next(r) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(orc_rng_next(r) >> 11) * (1.0 / 9007199254740992.0);
    return sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* ------------------------------------------------------------------ */
/* modulus switch (tfhe: modSwitchFromTorus32 / modSwitchToTorus32)    */
/* ------------------------------------------------------------------ */
int32_t orc_modswitch(Torus32 x, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t half = interv / 2;
    const uint64_t phase64 = ((uint64_t)(uint32_t)x << 32) + half;
    return (int32_t)(phase64 / interv);
}

Torus32 orc_modswitch_to_torus(int32_t mu, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t phase64 = (uint64_t)(int64_t)mu * interv;
    return (Torus32)(phase64 >> 32);
}

/* ------------------------------------------------------------------ */
/* exact negacyclic product, schoolbook (tfhe: torusPolynomialMultNaive */
/* semantics, mod X^N+1, wrapping Torus32)                              */
/* ------------------------------------------------------------------ */
void orc_negacyclic_schoolbook(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N) {
    uint32_t *r = (uint32_t *)res;
    for (int32_t i = 0; i < N; ++i) r[i] = 0;
    for (int32_t i = 0; i < N; ++i) {
        const uint32_t a = (uint32_t)ip[i];
        if (a == 0) continue;
        for (int32_t j = 0; j < N - i; ++j) r[i + j] += a * (uint32_t)tp[j];
        for (int32_t j = N - i; j < N; ++j) r[i + j - N] -= a * (uint32_t)tp[j];
    }
}

/* ------------------------------------------------------------------ */
/* Goldilocks field and negacyclic NTT                                  */
/* ------------------------------------------------------------------ */
#define GL_P UINT64_C(0xFFFFFFFF00000001)
#define GL_EPS UINT64_C(0xFFFFFFFF)

static inline uint64_t gl_reduce128(unsigned __int128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & GL_EPS;
    uint64_t t = lo - hh;               /* 2^96 = -1 */
    if (lo < hh) t -= GL_EPS;           /* wrapped: +2^64 too much, 2^64 = p + eps */
    const uint64_t t2 = (hl << 32) - hl; /* hl * (2^32-1), 2^64 = eps */
    uint64_t r = t + t2;
    if (r < t2) r += GL_EPS;
    if (r >= GL_P) r -= GL_P;
    return r;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((unsigned __int128)a * b); }
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t r = a + b;
    if (r < a || r >= GL_P) r -= GL_P;
    return r;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a - b + GL_P; }
static uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}

typedef struct NttTab { int32_t N; uint64_t *psi_br; uint64_t *ipsi_br; uint64_t ninv; } NttTab;
#define ORC_MAX_TABS 16
static NttTab g_tabs[ORC_MAX_TABS];
static int g_ntabs = 0;
static pthread_mutex_t g_tab_mtx = PTHREAD_MUTEX_INITIALIZER;

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static const NttTab *ntt_tab(int32_t N) {
    pthread_mutex_lock(&g_tab_mtx);
    for (int i = 0; i < g_ntabs; ++i)
        if (g_tabs[i].N == N) { pthread_mutex_unlock(&g_tab_mtx); return &g_tabs[i]; }
    if (g_ntabs >= ORC_MAX_TABS) abort();        /* more distinct ring degrees than this test helper expects */
    NttTab *t = &g_tabs[g_ntabs];
    int logn = 0; while ((1 << logn) < N) ++logn;
    /* 7 generates the multiplicative group; psi = primitive 2N-th root */
    const uint64_t psi = gl_pow(7, (GL_P - 1) / (uint64_t)(2 * N));
    const uint64_t ipsi = gl_pow(psi, GL_P - 2);
    t->N = N;
    t->psi_br = (uint64_t *)malloc(sizeof(uint64_t) * N);
    t->ipsi_br = (uint64_t *)malloc(sizeof(uint64_t) * N);
    uint64_t a = 1, b = 1;
    for (int32_t i = 0; i < N; ++i) {
        const uint32_t j = bitrev((uint32_t)i, logn);
        t->psi_br[j] = a; t->ipsi_br[j] = b;
        a = gl_mul(a, psi); b = gl_mul(b, ipsi);
    }
    t->ninv = gl_pow((uint64_t)N, GL_P - 2);
    ++g_ntabs;
    pthread_mutex_unlock(&g_tab_mtx);
    return t;
}

/* forward: natural order in, bit-reversed out (merged psi twist) */
static void gl_ntt_fwd(uint64_t *a, const NttTab *t) {
    const int32_t N = t->N;
    int32_t len = N / 2, m = 1;
    for (; m < N; m <<= 1, len >>= 1) {
        for (int32_t i = 0; i < m; ++i) {
            const uint64_t w = t->psi_br[m + i];
            uint64_t *x = a + 2 * i * len, *y = x + len;
            for (int32_t j = 0; j < len; ++j) {
                const uint64_t u = x[j], v = gl_mul(y[j], w);
                x[j] = gl_add(u, v); y[j] = gl_sub(u, v);
            }
        }
    }
}
/* inverse: bit-reversed in, natural out, scaled by 1/N */
static void gl_ntt_inv(uint64_t *a, const NttTab *t) {
    const int32_t N = t->N;
    int32_t len = 1, m = N / 2;
    for (; m >= 1; m >>= 1, len <<= 1) {
        for (int32_t i = 0; i < m; ++i) {
            const uint64_t w = t->ipsi_br[m + i];
            uint64_t *x = a + 2 * i * len, *y = x + len;
            for (int32_t j = 0; j < len; ++j) {
                const uint64_t u = x[j], v = y[j];
                x[j] = gl_add(u, v); y[j] = gl_mul(gl_sub(u, v), w);
            }
        }
    }
    for (int32_t j = 0; j < N; ++j) a[j] = gl_mul(a[j], t->ninv);
}
static inline uint64_t gl_from_i32(int32_t v) { return v >= 0 ? (uint64_t)v : GL_P - (uint64_t)(-(int64_t)v); }
/* centred lift then reduction mod 2^32 */
static inline Torus32 gl_to_torus(uint64_t v) {
    return v > GL_P / 2 ? (Torus32)(uint32_t)(0u - (uint32_t)(GL_P - v)) : (Torus32)(uint32_t)v;
}

void orc_negacyclic_ntt(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N) {
    const NttTab *t = ntt_tab(N);
    uint64_t *a = (uint64_t *)malloc(sizeof(uint64_t) * 2 * N), *b = a + N;
    for (int32_t i = 0; i < N; ++i) { a[i] = gl_from_i32(ip[i]); b[i] = gl_from_i32(tp[i]); }
    gl_ntt_fwd(a, t); gl_ntt_fwd(b, t);
    for (int32_t i = 0; i < N; ++i) a[i] = gl_mul(a[i], b[i]);
    gl_ntt_inv(a, t);
    for (int32_t i = 0; i < N; ++i) res[i] = gl_to_torus(a[i]);
    free(a);
}

/* ------------------------------------------------------------------ */
/* Third evaluator of the exact negacyclic product: two 27-bit primes    */
/* (the same pair the GPU uses) with 32-bit Montgomery arithmetic and    */
/* CRT.  Plain loops over uint32_t that gcc vectorises; it exists so     */
/* that the cpu_baseline is not handicapped by 128-bit scalar products.  */
/* Exact for |sum| < P0*P1/2 ~ 2^53 (checked at key generation).         */
/* ------------------------------------------------------------------ */
#define FP0 134111233u
#define FP1 134176769u
static const uint32_t FP[2] = {FP0, FP1};
static const uint32_t FGEN[2] = {10u, 3u};        /* primitive roots */

typedef struct FastTab {
    int32_t N;
    uint32_t pinv[2];          /* -P^-1 mod 2^32 */
    uint32_t *wf[2], *wi[2];   /* psi^{+-brv(i)} * R mod P */
    uint32_t scale[2];         /* N^-1 * R mod P (folded into the key image) */
    uint32_t p0inv_mont;       /* P0^-1 mod P1, Montgomery form */
} FastTab;
static FastTab g_ftabs[ORC_MAX_TABS];
static int g_nftabs = 0;

static uint64_t powmod64(uint64_t a, uint64_t e, uint64_t p) {
    uint64_t r = 1; a %= p;
    while (e) { if (e & 1) r = r * a % p; a = a * a % p; e >>= 1; }
    return r;
}
static inline uint32_t fmont(uint32_t a, uint32_t b, uint32_t P, uint32_t pinv) {
    const uint64_t T = (uint64_t)a * b;
    const uint32_t m = (uint32_t)T * pinv;
    const uint32_t r = (uint32_t)((T + (uint64_t)m * P) >> 32);   /* < 2P for a,b < P... a < 2^32 */
    return r >= P ? r - P : r;
}
static const FastTab *fast_tab(int32_t N) {
    pthread_mutex_lock(&g_tab_mtx);
    for (int i = 0; i < g_nftabs; ++i)
        if (g_ftabs[i].N == N) { pthread_mutex_unlock(&g_tab_mtx); return &g_ftabs[i]; }
    if (g_nftabs >= ORC_MAX_TABS) abort();
    FastTab *t = &g_ftabs[g_nftabs];
    int logn = 0; while ((1 << logn) < N) ++logn;
    t->N = N;
    for (int q = 0; q < 2; ++q) {
        const uint64_t P = FP[q], R = (UINT64_C(1) << 32) % P;
        uint32_t x = (uint32_t)P;                     /* Newton: P^-1 mod 2^32 */
        for (int i = 0; i < 5; ++i) x *= 2u - (uint32_t)P * x;
        t->pinv[q] = 0u - x;
        const uint64_t psi = powmod64(FGEN[q], (P - 1) / (uint64_t)(2 * N), P), ipsi = powmod64(psi, P - 2, P);
        t->wf[q] = (uint32_t *)malloc(sizeof(uint32_t) * N);
        t->wi[q] = (uint32_t *)malloc(sizeof(uint32_t) * N);
        uint64_t a = 1, b = 1;
        for (int32_t i = 0; i < N; ++i) {
            const uint32_t j = bitrev((uint32_t)i, logn);
            t->wf[q][j] = (uint32_t)(a * R % P);
            t->wi[q][j] = (uint32_t)(b * R % P);
            a = a * psi % P; b = b * ipsi % P;
        }
        t->scale[q] = (uint32_t)(powmod64((uint64_t)N, P - 2, P) * R % P);
    }
    t->p0inv_mont = (uint32_t)(powmod64(FP0, FP1 - 2, FP1) * ((UINT64_C(1) << 32) % FP1) % FP1);
    ++g_nftabs;
    pthread_mutex_unlock(&g_tab_mtx);
    return t;
}
/* forward / inverse transforms on canonical residues [0,P).  target_clones: gcc emits an
 * AVX2 and a baseline version and picks at load time, so the .so still runs anywhere. */
#define ORC_MULTIVERSION __attribute__((target_clones("avx2", "default")))
ORC_MULTIVERSION static void fast_fwd(uint32_t *a, const FastTab *t, int q) {
    const uint32_t P = FP[q], pinv = t->pinv[q];
    const int32_t N = t->N;
    for (int32_t m = 1, len = N / 2; m < N; m <<= 1, len >>= 1)
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t w = t->wf[q][m + i];
            uint32_t *x = a + 2 * i * len, *y = x + len;
            for (int32_t j = 0; j < len; ++j) {
                const uint32_t u = x[j], v = fmont(y[j], w, P, pinv);
                uint32_t s = u + v, d = u + P - v;
                x[j] = s >= P ? s - P : s;
                y[j] = d >= P ? d - P : d;
            }
        }
}
ORC_MULTIVERSION static void fast_inv(uint32_t *a, const FastTab *t, int q) {
    const uint32_t P = FP[q], pinv = t->pinv[q];
    const int32_t N = t->N;
    for (int32_t m = N / 2, len = 1; m >= 1; m >>= 1, len <<= 1)
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t w = t->wi[q][m + i];
            uint32_t *x = a + 2 * i * len, *y = x + len;
            for (int32_t j = 0; j < len; ++j) {
                const uint32_t u = x[j], v = y[j];
                uint32_t s = u + v;
                x[j] = s >= P ? s - P : s;
                y[j] = fmont(u + P - v, w, P, pinv);
            }
        }
}
/* sum[j] += x[j] * b[j] over one polynomial */
ORC_MULTIVERSION static void fast_mac(uint64_t *sum, const uint32_t *x, const uint32_t *b, int32_t N) {
    for (int32_t j = 0; j < N; ++j) sum[j] += (uint64_t)x[j] * b[j];
}
static inline uint32_t fast_from_i32(int32_t v, uint32_t P) { int32_t m = v % (int32_t)P; return (uint32_t)(m < 0 ? m + (int32_t)P : m); }
/* CRT of canonical residues, centred, mod 2^32 */
static inline Torus32 fast_crt(uint32_t r0, uint32_t r1, const FastTab *t) {
    const uint32_t tt = fmont(r1 + FP1 - r0, t->p0inv_mont, FP1, t->pinv[1]);
    const uint64_t v = (uint64_t)FP0 * tt + r0, M = (uint64_t)FP0 * FP1;
    return (Torus32)(uint32_t)(v > (M - 1) / 2 ? v - M : v);
}

/* ------------------------------------------------------------------ */
/* key generation (tfhe: new_random_gate_bootstrapping_secret_keyset,   */
/* tGswSymEncryptInt, lweCreateKeySwitchKey) with this repo's PRNG.     */
/* Draw order is part of the shared specification:                      */
/*   lwe_key bits, tlwe_key bits, then BK rows in memory order (mask     */
/*   polys uniform, then N gaussians for the body), then KSK rows in     */
/*   memory order skipping digit value 0 (mask uniform, one gaussian).   */
/* ------------------------------------------------------------------ */
static void negacyclic_mul_bits_add(Torus32 *res, const Torus32 *a, const int32_t *bits, int32_t N) {
    uint32_t *r = (uint32_t *)res;
    for (int32_t i = 0; i < N; ++i) {
        if (!bits[i]) continue;
        for (int32_t j = 0; j < N - i; ++j) r[i + j] += (uint32_t)a[j];
        for (int32_t j = N - i; j < N; ++j) r[i + j - N] -= (uint32_t)a[j];
    }
}

/* ---------------------------------------------------------------------------
 * fp64 FFT evaluator (use_ntt = 3): upstream TFHE's way of multiplying, restated as a plain
 * radix-2 transform.  A real polynomial mod X^N+1 is folded into N/2 complex points
 * z_j = (p_j + i p_{j+N/2}) w^j, w = exp(i pi / N), whose DFT of size N/2 gives the values at
 * the roots x with x^{N/2} = i; products are pointwise there.  Approximate (see header).
 * ------------------------------------------------------------------------- */
typedef struct FftTab { int32_t N; double *cr, *ci, *tr, *ti; } FftTab;   /* FFT twiddles, twist */
static FftTab g_fft_tabs[ORC_MAX_TABS];
static int g_fft_ntabs = 0;
static pthread_mutex_t g_fft_mtx = PTHREAD_MUTEX_INITIALIZER;

static const FftTab *fft_tab(int32_t N) {
    pthread_mutex_lock(&g_fft_mtx);
    for (int i = 0; i < g_fft_ntabs; ++i)
        if (g_fft_tabs[i].N == N) { pthread_mutex_unlock(&g_fft_mtx); return &g_fft_tabs[i]; }
    if (g_fft_ntabs == ORC_MAX_TABS) abort();     /* more distinct ring degrees than this test helper expects */
    FftTab *t = &g_fft_tabs[g_fft_ntabs];
    const int32_t M = N / 2;
    const double pi = 3.14159265358979323846;
    t->N = N;
    t->cr = (double *)malloc(sizeof(double) * M * 2);  t->ci = t->cr + M;      /* exp(-2 pi i k / M), k < M/2 used */
    t->tr = (double *)malloc(sizeof(double) * M * 2);  t->ti = t->tr + M;      /* exp(i pi j / N) */
    for (int32_t k = 0; k < M; ++k) {
        t->cr[k] = cos(2.0 * pi * k / M);  t->ci[k] = -sin(2.0 * pi * k / M);
        t->tr[k] = cos(pi * k / N);        t->ti[k] = sin(pi * k / N);
    }
    ++g_fft_ntabs;
    pthread_mutex_unlock(&g_fft_mtx);
    return t;
}
/* forward: natural order in, bit-reversed order out (decimation in frequency) */
static void fft_fwd(double *re, double *im, const FftTab *t) {
    const int32_t M = t->N / 2;
    for (int32_t h = M / 2, step = 1; h >= 1; h >>= 1, step <<= 1)
        for (int32_t b = 0; b < M; b += 2 * h)
            for (int32_t j = 0; j < h; ++j) {
                const double wr = t->cr[j * step], wi = t->ci[j * step];
                const double ur = re[b + j], ui = im[b + j], vr = re[b + j + h], vi = im[b + j + h];
                re[b + j] = ur + vr;  im[b + j] = ui + vi;
                const double dr = ur - vr, di = ui - vi;
                re[b + j + h] = dr * wr - di * wi;  im[b + j + h] = dr * wi + di * wr;
            }
}
/* inverse: bit-reversed in, natural out (decimation in time, conjugate twiddles), unscaled */
static void fft_inv(double *re, double *im, const FftTab *t) {
    const int32_t M = t->N / 2;
    for (int32_t h = 1, step = M / 2; h < M; h <<= 1, step >>= 1)
        for (int32_t b = 0; b < M; b += 2 * h)
            for (int32_t j = 0; j < h; ++j) {
                const double wr = t->cr[j * step], wi = -t->ci[j * step];
                const double xr = re[b + j + h], xi = im[b + j + h];
                const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                const double ur = re[b + j], ui = im[b + j];
                re[b + j] = ur + vr;  im[b + j] = ui + vi;
                re[b + j + h] = ur - vr;  im[b + j + h] = ui - vi;
            }
}
static void fft_from_i32(double *re, double *im, const int32_t *p, const FftTab *t) {
    const int32_t M = t->N / 2;
    for (int32_t j = 0; j < M; ++j) {
        const double a = (double)p[j], b = (double)p[j + M];
        re[j] = a * t->tr[j] - b * t->ti[j];
        im[j] = a * t->ti[j] + b * t->tr[j];
    }
    fft_fwd(re, im, t);
}
/* acc[j] += round(value_j) mod 2^32 */
static void fft_add_to_torus(uint32_t *acc, double *re, double *im, const FftTab *t) {
    const int32_t M = t->N / 2;
    fft_inv(re, im, t);
    const double sc = 1.0 / M;
    for (int32_t j = 0; j < M; ++j) {
        const double zr = re[j] * sc, zi = im[j] * sc;
        const double a = zr * t->tr[j] + zi * t->ti[j];          /* times conj(w^j) */
        const double b = zi * t->tr[j] - zr * t->ti[j];
        acc[j] += (uint32_t)(int64_t)llround(a);
        acc[j + M] += (uint32_t)(int64_t)llround(b);
    }
}

OrcKeySet *orc_keygen(const OrcParams *p, uint64_t seed) {
    OrcKeySet *ks = (OrcKeySet *)calloc(1, sizeof(OrcKeySet));
    ks->p = *p;
    const int32_t n = p->n, N = p->N, k = p->k, l = p->l, kpl = (k + 1) * l;
    const int32_t t = p->ks_t, base = 1 << p->ks_basebit;
    OrcRng rng; orc_rng_seed(&rng, seed);

    ks->lwe_key = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int32_t i = 0; i < n; ++i) ks->lwe_key[i] = (int32_t)(orc_rng_next(&rng) >> 63);
    ks->tlwe_key = (int32_t *)malloc(sizeof(int32_t) * k * N);
    for (int32_t i = 0; i < k * N; ++i) ks->tlwe_key[i] = (int32_t)(orc_rng_next(&rng) >> 63);

    /* BK_i = TGSW(lwe_key[i]): (k+1)l TLWE zero-encryptions + mu * gadget */
    ks->bk = (Torus32 *)malloc(sizeof(Torus32) * orc_bk_words(p));
    for (int32_t i = 0; i < n; ++i) {
        for (int32_t row = 0; row < kpl; ++row) {
            Torus32 *smp = ks->bk + ((size_t)i * kpl + row) * (size_t)(k + 1) * N;
            Torus32 *body = smp + (size_t)k * N;
            for (int32_t u = 0; u < k; ++u)
                for (int32_t j = 0; j < N; ++j) smp[(size_t)u * N + j] = orc_rng_torus(&rng);
            for (int32_t j = 0; j < N; ++j) body[j] = orc_dtot32(orc_rng_gauss(&rng, p->bk_stdev));
            for (int32_t u = 0; u < k; ++u)
                negacyclic_mul_bits_add(body, smp + (size_t)u * N, ks->tlwe_key + (size_t)u * N, N);
            /* tGswAddMuIntH: row (bloc,j) adds mu*2^{32-(j+1)Bgbit} to coef 0 of poly bloc */
            const int32_t bloc = row / l, jj = row % l;
            const uint32_t h = 1u << (32 - (jj + 1) * p->Bgbit);
            smp[(size_t)bloc * N] = (Torus32)((uint32_t)smp[(size_t)bloc * N] + (uint32_t)ks->lwe_key[i] * h);
        }
    }

    /* KSK[i][j][v] = LWE_s(v * s'_i * 2^{32-(j+1)basebit}); v = 0 rows stay zero */
    ks->ksk = (Torus32 *)calloc(orc_ksk_words(p), sizeof(Torus32));
    for (int32_t i = 0; i < k * N; ++i)
        for (int32_t j = 0; j < t; ++j)
            for (int32_t v = 1; v < base; ++v) {
                Torus32 *row = ks->ksk + (((size_t)i * t + j) * base + v) * (size_t)(n + 1);
                const uint32_t mess = (uint32_t)(ks->tlwe_key[i] * v) << (32 - (j + 1) * p->ks_basebit);
                uint32_t b = 0;
                for (int32_t q = 0; q < n; ++q) {
                    row[q] = orc_rng_torus(&rng);
                    b += (uint32_t)row[q] * (uint32_t)ks->lwe_key[q];
                }
                b += mess + (uint32_t)orc_dtot32(orc_rng_gauss(&rng, p->ks_stdev));
                row[n] = (Torus32)b;
            }

    /* image for the two-prime evaluator: [poly][prime][N], Montgomery form times N^-1 */
    {
        const FastTab *ft = fast_tab(N);
        const size_t npoly_f = (size_t)n * kpl * (k + 1);
        ks->bk_fast = (uint32_t *)malloc(sizeof(uint32_t) * npoly_f * 2 * N);
        for (size_t qq = 0; qq < npoly_f; ++qq)
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t *dst = ks->bk_fast + (qq * 2 + pr) * N;
                const Torus32 *src = ks->bk + qq * N;
                for (int32_t j = 0; j < N; ++j) dst[j] = fast_from_i32(src[j], FP[pr]);
                fast_fwd(dst, ft, pr);
                for (int32_t j = 0; j < N; ++j) dst[j] = (uint32_t)((uint64_t)dst[j] * ft->scale[pr] % FP[pr]);
            }
    }
    /* fp64 FFT image for the approximate evaluator (use_ntt = 3) */
    {
        const FftTab *ft = fft_tab(N);
        const size_t npoly_f = (size_t)n * kpl * (k + 1);
        ks->bk_fft = (double *)malloc(sizeof(double) * npoly_f * N);
        for (size_t qq = 0; qq < npoly_f; ++qq)
            fft_from_i32(ks->bk_fft + qq * N, ks->bk_fft + qq * N + N / 2, ks->bk + qq * N, ft);
    }
    /* evaluation-domain image of BK (tfhe: LweBootstrappingKeyFFT) */
    const NttTab *tab = ntt_tab(N);
    const size_t npoly = (size_t)n * kpl * (k + 1);
    ks->bk_ntt = (uint64_t *)malloc(sizeof(uint64_t) * npoly * N);
    for (size_t q = 0; q < npoly; ++q) {
        uint64_t *dst = ks->bk_ntt + q * N;
        const Torus32 *src = ks->bk + q * N;
        for (int32_t j = 0; j < N; ++j) dst[j] = gl_from_i32(src[j]);
        gl_ntt_fwd(dst, tab);
    }
    return ks;
}

void orc_keyset_free(OrcKeySet *ks) {
    if (!ks) return;
    free(ks->lwe_key); free(ks->tlwe_key); free(ks->bk); free(ks->ksk); free(ks->bk_ntt); free(ks->bk_fast); free(ks->bk_fft); free(ks->bk_fft4);
    free(ks);
}

/* ------------------------------------------------------------------ */
/* encrypt / decrypt (tfhe: bootsSymEncrypt, lwePhase, bootsSymDecrypt) */
/* draw order: one gaussian for the body, then n uniform mask words     */
/* ------------------------------------------------------------------ */
void orc_encrypt_bit(const OrcKeySet *ks, OrcRng *rng, int32_t message, Torus32 *ct) {
    const int32_t n = ks->p.n;
    const Torus32 mu = message ? (Torus32)(1 << 29) : (Torus32)(-(1 << 29));
    uint32_t b = (uint32_t)mu + (uint32_t)orc_dtot32(orc_rng_gauss(rng, ks->p.ks_stdev));
    for (int32_t i = 0; i < n; ++i) {
        ct[i] = orc_rng_torus(rng);
        b += (uint32_t)ct[i] * (uint32_t)ks->lwe_key[i];
    }
    ct[n] = (Torus32)b;
}

Torus32 orc_phase(const OrcKeySet *ks, const Torus32 *ct) {
    const int32_t n = ks->p.n;
    uint32_t ph = (uint32_t)ct[n];
    for (int32_t i = 0; i < n; ++i) ph -= (uint32_t)ct[i] * (uint32_t)ks->lwe_key[i];
    return (Torus32)ph;
}

int32_t orc_decrypt_bit(const OrcKeySet *ks, const Torus32 *ct) { return orc_phase(ks, ct) > 0 ? 1 : 0; }

/* ------------------------------------------------------------------ */
/* gadget decomposition (tfhe: tGswTorus32PolynomialDecompH)            */
/* ------------------------------------------------------------------ */
void orc_decompose(int32_t *digits, const Torus32 *poly, const OrcParams *p) {
    const int32_t N = p->N, l = p->l, Bgbit = p->Bgbit;
    const uint32_t mask = (1u << Bgbit) - 1u;
    const int32_t halfBg = 1 << (Bgbit - 1);
    uint32_t offset = 0;
    for (int32_t j = 1; j <= l; ++j) offset += (uint32_t)halfBg << (32 - j * Bgbit);
    for (int32_t q = 0; q < l; ++q) {
        const int32_t decal = 32 - (q + 1) * Bgbit;
        for (int32_t j = 0; j < N; ++j)
            digits[(size_t)q * N + j] = (int32_t)((((uint32_t)poly[j] + offset) >> decal) & mask) - halfBg;
    }
}

/* (X^a - 1) * src, a in [0,2N)  (tfhe: torusPolynomialMulByXaiMinusOne) */
static void mul_xai_minus_one(Torus32 *out, int32_t a, const Torus32 *in, int32_t N) {
    const uint32_t *s = (const uint32_t *)in; uint32_t *o = (uint32_t *)out;
    if (a < N) {
        for (int32_t i = 0; i < a; ++i) o[i] = 0u - s[i - a + N] - s[i];
        for (int32_t i = a; i < N; ++i) o[i] = s[i - a] - s[i];
    } else {
        const int32_t aa = a - N;
        for (int32_t i = 0; i < aa; ++i) o[i] = s[i - aa + N] - s[i];
        for (int32_t i = aa; i < N; ++i) o[i] = 0u - s[i - aa] - s[i];
    }
}
/* X^a * src (tfhe: torusPolynomialMulByXai) */
static void mul_xai(Torus32 *out, int32_t a, const Torus32 *in, int32_t N) {
    const uint32_t *s = (const uint32_t *)in; uint32_t *o = (uint32_t *)out;
    if (a < N) {
        for (int32_t i = 0; i < a; ++i) o[i] = 0u - s[i - a + N];
        for (int32_t i = a; i < N; ++i) o[i] = s[i - a];
    } else {
        const int32_t aa = a - N;
        for (int32_t i = 0; i < aa; ++i) o[i] = s[i - aa + N];
        for (int32_t i = aa; i < N; ++i) o[i] = 0u - s[i - aa];
    }
}

/* one blind-rotate step (tfhe: tfhe_MuxRotate_FFT + tGswFFTExternMulToTLwe,
 * with the product exact):  acc += BK_i (.) ((X^barai - 1) acc)            */
void orc_cmux_rotate(const OrcKeySet *ks, int32_t i, int32_t barai, Torus32 *acc, int use_ntt) {
    const OrcParams *p = &ks->p;
    const int32_t N = p->N, k = p->k, l = p->l, kpl = (k + 1) * l;
    Torus32 *d = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(k + 1) * N);
    int32_t *dig = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    for (int32_t u = 0; u <= k; ++u) {
        mul_xai_minus_one(d + (size_t)u * N, barai, acc + (size_t)u * N, N);
        orc_decompose(dig + (size_t)u * l * N, d + (size_t)u * N, p);
    }
    if (use_ntt == 3) {
        const FftTab *ft = fft_tab(N);
        const int32_t M = N / 2;
        double *dn = (double *)malloc(sizeof(double) * (size_t)(kpl + 1) * N);
        double *sr = dn + (size_t)kpl * N, *si = sr + M;
        for (int32_t q = 0; q < kpl; ++q) fft_from_i32(dn + (size_t)q * N, dn + (size_t)q * N + M, dig + (size_t)q * N, ft);
        for (int32_t w = 0; w <= k; ++w) {
            for (int32_t j = 0; j < M; ++j) { sr[j] = 0.0; si[j] = 0.0; }
            for (int32_t q = 0; q < kpl; ++q) {
                const double *br = ks->bk_fft + (((size_t)i * kpl + q) * (k + 1) + w) * N, *bi = br + M;
                const double *xr = dn + (size_t)q * N, *xi = xr + M;
                for (int32_t j = 0; j < M; ++j) {
                    sr[j] += xr[j] * br[j] - xi[j] * bi[j];
                    si[j] += xr[j] * bi[j] + xi[j] * br[j];
                }
            }
            fft_add_to_torus((uint32_t *)(acc + (size_t)w * N), sr, si, ft);
        }
        free(dn);
    } else if (use_ntt == 2) {
        const FastTab *ft = fast_tab(N);
        uint32_t *dn = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(kpl + 2 * (k + 1)) * N);
        uint32_t *res = dn + (size_t)kpl * N;                 /* [prime][k+1][N] canonical residues */
        uint64_t *sum = (uint64_t *)malloc(sizeof(uint64_t) * N);
        for (int pr = 0; pr < 2; ++pr) {
            const uint32_t P = FP[pr], pinv = ft->pinv[pr];
            for (int32_t q = 0; q < kpl; ++q) {               /* kpl forward transforms */
                uint32_t *x = dn + (size_t)q * N;
                /* gadget digits: |d| <= Bg/2 < P, so the canonical residue is d or d + P (no division) */
                for (int32_t j = 0; j < N; ++j) { const int32_t dj = dig[(size_t)q * N + j]; x[j] = (uint32_t)(dj + ((dj >> 31) & (int32_t)P)); }
                fast_fwd(x, ft, pr);
            }
            for (int32_t w = 0; w <= k; ++w) {                /* MAC against the key image, inverse */
                for (int32_t j = 0; j < N; ++j) sum[j] = 0;
                for (int32_t q = 0; q < kpl; ++q) {
                    const uint32_t *bkq = ks->bk_fast + ((((size_t)i * kpl + q) * (k + 1) + w) * 2 + pr) * N;
                    const uint32_t *x = dn + (size_t)q * N;
                    fast_mac(sum, x, bkq, N);                                             /* < kpl * P^2 < 2^57 */
                }
                uint32_t *r = res + ((size_t)pr * (k + 1) + w) * N;
                for (int32_t j = 0; j < N; ++j) {
                    const uint32_t m = (uint32_t)sum[j] * pinv;
                    /* sum < kpl P^2 < 2^32 * 0.2 P, so the reduced value is below 1.2 P: one conditional subtraction */
                    const uint32_t rr = (uint32_t)((sum[j] + (uint64_t)m * P) >> 32);
                    r[j] = rr >= P ? rr - P : rr;
                }
                fast_inv(r, ft, pr);
            }
        }
        for (int32_t w = 0; w <= k; ++w) {
            uint32_t *a = (uint32_t *)(acc + (size_t)w * N);
            const uint32_t *r0 = res + (size_t)w * N, *r1 = res + ((size_t)(k + 1) + w) * N;
            for (int32_t j = 0; j < N; ++j) a[j] += (uint32_t)fast_crt(r0[j], r1[j], ft);
        }
        free(dn); free(sum);
    } else if (use_ntt) {
        const NttTab *tab = ntt_tab(N);
        uint64_t *dn = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(kpl + k + 1) * N);
        uint64_t *sum = dn + (size_t)kpl * N;
        for (int32_t q = 0; q < kpl; ++q) {
            uint64_t *x = dn + (size_t)q * N;
            for (int32_t j = 0; j < N; ++j) x[j] = gl_from_i32(dig[(size_t)q * N + j]);
            gl_ntt_fwd(x, tab);
        }
        for (int32_t w = 0; w <= k; ++w) {
            uint64_t *s = sum + (size_t)w * N;
            for (int32_t j = 0; j < N; ++j) s[j] = 0;
            for (int32_t q = 0; q < kpl; ++q) {
                const uint64_t *bkq = ks->bk_ntt + (((size_t)i * kpl + q) * (k + 1) + w) * N;
                const uint64_t *x = dn + (size_t)q * N;
                for (int32_t j = 0; j < N; ++j) s[j] = gl_add(s[j], gl_mul(x[j], bkq[j]));
            }
            gl_ntt_inv(s, tab);
            uint32_t *a = (uint32_t *)(acc + (size_t)w * N);
            for (int32_t j = 0; j < N; ++j) a[j] += (uint32_t)gl_to_torus(s[j]);
        }
        free(dn);
    } else {
        Torus32 *prod = (Torus32 *)malloc(sizeof(Torus32) * N);
        for (int32_t w = 0; w <= k; ++w) {
            uint32_t *a = (uint32_t *)(acc + (size_t)w * N);
            for (int32_t q = 0; q < kpl; ++q) {
                const Torus32 *bkq = ks->bk + (((size_t)i * kpl + q) * (k + 1) + w) * N;
                orc_negacyclic_schoolbook(prod, dig + (size_t)q * N, bkq, N);
                for (int32_t j = 0; j < N; ++j) a[j] += (uint32_t)prod[j];
            }
        }
        free(prod);
    }
    free(d); free(dig);
}

/* tfhe: tfhe_blindRotateAndExtract_FFT up to (not including) the extract */
void orc_blind_rotate(const OrcKeySet *ks, const int32_t *bara, int32_t barb,
                      Torus32 mu, Torus32 *acc, int use_ntt) {
    const OrcParams *p = &ks->p;
    const int32_t N = p->N, k = p->k, n = p->n;
    Torus32 *tv = (Torus32 *)malloc(sizeof(Torus32) * N);
    for (int32_t j = 0; j < N; ++j) tv[j] = mu;
    memset(acc, 0, sizeof(Torus32) * (size_t)k * N);
    if (barb != 0) mul_xai(acc + (size_t)k * N, 2 * N - barb, tv, N);
    else memcpy(acc + (size_t)k * N, tv, sizeof(Torus32) * N);
    if (use_ntt == 4) use_ntt = orc_fft4_available() ? 4 : 3;
    if (use_ntt == 4) {
        orc_fft4_blind_rotate_steps(ks, bara, acc);      /* the AVX2 + FMA stand-in (fft_standin.c), not an oracle mode */
    } else {
        for (int32_t i = 0; i < n; ++i) {
            if (bara[i] == 0) continue;
            orc_cmux_rotate(ks, i, bara[i], acc, use_ntt);
        }
    }
    free(tv);
}

/* tfhe: tLweExtractLweSampleIndex(index 0) */
void orc_sample_extract(const OrcParams *p, const Torus32 *acc, Torus32 *u) {
    const int32_t N = p->N, k = p->k;
    for (int32_t i = 0; i < k; ++i) {
        const uint32_t *a = (const uint32_t *)(acc + (size_t)i * N);
        u[(size_t)i * N] = (Torus32)a[0];
        for (int32_t j = 1; j < N; ++j) u[(size_t)i * N + j] = (Torus32)(0u - a[N - j]);
    }
    u[(size_t)k * N] = acc[(size_t)k * N];
}

/* tfhe: lweKeySwitch / lweKeySwitchTranslate_fromArray */
void orc_keyswitch(const OrcKeySet *ks, const Torus32 *u, Torus32 *ct) {
    const OrcParams *p = &ks->p;
    const int32_t n = p->n, nin = p->k * p->N, t = p->ks_t, basebit = p->ks_basebit;
    const int32_t base = 1 << basebit;
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const uint32_t mask = (uint32_t)base - 1u;
    uint32_t *r = (uint32_t *)ct;
    for (int32_t q = 0; q < n; ++q) r[q] = 0;
    r[n] = (uint32_t)u[nin];
    for (int32_t i = 0; i < nin; ++i) {
        const uint32_t aibar = (uint32_t)u[i] + prec_offset;
        for (int32_t j = 0; j < t; ++j) {
            const uint32_t aij = (aibar >> (32 - (j + 1) * basebit)) & mask;
            if (aij == 0) continue;
            const uint32_t *row = (const uint32_t *)(ks->ksk + (((size_t)i * t + j) * base + aij) * (size_t)(n + 1));
            for (int32_t q = 0; q <= n; ++q) r[q] -= row[q];
        }
    }
}

/* tfhe: tfhe_bootstrap_woKS_FFT */
void orc_bootstrap_woks(const OrcKeySet *ks, const Torus32 *lin, Torus32 mu, Torus32 *u, int use_ntt) {
    const OrcParams *p = &ks->p;
    const int32_t n = p->n, N = p->N, k = p->k;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * n);
    Torus32 *acc = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(k + 1) * N);
    const int32_t barb = orc_modswitch(lin[n], 2 * N);
    for (int32_t i = 0; i < n; ++i) bara[i] = orc_modswitch(lin[i], 2 * N);
    orc_blind_rotate(ks, bara, barb, mu, acc, use_ntt);
    orc_sample_extract(p, acc, u);
    free(bara); free(acc);
}

/* ------------------------------------------------------------------ */
/* gates (tfhe: boot-gates.cpp)                                         */
/* ------------------------------------------------------------------ */
static const struct { int32_t c8; int32_t sa, sb; } GATE_TAB[ORC_NGATES2] = {
    /* NAND  */ { 1, -1, -1}, /* OR    */ { 1,  1,  1}, /* AND   */ {-1,  1,  1},
    /* NOR   */ {-1, -1, -1}, /* XOR   */ { 2,  2,  2}, /* XNOR  */ {-2, -2, -2},
    /* ANDNY */ {-1, -1,  1}, /* ANDYN */ {-1,  1, -1}, /* ORNY  */ { 1, -1,  1},
    /* ORYN  */ { 1,  1, -1},
};

void orc_gate_prelude(const OrcParams *p, int gate, const Torus32 *ca, const Torus32 *cb, Torus32 *t) {
    const int32_t n = p->n;
    const uint32_t sa = (uint32_t)GATE_TAB[gate].sa, sb = (uint32_t)GATE_TAB[gate].sb;
    for (int32_t i = 0; i <= n; ++i) t[i] = (Torus32)(sa * (uint32_t)ca[i] + sb * (uint32_t)cb[i]);
    t[n] = (Torus32)((uint32_t)t[n] + (uint32_t)orc_modswitch_to_torus(GATE_TAB[gate].c8, 8));
}

void orc_gate2(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb, int use_ntt) {
    const OrcParams *p = &ks->p;
    Torus32 *t = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(p->n + 1));
    Torus32 *u = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(p->k * p->N + 1));
    orc_gate_prelude(p, gate, ca, cb, t);
    orc_bootstrap_woks(ks, t, orc_modswitch_to_torus(1, 8), u, use_ntt);
    orc_keyswitch(ks, u, out);
    free(t); free(u);
}

void orc_mux(const OrcKeySet *ks, Torus32 *out, const Torus32 *a, const Torus32 *b, const Torus32 *c, int use_ntt) {
    const OrcParams *p = &ks->p;
    const int32_t n = p->n, nin = p->k * p->N;
    const Torus32 mu = orc_modswitch_to_torus(1, 8);
    Torus32 *t = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(n + 1));
    Torus32 *u1 = (Torus32 *)malloc(sizeof(Torus32) * (size_t)(nin + 1) * 2), *u2 = u1 + nin + 1;
    orc_gate_prelude(p, ORC_AND, a, b, t);      /* (0,-1/8) + a + b */
    orc_bootstrap_woks(ks, t, mu, u1, use_ntt);
    orc_gate_prelude(p, ORC_ANDNY, a, c, t);    /* (0,-1/8) - a + c */
    orc_bootstrap_woks(ks, t, mu, u2, use_ntt);
    for (int32_t i = 0; i <= nin; ++i) u1[i] = (Torus32)((uint32_t)u1[i] + (uint32_t)u2[i]);
    u1[nin] = (Torus32)((uint32_t)u1[nin] + (uint32_t)mu);
    orc_keyswitch(ks, u1, out);
    free(t); free(u1);
}

void orc_not(const OrcParams *p, Torus32 *out, const Torus32 *a) {
    for (int32_t i = 0; i <= p->n; ++i) out[i] = (Torus32)(0u - (uint32_t)a[i]);
}

void orc_constant(const OrcParams *p, Torus32 *out, int32_t value) {
    for (int32_t i = 0; i < p->n; ++i) out[i] = 0;
    const Torus32 mu = orc_modswitch_to_torus(1, 8);
    out[p->n] = value ? mu : (Torus32)(0u - (uint32_t)mu);
}

/* ------------------------------------------------------------------ */
/* threaded batch for the cpu_baseline leg                              */
/* ------------------------------------------------------------------ */
typedef struct BatchJob {
    const OrcKeySet *ks; int gate; Torus32 *out; const Torus32 *ca, *cb;
    int32_t begin, end;
    int mode;
} BatchJob;

static void *batch_worker(void *arg) {
    BatchJob *j = (BatchJob *)arg;
    const size_t w = (size_t)j->ks->p.n + 1;
    for (int32_t g = j->begin; g < j->end; ++g)
        orc_gate2(j->ks, j->gate, j->out + g * w, j->ca + g * w, j->cb + g * w, j->mode);
    return NULL;
}

void orc_gate2_batch(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb,
                     int32_t count, int32_t nthreads) {
    orc_gate2_batch_mode(ks, gate, out, ca, cb, count, nthreads, 2);
}

void orc_gate2_batch_mode(const OrcKeySet *ks, int gate, Torus32 *out, const Torus32 *ca, const Torus32 *cb,
                          int32_t count, int32_t nthreads, int use_ntt) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > count) nthreads = count > 0 ? count : 1;
    (void)fast_tab(ks->p.N);
    (void)fft_tab(ks->p.N);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    BatchJob *jobs = (BatchJob *)malloc(sizeof(BatchJob) * nthreads);
    for (int32_t t = 0; t < nthreads; ++t) {
        jobs[t] = (BatchJob){ks, gate, out, ca, cb,
                             (int32_t)((int64_t)count * t / nthreads),
                             (int32_t)((int64_t)count * (t + 1) / nthreads), use_ntt};
        pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    for (int32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
