/*
 * fft_standin.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE, AND NOT THE PARITY ORACLE.
 *
 * use_ntt = 4: a cost-faithful stand-in for the CPU path the reference actually links.
 * /root/reference/CMakeLists.txt:9-15 links a libtfhe FFT flavour (absent from
 * /root/reference and from this image); upstream TFHE multiplies polynomials mod X^N+1 with
 * an fp64 complex FFT of N/2 points, its fastest flavour (spqlios-fma) in AVX2 + FMA
 * assembly.  This file restates that evaluator -- folded N/2-point transform, split re/im
 * arrays, AVX2 + FMA intrinsics, key image kept in the evaluation domain -- so that
 * bench.py's cpu_baseline can time something that costs what upstream's CPU path costs on
 * the GPU box's host cores.  It is APPROXIMATE like upstream (the low-order noise bits of a
 * ciphertext differ from the exact evaluators of tfhe_oracle.c); it is checked at decrypt
 * level against the exact two-prime evaluator (tests/test_host_cpu.py) and is never the
 * thing a parity test compares with.
 *
 * Every function that uses AVX2/FMA carries a target attribute; the file itself is compiled
 * for baseline x86-64, and orc_fft4_available() says whether the host can run it (the
 * callers fall back to the scalar fp64 evaluator, use_ntt = 3, when it cannot).
 *
 * Transform: M = N/2 complex points z_j = (p_j + i p_{j+M}) w^j, w = exp(i pi / N).
 * Forward = decimation in frequency, radix 2, half sizes M/2 .. 4 vectorised four butterflies
 * per instruction, then the last two stages (a 4-point DFT inside every group of four
 * consecutive points) on sixteen points at a time through one 4x4 register transpose.  The
 * result is left in that transposed, bit-reversed order: products are pointwise, the inverse
 * starts by undoing exactly that pass, and the key image is stored in the same order.
 */
#include "tfhe_oracle.h"
#include "fft_standin.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define FFT4 __attribute__((target("avx2,fma")))
#define FFT4_MAX_TABS 8
/* A complex array is [re M | pad | im M]: with im exactly M doubles (4 KB at N = 1024) behind re, every load of one half
 * aliases a store to the other in the low 12 address bits and waits for it (measured: a transform 2,130 -> 1,620 cycles). */
#define FFT4_IM(M) ((size_t)(M) + 8)
#define FFT4_XSTRIDE(N) ((size_t)(N) + 24)    /* one complex array incl. the pad, plus a line so that arrays do not share L1 sets */

int orc_fft4_available(void) {
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}

typedef struct Fft4Tab {
    int32_t N;
    double *wr, *wi;   /* stage tables: half size h lives at offset M - 2h, h entries: exp(-2 pi i j / 2h) */
    double *tr, *ti;   /* twist exp(i pi j / N), j < M */
} Fft4Tab;
static Fft4Tab g_tabs[FFT4_MAX_TABS];
static int g_ntabs = 0;
static pthread_mutex_t g_mtx = PTHREAD_MUTEX_INITIALIZER;

static void *alloc64(size_t bytes) {
    void *p = NULL;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) abort();
    return p;
}

static const Fft4Tab *fft4_tab(int32_t N) {
    pthread_mutex_lock(&g_mtx);
    for (int i = 0; i < g_ntabs; ++i)
        if (g_tabs[i].N == N) { pthread_mutex_unlock(&g_mtx); return &g_tabs[i]; }
    if (g_ntabs == FFT4_MAX_TABS) abort();
    Fft4Tab *t = &g_tabs[g_ntabs];
    const int32_t M = N / 2;
    const double pi = 3.14159265358979323846;
    t->N = N;
    t->wr = (double *)alloc64(sizeof(double) * M);  t->wi = (double *)alloc64(sizeof(double) * M);
    t->tr = (double *)alloc64(sizeof(double) * M);  t->ti = (double *)alloc64(sizeof(double) * M);
    for (int32_t h = M / 2; h >= 1; h >>= 1)
        for (int32_t j = 0; j < h; ++j) {
            t->wr[M - 2 * h + j] = cos(pi * j / h);
            t->wi[M - 2 * h + j] = -sin(pi * j / h);
        }
    for (int32_t j = 0; j < M; ++j) { t->tr[j] = cos(pi * j / N);  t->ti[j] = sin(pi * j / N); }
    ++g_ntabs;
    pthread_mutex_unlock(&g_mtx);
    return t;
}

#define TRANSPOSE4(a0, a1, a2, a3, r0, r1, r2, r3) do {                 \
        const __m256d t0_ = _mm256_unpacklo_pd(r0, r1), t1_ = _mm256_unpackhi_pd(r0, r1); \
        const __m256d t2_ = _mm256_unpacklo_pd(r2, r3), t3_ = _mm256_unpackhi_pd(r2, r3); \
        a0 = _mm256_permute2f128_pd(t0_, t2_, 0x20);  a1 = _mm256_permute2f128_pd(t1_, t3_, 0x20); \
        a2 = _mm256_permute2f128_pd(t0_, t2_, 0x31);  a3 = _mm256_permute2f128_pd(t1_, t3_, 0x31); \
    } while (0)

/* natural order in, transposed bit-reversed order out */
FFT4 static void fft4_fwd(double *re, double *im, const Fft4Tab *t) {
    const int32_t M = t->N / 2;
    for (int32_t h = M / 2; h >= 4; h >>= 1) {
        const double *wr = t->wr + (M - 2 * h), *wi = t->wi + (M - 2 * h);
        for (int32_t b = 0; b < M; b += 2 * h)
            for (int32_t j = 0; j < h; j += 4) {
                double *pr = re + b + j, *pi_ = im + b + j;
                const __m256d ur = _mm256_load_pd(pr), ui = _mm256_load_pd(pi_);
                const __m256d vr = _mm256_load_pd(pr + h), vi = _mm256_load_pd(pi_ + h);
                const __m256d cr = _mm256_load_pd(wr + j), ci = _mm256_load_pd(wi + j);
                _mm256_store_pd(pr, _mm256_add_pd(ur, vr));
                _mm256_store_pd(pi_, _mm256_add_pd(ui, vi));
                const __m256d dr = _mm256_sub_pd(ur, vr), di = _mm256_sub_pd(ui, vi);
                _mm256_store_pd(pr + h, _mm256_fmsub_pd(dr, cr, _mm256_mul_pd(di, ci)));
                _mm256_store_pd(pi_ + h, _mm256_fmadd_pd(dr, ci, _mm256_mul_pd(di, cr)));
            }
    }
    for (int32_t g = 0; g < M; g += 16) {
        __m256d a0, a1, a2, a3, c0, c1, c2, c3;
        TRANSPOSE4(a0, a1, a2, a3, _mm256_load_pd(re + g), _mm256_load_pd(re + g + 4), _mm256_load_pd(re + g + 8), _mm256_load_pd(re + g + 12));
        TRANSPOSE4(c0, c1, c2, c3, _mm256_load_pd(im + g), _mm256_load_pd(im + g + 4), _mm256_load_pd(im + g + 8), _mm256_load_pd(im + g + 12));
        /* half size 2: twiddles 1 and -i */
        const __m256d p0r = _mm256_add_pd(a0, a2), p0i = _mm256_add_pd(c0, c2);
        const __m256d p2r = _mm256_sub_pd(a0, a2), p2i = _mm256_sub_pd(c0, c2);
        const __m256d p1r = _mm256_add_pd(a1, a3), p1i = _mm256_add_pd(c1, c3);
        const __m256d p3r = _mm256_sub_pd(c1, c3), p3i = _mm256_sub_pd(a3, a1);   /* (x + iy)(-i) = y - ix */
        /* half size 1 */
        _mm256_store_pd(re + g,      _mm256_add_pd(p0r, p1r));  _mm256_store_pd(im + g,      _mm256_add_pd(p0i, p1i));
        _mm256_store_pd(re + g + 4,  _mm256_sub_pd(p0r, p1r));  _mm256_store_pd(im + g + 4,  _mm256_sub_pd(p0i, p1i));
        _mm256_store_pd(re + g + 8,  _mm256_add_pd(p2r, p3r));  _mm256_store_pd(im + g + 8,  _mm256_add_pd(p2i, p3i));
        _mm256_store_pd(re + g + 12, _mm256_sub_pd(p2r, p3r));  _mm256_store_pd(im + g + 12, _mm256_sub_pd(p2i, p3i));
    }
}

/* the inverse of fft4_fwd up to the factor M (which the key image carries) */
FFT4 static void fft4_inv(double *re, double *im, const Fft4Tab *t) {
    const int32_t M = t->N / 2;
    for (int32_t g = 0; g < M; g += 16) {
        const __m256d o0r = _mm256_load_pd(re + g), o1r = _mm256_load_pd(re + g + 4), o2r = _mm256_load_pd(re + g + 8), o3r = _mm256_load_pd(re + g + 12);
        const __m256d o0i = _mm256_load_pd(im + g), o1i = _mm256_load_pd(im + g + 4), o2i = _mm256_load_pd(im + g + 8), o3i = _mm256_load_pd(im + g + 12);
        const __m256d p0r = _mm256_add_pd(o0r, o1r), p0i = _mm256_add_pd(o0i, o1i);
        const __m256d p1r = _mm256_sub_pd(o0r, o1r), p1i = _mm256_sub_pd(o0i, o1i);
        const __m256d p2r = _mm256_add_pd(o2r, o3r), p2i = _mm256_add_pd(o2i, o3i);
        const __m256d p3r = _mm256_sub_pd(o2r, o3r), p3i = _mm256_sub_pd(o2i, o3i);
        /* half size 2 with the conjugate twiddles 1 and +i: v = p3 * i = -p3i + i p3r */
        const __m256d a0 = _mm256_add_pd(p0r, p2r), c0 = _mm256_add_pd(p0i, p2i);
        const __m256d a2 = _mm256_sub_pd(p0r, p2r), c2 = _mm256_sub_pd(p0i, p2i);
        const __m256d a1 = _mm256_sub_pd(p1r, p3i), c1 = _mm256_add_pd(p1i, p3r);
        const __m256d a3 = _mm256_add_pd(p1r, p3i), c3 = _mm256_sub_pd(p1i, p3r);
        __m256d r0, r1, r2, r3;
        TRANSPOSE4(r0, r1, r2, r3, a0, a1, a2, a3);
        _mm256_store_pd(re + g, r0);  _mm256_store_pd(re + g + 4, r1);  _mm256_store_pd(re + g + 8, r2);  _mm256_store_pd(re + g + 12, r3);
        TRANSPOSE4(r0, r1, r2, r3, c0, c1, c2, c3);
        _mm256_store_pd(im + g, r0);  _mm256_store_pd(im + g + 4, r1);  _mm256_store_pd(im + g + 8, r2);  _mm256_store_pd(im + g + 12, r3);
    }
    for (int32_t h = 4; h < M; h <<= 1) {
        const double *wr = t->wr + (M - 2 * h), *wi = t->wi + (M - 2 * h);
        for (int32_t b = 0; b < M; b += 2 * h)
            for (int32_t j = 0; j < h; j += 4) {
                double *pr = re + b + j, *pi_ = im + b + j;
                const __m256d xr = _mm256_load_pd(pr + h), xi = _mm256_load_pd(pi_ + h);
                const __m256d cr = _mm256_load_pd(wr + j), ci = _mm256_load_pd(wi + j);
                const __m256d vr = _mm256_fmadd_pd(xr, cr, _mm256_mul_pd(xi, ci));    /* x * conj(w) */
                const __m256d vi = _mm256_fmsub_pd(xi, cr, _mm256_mul_pd(xr, ci));
                const __m256d ur = _mm256_load_pd(pr), ui = _mm256_load_pd(pi_);
                _mm256_store_pd(pr, _mm256_add_pd(ur, vr));      _mm256_store_pd(pi_, _mm256_add_pd(ui, vi));
                _mm256_store_pd(pr + h, _mm256_sub_pd(ur, vr));  _mm256_store_pd(pi_ + h, _mm256_sub_pd(ui, vi));
            }
    }
}

/* int32 coefficients -> twisted complex points -> evaluation domain */
FFT4 static void fft4_from_i32(double *re, double *im, const int32_t *p, const Fft4Tab *t, double scale) {
    const int32_t M = t->N / 2;
    const __m256d sc = _mm256_set1_pd(scale);
    for (int32_t j = 0; j < M; j += 4) {
        const __m256d a = _mm256_mul_pd(_mm256_cvtepi32_pd(_mm_loadu_si128((const __m128i *)(p + j))), sc);
        const __m256d b = _mm256_mul_pd(_mm256_cvtepi32_pd(_mm_loadu_si128((const __m128i *)(p + j + M))), sc);
        const __m256d tr = _mm256_load_pd(t->tr + j), ti = _mm256_load_pd(t->ti + j);
        _mm256_store_pd(re + j, _mm256_fmsub_pd(a, tr, _mm256_mul_pd(b, ti)));
        _mm256_store_pd(im + j, _mm256_fmadd_pd(a, ti, _mm256_mul_pd(b, tr)));
    }
    fft4_fwd(re, im, t);
}

/* four doubles -> their nearest integers mod 2^32 (any magnitude below 2^62) */
FFT4 static inline __m128i round_mod32(__m256d x) {
    const __m256d q = _mm256_round_pd(_mm256_mul_pd(x, _mm256_set1_pd(1.0 / 4294967296.0)), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    const __m256d r = _mm256_fnmadd_pd(q, _mm256_set1_pd(4294967296.0), x);          /* |r| <= 2^31, exact */
    const __m256i bits = _mm256_castpd_si256(_mm256_add_pd(r, _mm256_set1_pd(6755399441055744.0)));   /* 1.5 * 2^52 */
    return _mm256_castsi256_si128(_mm256_permutevar8x32_epi32(bits, _mm256_setr_epi32(0, 2, 4, 6, 0, 2, 4, 6)));
}

/* acc[j] += round(value_j) mod 2^32 from an evaluation-domain sum */
FFT4 static void fft4_add_to_torus(uint32_t *acc, double *re, double *im, const Fft4Tab *t) {
    const int32_t M = t->N / 2;
    fft4_inv(re, im, t);
    for (int32_t j = 0; j < M; j += 4) {
        const __m256d zr = _mm256_load_pd(re + j), zi = _mm256_load_pd(im + j);
        const __m256d tr = _mm256_load_pd(t->tr + j), ti = _mm256_load_pd(t->ti + j);
        const __m256d a = _mm256_fmadd_pd(zr, tr, _mm256_mul_pd(zi, ti));         /* times conj(w^j) */
        const __m256d b = _mm256_fmsub_pd(zi, tr, _mm256_mul_pd(zr, ti));
        __m128i *lo = (__m128i *)(acc + j), *hi = (__m128i *)(acc + j + M);
        _mm_storeu_si128(lo, _mm_add_epi32(_mm_loadu_si128(lo), round_mod32(a)));
        _mm_storeu_si128(hi, _mm_add_epi32(_mm_loadu_si128(hi), round_mod32(b)));
    }
}

/* evaluation-domain image of the bootstrapping key, times 2/N, built on first use.  Laid out for ONE sequential stream per
 * blind-rotate step: [n][M/4][(k+1)l][k+1][re x4 | im x4] -- everything the step multiplies four points by, side by side.
 * (Rows 16 KB apart, as in a [row][re M | im M] image, all fall into one L1 set and evict each other.) */
static pthread_mutex_t g_img_mtx = PTHREAD_MUTEX_INITIALIZER;
FFT4 static const double *fft4_image(const OrcKeySet *ks_) {
    OrcKeySet *ks = (OrcKeySet *)ks_;
    pthread_mutex_lock(&g_img_mtx);
    if (!ks->bk_fft4) {
        const OrcParams *p = &ks->p;
        const int32_t N = p->N, M = N / 2, kpl = (p->k + 1) * p->l, kp1 = p->k + 1;
        const Fft4Tab *t = fft4_tab(N);
        double *img = (double *)alloc64(sizeof(double) * (size_t)p->n * kpl * kp1 * N);
        double *tmp = (double *)alloc64(sizeof(double) * FFT4_XSTRIDE(N));
        for (int32_t i = 0; i < p->n; ++i)
            for (int32_t q = 0; q < kpl; ++q)
                for (int32_t w = 0; w < kp1; ++w) {
                    fft4_from_i32(tmp, tmp + FFT4_IM(M), ks->bk + (((size_t)i * kpl + q) * kp1 + w) * N, t, 2.0 / N);
                    for (int32_t jv = 0; jv < M / 4; ++jv) {
                        double *dst = img + ((((size_t)i * (M / 4) + jv) * kpl + q) * kp1 + w) * 8;
                        memcpy(dst, tmp + 4 * jv, 4 * sizeof(double));
                        memcpy(dst + 4, tmp + FFT4_IM(M) + 4 * jv, 4 * sizeof(double));
                    }
                }
        free(tmp);
        ks->bk_fft4 = img;
    }
    pthread_mutex_unlock(&g_img_mtx);
    return ks->bk_fft4;
}

/* per-thread work arrays; complex arrays sit FFT4_XSTRIDE doubles apart */
typedef struct Fft4Scratch {
    int32_t *d;        /* (k+1) N rotated differences */
    double *x;         /* (k+1) l transformed digit polynomials [re M | im M] */
    double *s;         /* k+1 sums [re M | im M] */
} Fft4Scratch;

/* d = (X^a - 1) * acc, a in [1, 2N)  (tfhe: torusPolynomialMulByXaiMinusOne) */
FFT4 static void rotate_minus_one(int32_t *out, int32_t a, const int32_t *in, int32_t N) {
    const uint32_t *s = (const uint32_t *)in;  uint32_t *o = (uint32_t *)out;
    if (a < N) {
        for (int32_t i = 0; i < a; ++i) o[i] = 0u - s[i - a + N] - s[i];
        for (int32_t i = a; i < N; ++i) o[i] = s[i - a] - s[i];
    } else {
        const int32_t aa = a - N;
        for (int32_t i = 0; i < aa; ++i) o[i] = s[i - aa + N] - s[i];
        for (int32_t i = aa; i < N; ++i) o[i] = 0u - s[i - aa] - s[i];
    }
}

/* gadget digit q of a polynomial (tfhe: tGswTorus32PolynomialDecompH), twisted and transformed in one pass */
FFT4 static void digit_transform(double *re, double *im, const int32_t *d, int32_t q, const OrcParams *p, const Fft4Tab *t) {
    const int32_t M = t->N / 2, Bgbit = p->Bgbit;
    const int32_t halfBg = 1 << (Bgbit - 1);
    uint32_t offset = 0;
    for (int32_t j = 1; j <= p->l; ++j) offset += (uint32_t)halfBg << (32 - j * Bgbit);
    const __m128i voff = _mm_set1_epi32((int32_t)offset), vmask = _mm_set1_epi32((1 << Bgbit) - 1), vhalf = _mm_set1_epi32(halfBg);
    const __m128i sh = _mm_cvtsi32_si128(32 - (q + 1) * Bgbit);
    for (int32_t j = 0; j < M; j += 4) {
        const __m128i lo = _mm_sub_epi32(_mm_and_si128(_mm_srl_epi32(_mm_add_epi32(_mm_loadu_si128((const __m128i *)(d + j)), voff), sh), vmask), vhalf);
        const __m128i hi = _mm_sub_epi32(_mm_and_si128(_mm_srl_epi32(_mm_add_epi32(_mm_loadu_si128((const __m128i *)(d + j + M)), voff), sh), vmask), vhalf);
        const __m256d a = _mm256_cvtepi32_pd(lo), b = _mm256_cvtepi32_pd(hi);
        const __m256d tr = _mm256_load_pd(t->tr + j), ti = _mm256_load_pd(t->ti + j);
        _mm256_store_pd(re + j, _mm256_fmsub_pd(a, tr, _mm256_mul_pd(b, ti)));
        _mm256_store_pd(im + j, _mm256_fmadd_pd(a, ti, _mm256_mul_pd(b, tr)));
    }
    fft4_fwd(re, im, t);
}

/* one blind-rotate step (tfhe: tfhe_MuxRotate_FFT + tGswFFTExternMulToTLwe): acc += BK_i (.) ((X^a - 1) acc) */
FFT4 static void fft4_cmux(const OrcKeySet *ks, const double *img, const Fft4Tab *t, const Fft4Scratch *sc,
                           int32_t i, int32_t barai, Torus32 *acc) {
    const OrcParams *p = &ks->p;
    const int32_t N = p->N, M = N / 2, k = p->k, l = p->l, kpl = (k + 1) * l;
    for (int32_t u = 0; u <= k; ++u) {
        rotate_minus_one(sc->d + (size_t)u * N, barai, acc + (size_t)u * N, N);
        for (int32_t q = 0; q < l; ++q) {
            double *x = sc->x + (size_t)(u * l + q) * FFT4_XSTRIDE(N);
            digit_transform(x, x + FFT4_IM(M), sc->d + (size_t)u * N, q, p, t);
        }
    }
    if (k == 1) {            /* every built-in set: both output polynomials in one pass over the key rows */
        const double *b = img + (size_t)i * (M / 4) * kpl * 16;
        double *s0 = sc->s, *s1 = sc->s + FFT4_XSTRIDE(N);
        for (int32_t j = 0; j < M; j += 4) {
            /* eight independent accumulators: with four, each carries 2 (k+1) l dependent FMAs per group of four points and the
             * loop runs at the FMA latency instead of its throughput */
            __m256d ar0 = _mm256_setzero_pd(), ai0 = ar0, ar1 = ar0, ai1 = ar0, nr0 = ar0, bi0_ = ar0, nr1 = ar0, bi1_ = ar0;
            for (int32_t q = 0; q < kpl; ++q, b += 16) {
                const double *xr = sc->x + (size_t)q * FFT4_XSTRIDE(N) + j;
                const __m256d vxr = _mm256_load_pd(xr), vxi = _mm256_load_pd(xr + FFT4_IM(M));
                const __m256d br0 = _mm256_load_pd(b), bi0 = _mm256_load_pd(b + 4), br1 = _mm256_load_pd(b + 8), bi1 = _mm256_load_pd(b + 12);
                ar0 = _mm256_fmadd_pd(vxr, br0, ar0);  nr0 = _mm256_fmadd_pd(vxi, bi0, nr0);
                ai0 = _mm256_fmadd_pd(vxr, bi0, ai0);  bi0_ = _mm256_fmadd_pd(vxi, br0, bi0_);
                ar1 = _mm256_fmadd_pd(vxr, br1, ar1);  nr1 = _mm256_fmadd_pd(vxi, bi1, nr1);
                ai1 = _mm256_fmadd_pd(vxr, bi1, ai1);  bi1_ = _mm256_fmadd_pd(vxi, br1, bi1_);
            }
            _mm256_store_pd(s0 + j, _mm256_sub_pd(ar0, nr0));  _mm256_store_pd(s0 + FFT4_IM(M) + j, _mm256_add_pd(ai0, bi0_));
            _mm256_store_pd(s1 + j, _mm256_sub_pd(ar1, nr1));  _mm256_store_pd(s1 + FFT4_IM(M) + j, _mm256_add_pd(ai1, bi1_));
        }
        fft4_add_to_torus((uint32_t *)acc, s0, s0 + FFT4_IM(M), t);
        fft4_add_to_torus((uint32_t *)(acc + N), s1, s1 + FFT4_IM(M), t);
        return;
    }
    for (int32_t w = 0; w <= k; ++w) {
        double *sr = sc->s, *si = sc->s + FFT4_IM(M);
        for (int32_t j = 0; j < M; j += 4) {
            __m256d ar = _mm256_setzero_pd(), ai = _mm256_setzero_pd();
            for (int32_t q = 0; q < kpl; ++q) {
                const double *b = img + ((((size_t)i * (M / 4) + j / 4) * kpl + q) * (k + 1) + w) * 8;
                const double *xr = sc->x + (size_t)q * FFT4_XSTRIDE(N) + j;
                const __m256d vbr = _mm256_load_pd(b), vbi = _mm256_load_pd(b + 4);
                const __m256d vxr = _mm256_load_pd(xr), vxi = _mm256_load_pd(xr + FFT4_IM(M));
                ar = _mm256_fmadd_pd(vxr, vbr, ar);  ar = _mm256_fnmadd_pd(vxi, vbi, ar);
                ai = _mm256_fmadd_pd(vxr, vbi, ai);  ai = _mm256_fmadd_pd(vxi, vbr, ai);
            }
            _mm256_store_pd(sr + j, ar);  _mm256_store_pd(si + j, ai);
        }
        fft4_add_to_torus((uint32_t *)(acc + (size_t)w * N), sr, si, t);
    }
}

static void fft4_scratch_alloc(Fft4Scratch *sc, const OrcParams *p) {
    const size_t N = (size_t)p->N, kp1 = (size_t)p->k + 1;
    sc->d = (int32_t *)alloc64(sizeof(int32_t) * kp1 * N);
    sc->x = (double *)alloc64(sizeof(double) * kp1 * (size_t)p->l * FFT4_XSTRIDE(N));
    sc->s = (double *)alloc64(sizeof(double) * kp1 * FFT4_XSTRIDE(N));
}
static void fft4_scratch_free(Fft4Scratch *sc) { free(sc->d); free(sc->x); free(sc->s); }

/* all n steps on an accumulator the caller initialised (tfhe: tfhe_blindRotate_FFT) */
void orc_fft4_blind_rotate_steps(const OrcKeySet *ks, const int32_t *bara, Torus32 *acc) {
    if (!orc_fft4_available()) abort();      /* callers check orc_fft4_available() first */
    const Fft4Tab *t = fft4_tab(ks->p.N);
    const double *img = fft4_image(ks);
    Fft4Scratch sc;
    fft4_scratch_alloc(&sc, &ks->p);
    for (int32_t i = 0; i < ks->p.n; ++i) {
        if (bara[i] == 0) continue;
        fft4_cmux(ks, img, t, &sc, i, bara[i], acc);
    }
    fft4_scratch_free(&sc);
}

/* res = ip * tp mod (X^N+1), rounded, mod 2^32 -- for the tests that bound this evaluator's error */
FFT4 static void fft4_negacyclic(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N) {
    const Fft4Tab *t = fft4_tab(N);
    const int32_t M = N / 2;
    const size_t im = FFT4_IM(M);
    double *buf = (double *)alloc64(sizeof(double) * 3 * FFT4_XSTRIDE(N));
    double *a = buf, *b = buf + FFT4_XSTRIDE(N), *s = buf + 2 * FFT4_XSTRIDE(N);
    fft4_from_i32(a, a + im, ip, t, 1.0);
    fft4_from_i32(b, b + im, tp, t, 2.0 / N);
    for (int32_t j = 0; j < M; ++j) {
        s[j] = a[j] * b[j] - a[j + im] * b[j + im];
        s[j + im] = a[j] * b[j + im] + a[j + im] * b[j];
    }
    memset(res, 0, sizeof(Torus32) * (size_t)N);
    fft4_add_to_torus((uint32_t *)res, s, s + im, t);
    free(buf);
}
void orc_fft4_negacyclic(Torus32 *res, const int32_t *ip, const Torus32 *tp, int32_t N) {
    if (!orc_fft4_available()) abort();
    fft4_negacyclic(res, ip, tp, N);
}
