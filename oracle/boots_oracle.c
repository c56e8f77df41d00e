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
static const Torus32 *value_words(const LweSample *s);
static int32_t const_node(int v);
static int g_rec_threads;
void orc_boots_export(const LweSample *s, int32_t count, Torus32 *out) {
    const size_t nw = (size_t)g_ks->p.n + 1;
    for (int32_t i = 0; i < count; ++i) {
        if (g_rec_threads > 0) memcpy(out + (size_t)i * nw, value_words(&s[i]), sizeof(Torus32) * nw);
        else to_words(&s[i], out + (size_t)i * nw);
    }
}

LweSample *new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet *params) {
    const int32_t n = params->in_out_params->n;
    char *mem = (char *)calloc(1, (size_t)nbelems * (sizeof(LweSample) + sizeof(Torus32) * n) + 16);
    LweSample *s = (LweSample *)mem;
    Torus32 *words = (Torus32 *)(mem + (size_t)nbelems * sizeof(LweSample));
    for (int32_t i = 0; i < nbelems; ++i) {
        s[i].a = words + (size_t)i * n;
        s[i].b = -(1 << 29);                 /* trivial encryption of 0 */
        s[i].slot = g_rec_threads > 0 ? const_node(0) : -1;   /* recording: every fresh sample is THE zero node */
    }
    return s;
}
void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample *samples) { (void)nbelems; free(samples); }

void bootsSymEncrypt(LweSample *r, int32_t m, const TFheGateBootstrappingSecretKeySet *key) {
    (void)key;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * (g_ks->p.n + 1));
    orc_encrypt_bit(g_ks, &g_rng, m, w);
    from_words(r, w);
    r->slot = -1;
    free(w);
}
int32_t bootsSymDecrypt(const LweSample *s, const TFheGateBootstrappingSecretKeySet *key) {
    (void)key;
    Torus32 *w = (Torus32 *)malloc(sizeof(Torus32) * (g_ks->p.n + 1));
    if (g_rec_threads > 0) memcpy(w, value_words(s), sizeof(Torus32) * (g_ks->p.n + 1));
    else to_words(s, w);
    const int32_t bit = orc_decrypt_bit(g_ks, w);
    free(w);
    return bit;
}


/* ------------------------------------------------------------------------------------------
 * Recording mode (orc_boots_set_recording(nthreads > 0)): the gate calls build a DAG instead of
 * evaluating; export / decrypt evaluate it level by level on `nthreads` host threads.  A gate's
 * output depends only on its operands' words, so the ciphertexts are the ones the immediate mode
 * produces (tests/test_circuits_cpu.py compares the two on a small circuit; the 2-slot golden
 * digest is reproduced by both).  This is what makes the full-size 128-slot Function_f digest
 * (215,544 blind rotations) affordable: ~0.11 s per rotation and core.
 * Identical gates on identical operands are recorded once (so two Function_f calls on the same
 * inputs with different bounds share the distance).
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <stdio.h>
#include <time.h>

typedef struct Node { int32_t op, in[3], level; Torus32 *w; } Node;   /* op: OrcGate, or -1 = input words */
static Node *g_nodes = NULL;
static int32_t g_nnodes = 0, g_capnodes = 0, g_done = 0;      /* nodes below g_done are evaluated */
static int g_rec_threads = 0;
static int32_t *g_htab = NULL; static uint32_t g_hmask = 0;   /* (op, operands) -> node */
static long long g_unique_rot = 0;

void orc_boots_set_recording(int nthreads) { g_rec_threads = nthreads; }
long long orc_boots_unique_rotations(void) { return g_unique_rot; }

static int32_t node_new(int op, int32_t a, int32_t b, int32_t c) {
    if (g_nnodes == g_capnodes) {
        g_capnodes = g_capnodes ? g_capnodes * 2 : 1 << 16;
        g_nodes = (Node *)realloc(g_nodes, sizeof(Node) * (size_t)g_capnodes);
    }
    Node *nd = &g_nodes[g_nnodes];
    nd->op = op; nd->in[0] = a; nd->in[1] = b; nd->in[2] = c; nd->level = 0; nd->w = NULL;
    return g_nnodes++;
}
static uint32_t hkey(int op, int32_t a, int32_t b, int32_t c) {
    uint64_t h = (uint64_t)(uint32_t)op * 0x9E3779B97F4A7C15ull;
    h = (h ^ (uint32_t)a) * 0xBF58476D1CE4E5B9ull; h = (h ^ (uint32_t)b) * 0x94D049BB133111EBull;
    h = (h ^ (uint32_t)c) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(h >> 29);
}
static int32_t node_gate(int op, int32_t a, int32_t b, int32_t c) {
    if (!g_htab || (uint32_t)g_nnodes * 2u > g_hmask) {          /* grow + rehash */
        const uint32_t size = g_htab ? (g_hmask + 1) * 2 : 1u << 18;
        int32_t *t = (int32_t *)malloc(sizeof(int32_t) * size);
        for (uint32_t i = 0; i < size; ++i) t[i] = -1;
        for (int32_t i = 0; i < g_nnodes; ++i) {
            const Node *nd = &g_nodes[i];
            if (nd->op < 0) continue;
            uint32_t h = hkey(nd->op, nd->in[0], nd->in[1], nd->in[2]) & (size - 1);
            while (t[h] >= 0) h = (h + 1) & (size - 1);
            t[h] = i;
        }
        free(g_htab); g_htab = t; g_hmask = size - 1;
    }
    uint32_t h = hkey(op, a, b, c) & g_hmask;
    for (; g_htab[h] >= 0; h = (h + 1) & g_hmask) {
        const Node *nd = &g_nodes[g_htab[h]];
        if (nd->op == op && nd->in[0] == a && nd->in[1] == b && nd->in[2] == c) return g_htab[h];
    }
    const int32_t id = node_new(op, a, b, c);
    g_htab[h] = id;
    if (op == ORC_MUX) g_unique_rot += 2; else if (op < ORC_NGATES2) g_unique_rot += 1;
    return id;
}
/* the node holding a sample's current value; a host-valued sample (fresh, encrypted, constant)
 * becomes an input node the first time it is used */
static int32_t node_of(const LweSample *s) {
    if (s->slot >= 0) return s->slot;
    const int32_t id = node_new(-1, -1, -1, -1);
    g_nodes[id].w = (Torus32 *)malloc(sizeof(Torus32) * ((size_t)g_ks->p.n + 1));
    memcpy(g_nodes[id].w, s->a, sizeof(Torus32) * g_ks->p.n); g_nodes[id].w[g_ks->p.n] = s->b;
    ((LweSample *)s)->slot = id;
    return id;
}
static int32_t g_const_node[2] = {-1, -1};
static int32_t const_node(int v) {
    if (g_const_node[v] < 0) {
        const int32_t id = node_new(-1, -1, -1, -1);
        g_nodes[id].w = (Torus32 *)calloc((size_t)g_ks->p.n + 1, sizeof(Torus32));
        g_nodes[id].w[g_ks->p.n] = v ? (1 << 29) : -(1 << 29);
        g_const_node[v] = id;
    }
    return g_const_node[v];
}

typedef struct LevelJob { const int32_t *ids; int32_t count; volatile int32_t next; } LevelJob;
static void eval_node(Node *nd) {
    const size_t nw = (size_t)g_ks->p.n + 1;
    nd->w = (Torus32 *)malloc(sizeof(Torus32) * nw);
    if (nd->op == ORC_MUX) orc_mux(g_ks, nd->w, g_nodes[nd->in[0]].w, g_nodes[nd->in[1]].w, g_nodes[nd->in[2]].w, 2);
    else if (nd->op == ORC_NOT) orc_not(&g_ks->p, nd->w, g_nodes[nd->in[0]].w);
    else orc_gate2(g_ks, nd->op, nd->w, g_nodes[nd->in[0]].w, g_nodes[nd->in[1]].w, 2);
}
static void *level_worker(void *arg) {
    LevelJob *j = (LevelJob *)arg;
    for (;;) {
        const int32_t i = __sync_fetch_and_add(&j->next, 1);
        if (i >= j->count) return NULL;
        eval_node(&g_nodes[j->ids[i]]);
    }
}
static int cmp_level(const void *a, const void *b) {
    const Node *x = &g_nodes[*(const int32_t *)a], *y = &g_nodes[*(const int32_t *)b];
    if (x->level != y->level) return x->level < y->level ? -1 : 1;
    return *(const int32_t *)a < *(const int32_t *)b ? -1 : 1;
}
/* evaluates every recorded node not evaluated yet: levels in ascending order, a level's
 * bootstrapped gates in parallel, then its NOTs (a NOT sits on its operand's level) */
void orc_boots_flush(void) {
    if (g_done == g_nnodes) return;
    const int32_t first = g_done, total = g_nnodes - first;
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)total);
    int32_t maxlevel = 0;
    for (int32_t i = first; i < g_nnodes; ++i) {
        Node *nd = &g_nodes[i];
        ids[i - first] = i;
        if (nd->op < 0 || nd->w) { nd->level = 0; continue; }
        int32_t lv = 0;
        const int nin = nd->op == ORC_MUX ? 3 : nd->op == ORC_NOT ? 1 : 2;
        for (int q = 0; q < nin; ++q) {
            const Node *src = &g_nodes[nd->in[q]];
            const int32_t sl = src->w && nd->in[q] < first ? 0 : src->level;
            if (sl > lv) lv = sl;
        }
        nd->level = nd->op == ORC_NOT ? lv : lv + 1;
        if (nd->level > maxlevel) maxlevel = nd->level;
    }
    qsort(ids, (size_t)total, sizeof(int32_t), cmp_level);
    const int nth = g_rec_threads > 0 ? g_rec_threads : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nth);
    int32_t *gates = (int32_t *)malloc(sizeof(int32_t) * (size_t)total);
    long long rot_done = 0;
    const time_t t0 = time(NULL); time_t last = t0;
    for (int32_t pos = 0; pos < total;) {
        const int32_t lv = g_nodes[ids[pos]].level;
        int32_t end = pos, ng = 0;
        while (end < total && g_nodes[ids[end]].level == lv) ++end;
        for (int32_t i = pos; i < end; ++i) {
            Node *nd = &g_nodes[ids[i]];
            if (nd->op >= 0 && nd->op != ORC_NOT && !nd->w) { gates[ng++] = ids[i]; rot_done += nd->op == ORC_MUX ? 2 : 1; }
        }
        if (ng) {
            LevelJob job = {gates, ng, 0};
            const int use = ng < nth ? ng : nth;
            for (int t = 0; t < use; ++t) pthread_create(&th[t], NULL, level_worker, &job);
            for (int t = 0; t < use; ++t) pthread_join(th[t], NULL);
        }
        for (int32_t i = pos; i < end; ++i) {              /* NOTs of this level, in recording order */
            Node *nd = &g_nodes[ids[i]];
            if (nd->op == ORC_NOT && !nd->w) eval_node(nd);
        }
        if (getenv("ORC_BOOTS_PROGRESS") && time(NULL) - last >= 60) {
            last = time(NULL);
            fprintf(stderr, "[oracle dag] level %d/%d, %lld rotations done, %ld s\n", lv, maxlevel, rot_done, (long)(last - t0));
        }
        pos = end;
    }
    free(th); free(gates); free(ids);
    g_done = g_nnodes;
}
static const Torus32 *value_words(const LweSample *s) {     /* recording mode: the sample's words */
    orc_boots_flush();
    return g_nodes[node_of(s)].w;
}

/* Constant folding (recording mode only; orc_boots_set_fold(1)): the rule the product's recorder applies under its opt-in
 * tuning "fold_constants" (peba1_amd/csrc/shim.cpp), restated -- a gate with a public constant operand (THE zero / one
 * node: bootsCONSTANT, a fresh sample, copies of them) is not bootstrapped: its result is a constant, the other operand or
 * its negation; a MUX with a constant data operand is a two-input gate.  TFHE itself bootstraps every gate: a folded
 * circuit's ciphertexts are NOT TFHE's, only its decryptions -- which is why both sides keep it off unless asked. */
static int g_fold = 0;
static long long g_folded = 0;
void orc_boots_set_fold(int on) { g_fold = on != 0; }
long long orc_boots_folded(void) { return g_folded; }
static int const_of(int32_t node) { return node >= 0 && node == g_const_node[0] ? 0 : node >= 0 && node == g_const_node[1] ? 1 : -1; }
static const unsigned char GATE_TT[ORC_NGATES2][4] = {      /* [gate][2 a + b], enum OrcGate order */
    {1, 1, 1, 0}, {0, 1, 1, 1}, {0, 0, 0, 1}, {1, 0, 0, 0}, {0, 1, 1, 0}, {1, 0, 0, 1},
    {0, 1, 0, 0}, {0, 0, 1, 0}, {1, 1, 0, 1}, {1, 0, 1, 1},
};
static int32_t not_node(int32_t ia) {
    if (g_fold && const_of(ia) >= 0) return const_node(1 - const_of(ia));
    return node_gate(ORC_NOT, ia, -1, -1);
}
static int32_t gate2_node(int g, int32_t ia, int32_t ib) {
    if (g_fold) {
        const int ka = const_of(ia), kb = const_of(ib);
        if (ka >= 0 || kb >= 0) {
            const unsigned char *tt = GATE_TT[g];
            const int f0 = ka >= 0 ? tt[2 * ka + (kb >= 0 ? kb : 0)] : tt[kb];
            const int f1 = ka >= 0 ? tt[2 * ka + (kb >= 0 ? kb : 1)] : tt[2 + kb];
            ++g_folded;
            if (f0 == f1) return const_node(f0);
            if (f0 == 0) return ka >= 0 ? ib : ia;
            return not_node(ka >= 0 ? ib : ia);
        }
    }
    ++g_gates;
    return node_gate(g, ia, ib, -1);
}

static void gate2(int g, LweSample *r, const LweSample *a, const LweSample *b) {
    if (g_rec_threads > 0) {
        const int32_t ia = node_of(a), ib = node_of(b);
        r->slot = gate2_node(g, ia, ib);
        return;
    }
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
    if (g_rec_threads > 0) {
        const int32_t ia = node_of(a), ib = node_of(b), ic = node_of(c);
        if (g_fold) {
            const int ka = const_of(ia), kb = const_of(ib), kc = const_of(ic);
            if (ka >= 0 || kb >= 0 || kc >= 0 || ib == ic) {
                ++g_folded;
                if (ka >= 0) r->slot = ka ? ib : ic;
                else if (ib == ic) r->slot = ib;
                else if (kb >= 0 && kc >= 0) r->slot = kb == 1 ? ia : not_node(ia);
                else if (kc >= 0) r->slot = gate2_node(kc == 0 ? ORC_AND : ORC_ORNY, ia, ib);
                else r->slot = gate2_node(kb == 0 ? ORC_ANDNY : ORC_OR, ia, ic);
                return;
            }
        }
        r->slot = node_gate(ORC_MUX, ia, ib, ic);
        g_gates += 2;
        return;
    }
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
    if (g_rec_threads > 0) { r->slot = not_node(node_of(a)); return; }
    for (int32_t i = 0; i < g_ks->p.n; ++i) r->a[i] = (Torus32)(0u - (uint32_t)a->a[i]);
    r->b = (Torus32)(0u - (uint32_t)a->b);
}
void bootsCOPY(LweSample *r, const LweSample *a, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    if (r == a) return;
    if (g_rec_threads > 0) { r->slot = node_of(a); return; }
    memmove(r->a, a->a, sizeof(Torus32) * g_ks->p.n);
    r->b = a->b;
}
void bootsCONSTANT(LweSample *r, int32_t v, const TFheGateBootstrappingCloudKeySet *k) {
    (void)k;
    if (g_rec_threads > 0) { r->slot = const_node(v ? 1 : 0); return; }
    memset(r->a, 0, sizeof(Torus32) * g_ks->p.n);
    r->b = v ? (1 << 29) : -(1 << 29);
}
