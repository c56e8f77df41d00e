// plain_tfhe.cpp -- TEST INFRASTRUCTURE: a plaintext-bit provider of the tfhe gate
// API (the 16 symbols of SURVEY.md 8b).  The "ciphertext" is the bit itself, kept
// in LweSample::b.  It counts calls and hashes the gate sequence (operation,
// destination, operands identified by allocation serial + index), so that two
// circuit libraries can be compared gate for gate without a GPU.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>

#include "tfhe/tfhe.h"

namespace {
struct ArrInfo { int64_t serial; int32_t count; };
std::map<uintptr_t, ArrInfo> g_arrays;
int64_t g_serial = 0;
uint64_t g_hash = 1469598103934665603ull;
int64_t g_counts[16];   // XOR AND OR XNOR MUX NOT COPY CONST alloc free uninit_reads NAND NOR ...
enum { C_XOR, C_AND, C_OR, C_XNOR, C_MUX, C_NOT, C_COPY, C_CONST, C_ALLOC, C_FREE, C_UNINIT, C_OTHER };

TFheGateBootstrappingParameterSet g_params;
LweParams g_lwe = {1, 0.0, 0.0};

void mix(uint64_t v) {
    for (int i = 0; i < 8; ++i) { g_hash ^= (v >> (8 * i)) & 0xff; g_hash *= 1099511628211ull; }
}
int64_t id_of(const LweSample *s) {
    auto it = g_arrays.upper_bound(reinterpret_cast<uintptr_t>(s));
    if (it == g_arrays.begin()) return -1;
    --it;
    const auto idx = (reinterpret_cast<uintptr_t>(s) - it->first) / sizeof(LweSample);
    if ((int64_t)idx >= it->second.count) return -1;
    return it->second.serial * 65536 + (int64_t)idx;
}
int32_t rd(const LweSample *s) {
    if (s->slot == -2) ++g_counts[C_UNINIT];
    return s->b & 1;
}
void wr(LweSample *s, int32_t v) { s->b = v & 1; s->slot = -1; }
void ev(int op, const LweSample *d, const LweSample *a, const LweSample *b, const LweSample *c) {
    mix((uint64_t)op); mix((uint64_t)id_of(d));
    if (a) mix((uint64_t)id_of(a));
    if (b) mix((uint64_t)id_of(b));
    if (c) mix((uint64_t)id_of(c));
}
}  // namespace

extern "C" {

void mock_reset(void) { std::memset(g_counts, 0, sizeof g_counts); g_hash = 1469598103934665603ull; g_serial = 0; }
void mock_counts(int64_t *out) { std::memcpy(out, g_counts, sizeof g_counts); }
uint64_t mock_trace_hash(void) { return g_hash; }
int64_t mock_bootstraps(void) {
    return g_counts[C_XOR] + g_counts[C_AND] + g_counts[C_OR] + g_counts[C_XNOR] + 2 * g_counts[C_MUX] + g_counts[C_OTHER];
}

TFheGateBootstrappingParameterSet *new_default_gate_bootstrapping_parameters(int32_t) {
    g_params.ks_t = 8; g_params.ks_basebit = 2; g_params.in_out_params = &g_lwe; g_params.tgsw_params = nullptr;
    return &g_params;
}
void delete_gate_bootstrapping_parameters(TFheGateBootstrappingParameterSet *) {}
TFheGateBootstrappingSecretKeySet *new_random_gate_bootstrapping_secret_keyset(const TFheGateBootstrappingParameterSet *p) {
    auto *k = static_cast<TFheGateBootstrappingSecretKeySet *>(std::calloc(1, sizeof(TFheGateBootstrappingSecretKeySet)));
    k->params = p; k->cloud.params = p;
    return k;
}
void delete_gate_bootstrapping_secret_keyset(TFheGateBootstrappingSecretKeySet *k) { std::free(k); }

LweSample *new_gate_bootstrapping_ciphertext_array(int32_t n, const TFheGateBootstrappingParameterSet *) {
    // two samples of slack: the reference's Function_g writes bitsize+1 samples into a bitsize-sample
    // array (Math.cpp:399-401); the slack keeps that inside the block, as upstream's separate
    // heap blocks usually do by luck, so that the reference's own main() can run to its end over this mock
    auto *p = static_cast<LweSample *>(std::calloc((size_t)(n > 0 ? n : 1) + 2, sizeof(LweSample)));
    for (int i = 0; i < n; ++i) p[i].slot = -2;   // never written
    g_arrays[reinterpret_cast<uintptr_t>(p)] = ArrInfo{g_serial++, n};
    ++g_counts[C_ALLOC];
    return p;
}
void delete_gate_bootstrapping_ciphertext_array(int32_t, LweSample *p) {
    g_arrays.erase(reinterpret_cast<uintptr_t>(p));
    std::free(p);
    ++g_counts[C_FREE];
}
void bootsSymEncrypt(LweSample *r, int32_t m, const TFheGateBootstrappingSecretKeySet *) { wr(r, m); }
int32_t bootsSymDecrypt(const LweSample *s, const TFheGateBootstrappingSecretKeySet *) { return s->b & 1; }

void bootsCONSTANT(LweSample *r, int32_t v, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_CONST]; ev(C_CONST, r, 0, 0, 0); mix((uint64_t)v); wr(r, v); }
void bootsNOT(LweSample *r, const LweSample *a, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_NOT]; ev(C_NOT, r, a, 0, 0); wr(r, 1 - rd(a)); }
void bootsCOPY(LweSample *r, const LweSample *a, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_COPY]; ev(C_COPY, r, a, 0, 0); wr(r, rd(a)); }
void bootsAND(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_AND]; ev(C_AND, r, a, b, 0); wr(r, rd(a) & rd(b)); }
void bootsOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_OR]; ev(C_OR, r, a, b, 0); wr(r, rd(a) | rd(b)); }
void bootsXOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_XOR]; ev(C_XOR, r, a, b, 0); wr(r, rd(a) ^ rd(b)); }
void bootsXNOR(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *) { ++g_counts[C_XNOR]; ev(C_XNOR, r, a, b, 0); wr(r, 1 - (rd(a) ^ rd(b))); }
// the other upstream two-input gates (used by the optimised circuits, not by the reference)
#define MOCK_OTHER(NAME, CODE, EXPR)                                                                                   \
    void NAME(LweSample *r, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *) {        \
        ++g_counts[C_OTHER]; ev(CODE, r, a, b, 0);                                                                     \
        const int32_t x = rd(a), y = rd(b);                                                                            \
        wr(r, EXPR);                                                                                                   \
    }
MOCK_OTHER(bootsNAND, 20, 1 - (x & y))
MOCK_OTHER(bootsNOR, 21, 1 - (x | y))
MOCK_OTHER(bootsANDNY, 22, (1 - x) & y)
MOCK_OTHER(bootsANDYN, 23, x & (1 - y))
MOCK_OTHER(bootsORNY, 24, (1 - x) | y)
MOCK_OTHER(bootsORYN, 25, x | (1 - y))
#undef MOCK_OTHER
void bootsMUX(LweSample *r, const LweSample *a, const LweSample *b, const LweSample *c, const TFheGateBootstrappingCloudKeySet *) {
    ++g_counts[C_MUX]; ev(C_MUX, r, a, b, c);
    const int32_t va = rd(a), vb = rd(b), vc = rd(c);
    wr(r, va ? vb : vc);
}

// raw words of the plaintext "ciphertext": n = 1 mask word (always 0) then the bit
int32_t tfhe_hip_sample_words(const TFheGateBootstrappingParameterSet *) { return 2; }
int tfhe_hip_export_samples(const LweSample *s, int32_t count, const TFheGateBootstrappingParameterSet *, Torus32 *out) {
    for (int32_t i = 0; i < count; ++i) { out[2 * i] = 0; out[2 * i + 1] = rd(&s[i]); }
    return 0;
}
int tfhe_hip_import_samples(LweSample *s, int32_t count, const TFheGateBootstrappingParameterSet *, const Torus32 *in) {
    for (int32_t i = 0; i < count; ++i) wr(&s[i], in[2 * i + 1]);
    return 0;
}

}  // extern "C"
