// driver.cpp -- TEST INFRASTRUCTURE.  Runs the known-answer scenarios of SURVEY.md 8c
// and prints one JSON object.  Built three ways:
//   -DUSE_REFERENCE : against /root/reference/src/Math.cpp (compiled in a temp dir,
//                     never copied) over the plaintext-bit provider (tests/mock), proving
//                     source compatibility of include/tfhe/*.h
//   default         : this repo's libpeba1-circuits over the same provider
//   -DUSE_REFERENCE -DREAL_PROVIDER : the reference's Math.cpp linked against libtfhe-hip.so
//                     itself (oracle/Makefile -> oracle/_ref/refdriver_hip): the reference's own
//                     object code on the GPU library, run by tests/test_gpu_circuits.py
// Equal output = same values (and, over the mock, same gate counts and gate sequence).
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#ifdef USE_REFERENCE
#include "Math.h"
#else
#include "peba1_circuits.h"
#endif

#ifdef REAL_PROVIDER
#include "tfhe_hip.h"
// no gate counters in the real library: blind rotations from its statistics, the rest zero
static void mock_reset(void) { tfhe_hip_flush(); tfhe_hip_reset_stats(); }
static void mock_counts(int64_t *out) { for (int i = 0; i < 16; ++i) out[i] = 0; }
static uint64_t mock_trace_hash(void) { return 0; }
static int64_t mock_bootstraps(void) {
    TfheHipStats st;
    tfhe_hip_flush();
    tfhe_hip_get_stats(&st);
    return (int64_t)st.blind_rotates;
}
#else
extern "C" {
void mock_reset(void);
void mock_counts(int64_t *out);
uint64_t mock_trace_hash(void);
int64_t mock_bootstraps(void);
}
#endif

static TFheGateBootstrappingParameterSet *params;
static TFheGateBootstrappingSecretKeySet *key;
static const TFheGateBootstrappingCloudKeySet *ck;

static LweSample *enc(uint64_t v, int bits) {
    LweSample *p = new_gate_bootstrapping_ciphertext_array(bits, params);
    for (int i = 0; i < bits; ++i) bootsSymEncrypt(&p[i], (v >> i) & 1, key);
    return p;
}
static uint64_t dec(LweSample *p, int bits) {
    uint64_t v = 0;
    for (int i = 0; i < bits; ++i) v |= (uint64_t)bootsSymDecrypt(&p[i], key) << i;
    return v;
}
static void report(const char *name, uint64_t value, bool last = false) {
    int64_t c[16];
    mock_counts(c);
    std::printf("  \"%s\": {\"value\": %llu, \"xor\": %lld, \"and\": %lld, \"or\": %lld, \"xnor\": %lld, \"mux\": %lld, "
                "\"not\": %lld, \"copy\": %lld, \"const\": %lld, \"allocs\": %lld, \"uninit_reads\": %lld, "
                "\"blind_rotates\": %lld, \"trace\": \"%016llx\"}%s\n",
                name, (unsigned long long)value, (long long)c[0], (long long)c[1], (long long)c[2], (long long)c[3],
                (long long)c[4], (long long)c[5], (long long)c[6], (long long)c[7], (long long)c[8], (long long)c[10],
                (long long)mock_bootstraps(), (unsigned long long)mock_trace_hash(), last ? "" : ",");
}

#ifdef USE_REFERENCE
#define ADDN(r, a, b, c, n) bootsADDNbit(r, a, b, c, n, ck)
#define SUBN(r, a, b, n) bootsSUBNbit(r, a, b, n, ck)
#define MULT(r, a, b, n) bootsMultiply(r, a, b, n, ck)
#define TWOSC(r, a, n) bootsTwoSComplement(r, a, n, ck)
#define ABSV(r, a, n) bootsABS(r, a, n, ck)
#define SHL(r, a, n, s) bootsShiftLeft(r, a, n, s, ck)
#define SHR(r, a, n, s) bootsShiftRight(r, a, n, s, ck)
#define SHLNR(a, n, s) bootsShiftLeftNR(a, n, s, ck)
#define EUCLID(r, a, b, n) HE_EuclideanDistance(r, a, b, n, ck)
#define FUNC_F(r, a, b, bound, n) Function_f(r, a, b, bound, n, ck)
#else
#define ADDN(r, a, b, c, n) peba1_add_nbit(r, a, b, c, n, ck)
#define SUBN(r, a, b, n) peba1_sub_nbit(r, a, b, n, ck)
#define MULT(r, a, b, n) peba1_multiply(r, a, b, n, ck)
#define TWOSC(r, a, n) peba1_twos_complement(r, a, n, ck)
#define ABSV(r, a, n) peba1_abs(r, a, n, ck)
#define SHL(r, a, n, s) peba1_shift_left(r, a, n, s, ck)
#define SHR(r, a, n, s) peba1_shift_right(r, a, n, s, ck)
#define SHLNR(a, n, s) peba1_shift_left_inplace(a, n, s, ck)
#define EUCLID(r, a, b, n) peba1_euclidean_distance(r, a.data(), b.data(), (int)a.size(), n, ck)
#define FUNC_F(r, a, b, bound, n) peba1_function_f(r, a.data(), b.data(), (int)a.size(), bound, n, ck)
#endif

int main() {
    params = new_default_gate_bootstrapping_parameters(128);
    key = new_random_gate_bootstrapping_secret_keyset(params);
    ck = &key->cloud;
    const int nslots = 128, bits = 8;
    std::vector<uint8_t> tmpl(nslots), genuine(nslots), impostor(nslots);
    for (int i = 0; i < nslots; ++i) {
        tmpl[i] = (uint8_t)((37 * i + 11) % 255);
        genuine[i] = (uint8_t)(tmpl[i] + 1);
        impostor[i] = (uint8_t)((91 * i + 5) % 256);
    }
    std::printf("{\n");
    {
        LweSample *a = enc(122, 8), *b = enc(204, 8), *r = new_gate_bootstrapping_ciphertext_array(8, params),
                  *c = new_gate_bootstrapping_ciphertext_array(1, params);
        mock_reset();
        ADDN(r, a, b, c, 8);
        report("addn8_122_204", dec(r, 8) | (dec(c, 1) << 8));
    }
    {
        LweSample *a = enc(5, 8), *r = new_gate_bootstrapping_ciphertext_array(9, params);
        mock_reset();
        TWOSC(r, a, 8);
        report("twosc8_5", dec(r, 8));
    }
    {
        LweSample *a = enc(122, 8), *b = enc(204, 8), *r = new_gate_bootstrapping_ciphertext_array(9, params);
        mock_reset();
        SUBN(r, a, b, 8);
        report("subn8_122_204", dec(r, 9));
    }
    {
        LweSample *a = enc(122, 9), *b = enc(204, 9), *r = new_gate_bootstrapping_ciphertext_array(24, params);
        mock_reset();
        MULT(r, a, b, 8);
        report("mult8_122_204", dec(r, 23));
    }
    // bootsABS (Math.cpp:97-119; main.cpp:374 calls it on 9-bit two's-complement numbers) and the
    // three shift helpers (Math.cpp:183-211)
    for (int which = 0; which < 2; ++which) {
        const uint64_t v = which == 0 ? ((uint64_t)(-82) & 0x1FF) : 77;      // -82 and +77 in 9 bits
        LweSample *a = enc(v, 9), *r = new_gate_bootstrapping_ciphertext_array(9, params);
        mock_reset();
        ABSV(r, a, 9);
        report(which == 0 ? "abs9_minus82" : "abs9_plus77", dec(r, 9));
    }
    {
        LweSample *a = enc(0xB5, 8), *r = new_gate_bootstrapping_ciphertext_array(8, params);
        mock_reset();
        SHL(r, a, 8, 3);
        report("shl8_b5_by3", dec(r, 8));
        mock_reset();
        SHR(r, a, 8, 3);
        report("shr8_b5_by3", dec(r, 8));
        mock_reset();
        SHLNR(a, 8, 2);
        report("shlnr8_b5_by2", dec(a, 8));
    }
    std::vector<LweSample *> et(nslots), eg(nslots), ei(nslots);
    for (int i = 0; i < nslots; ++i) { et[i] = enc(tmpl[i], 8); eg[i] = enc(genuine[i], 8); ei[i] = enc(impostor[i], 8); }
    for (int which = 0; which < 2; ++which) {
        LweSample *r = new_gate_bootstrapping_ciphertext_array(24, params);
        for (int i = 0; i < 24; ++i) bootsCONSTANT(&r[i], 0, ck);     // main.cpp:498-500 zeroes it first
        mock_reset();
        if (which == 0) { EUCLID(r, eg, et, bits); } else { EUCLID(r, ei, et, bits); }
        report(which == 0 ? "euclid128_genuine" : "euclid128_impostor", dec(r, 24));
    }
    const uint64_t bounds[2] = {0, 256};
    for (int bi = 0; bi < 2; ++bi)
        for (int which = 0; which < 2; ++which) {
            LweSample *bound = enc(bounds[bi], 24), *rb = new_gate_bootstrapping_ciphertext_array(24, params);
            mock_reset();
            if (which == 0) { FUNC_F(rb, eg, et, bound, bits); } else { FUNC_F(rb, ei, et, bound, bits); }
            const std::string name = std::string("function_f_") + (which == 0 ? "genuine" : "impostor") + "_bound" +
                                     std::to_string(bounds[bi]);
            report(name.c_str(), dec(rb, 24), bi == 1 && which == 1);
        }
    std::printf("}\n");
    return 0;
}
