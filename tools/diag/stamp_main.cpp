// diagnostic harness: one latency-kernel launch of G gates, print per-phase cycle sums
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tfhe/tfhe.h"
#include "tfhe_hip.h"
namespace tfhe_hip { void read_stamps(unsigned long long *out, bool reset); }
int main(int argc, char **argv) {
    int G = argc > 1 ? atoi(argv[1]) : 64;
    tfhe_hip_set_deferred(0);                      // the launches below must run when they are issued
    auto *params = new_default_gate_bootstrapping_parameters(128);
    auto *key = tfhe_hip_new_secret_keyset_seeded(params, 0x5EBA2);
    LweSample *a = new_gate_bootstrapping_ciphertext_array(G, params), *b = new_gate_bootstrapping_ciphertext_array(G, params),
              *r = new_gate_bootstrapping_ciphertext_array(G, params);
    for (int i = 0; i < G; ++i) { bootsSymEncrypt(&a[i], i & 1, key); bootsSymEncrypt(&b[i], (i >> 1) & 1, key); }
    tfhe_hip_gate_batch(TFHE_HIP_AND, r, a, b, G, &key->cloud);   // warm
    unsigned long long st[64] = {0};
    tfhe_hip::read_stamps(st, true);
    tfhe_hip_gate_batch(TFHE_HIP_AND, r, a, b, G, &key->cloud);
    tfhe_hip::read_stamps(st, false);
    const char *names[8] = {"loop/skip", "D+3fwd+MAC", "redc+X1 write", "barrier1", "X1 read+inverse+canon", "X2+CRT (incl barrier2)", "barrier3", "-"};
    // the 8-wave form (launches of at most one workgroup per CU) stamps 8 waves and 8 phases
    const bool eight = st[4 * 8] != 0;
    const char *names8[8] = {"loop / last barrier", "D + rows + MAC", "redc + rows to LDS", "barrier A", "sum rows + half inverse", "barrier B",
                             "finish (stage 0, CRT, acc)", "barrier C"};
    if (eight) {
        for (int w = 0; w < 8; ++w) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; ++k) tot += st[w * 8 + k];
            printf("wave %d (q=%d,u=%d,%s): per step %.0f cycles\n", w, w & 1, (w >> 1) & 1, w < 4 ? "A" : "B", (double)tot / G / 630);
            for (int k = 0; k < 8; ++k) printf("   %-28s %8.0f cyc/step  %5.1f%%\n", names8[k], (double)st[w * 8 + k] / G / 630, 100.0 * st[w * 8 + k] / tot);
        }
        return 0;
    }
    for (int w = 0; w < 4; ++w) {
        unsigned long long tot = 0;
        for (int k = 0; k < 8; ++k) tot += st[w * 8 + k];
        printf("wave %d (q=%d,u=%d): total %.0f cyc/gate, per step %.0f\n", w, w & 1, w >> 1, (double)tot / G, (double)tot / G / 630);
        for (int k = 0; k < 7; ++k) printf("   %-28s %8.0f cyc/step  %5.1f%%\n", names[k], (double)st[w * 8 + k] / G / 630, 100.0 * st[w * 8 + k] / tot);
    }
    return 0;
}
