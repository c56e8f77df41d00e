// check_against_upstream.cpp -- the upstream-side half of the parity kit.
//
// NOT COMPILED OR RUN IN THIS REPOSITORY'S ENVIRONMENT: it needs the tfhe/tfhe library (github.com/tfhe/tfhe,
// v1.1), which is absent from /root/reference and from the build image (the reference only links it:
// /root/reference/CMakeLists.txt:9-15).  It is written against upstream's PUBLIC structs and functions as recalled
// (INTEGRATION.md "Moving keys ..."); if a name differs in your checkout, the table in
// INTEGRATION.md says which upstream object each file maps to.  Nothing here guesses upstream's FILE formats: the
// kit is raw int32 words (make_kit.py).
//
// What it does, on a machine that has upstream installed:
//   1. loads the kit's keys into upstream's LweBootstrappingKey (TGSW samples + key-switching key),
//   2. evaluates every case of kit.json with upstream's EXACT bootstrap (tfhe_bootstrap / tfhe_bootstrap_woKS:
//      the non-FFT path, torus polynomials multiplied exactly), with upstream's own gate preludes,
//   3. compares the output words with expected.i32 -- which is what this repo's CPU oracle AND its MI355X kernels
//      produce (tests/test_gpu_kernels.py::test_parity_kit_words_are_what_the_gpu_computes).
// All cases equal  =>  the oracle's reading of TFHE is upstream's, word for word: "parity unpinned" is lifted.
// A mismatch in case 0 is localised by extracted.i32 (the sample before the key switch).
// It also runs the FFT flavour you linked and reports how far it sits from the exact words (low-order noise bits
// only: same decryption), which is the relation the repo claims between its exact path and upstream's default path.
//
//   g++ -O2 check_against_upstream.cpp -o check -ltfhe-nayuki-portable     (or any other flavour)
//   ./check /path/to/repo/tests/golden/parity_kit
#include <tfhe/tfhe.h>
#include <tfhe/tfhe_io.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static std::vector<int32_t> load(const std::string &dir, const char *name, size_t words) {
    std::vector<int32_t> v(words);
    FILE *f = std::fopen((dir + "/" + name).c_str(), "rb");
    if (!f || std::fread(v.data(), 4, words, f) != words) { std::fprintf(stderr, "cannot read %s (%zu words)\n", name, words); std::exit(2); }
    std::fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <kit directory>\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    // the tuple of kit.json
    const int32_t n = 4, N = 1024, k = 1, l = 3, Bgbit = 7, ks_t = 8, ks_basebit = 2, base = 1 << ks_basebit;
    const double ks_stdev = 3.0517578125e-05 /* 2^-15 */, bk_stdev = 2.98023223876953125e-08 /* 2^-25 */, max_stdev = 0.012467;
    const int32_t kpl = (k + 1) * l;

    LweParams *lwe_params = new_LweParams(n, ks_stdev, max_stdev);
    TLweParams *tlwe_params = new_TLweParams(N, k, bk_stdev, max_stdev);
    TGswParams *tgsw_params = new_TGswParams(l, Bgbit, tlwe_params);
    LweBootstrappingKey *bk = new_LweBootstrappingKey(ks_t, ks_basebit, lwe_params, tgsw_params);

    // ---- keys ----
    const std::vector<int32_t> bkw = load(dir, "bk.i32", (size_t)n * kpl * (k + 1) * N);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t row = 0; row < kpl; ++row)
            for (int32_t poly = 0; poly <= k; ++poly) {
                Torus32 *dst = bk->bk[i].all_sample[row].a[poly].coefsT;       // poly == k is the body (->b)
                const int32_t *src = &bkw[(((size_t)i * kpl + row) * (k + 1) + poly) * N];
                for (int32_t j = 0; j < N; ++j) dst[j] = src[j];
            }
    const std::vector<int32_t> ksw = load(dir, "ksk.i32", (size_t)k * N * ks_t * (base - 1) * (n + 1));
    for (int32_t i = 0; i < k * N; ++i)
        for (int32_t j = 0; j < ks_t; ++j) {
            LweSample *zero = &bk->ks->ks[i][j][0];
            for (int32_t q = 0; q < n; ++q) zero->a[q] = 0;
            zero->b = 0;
            for (int32_t v = 1; v < base; ++v) {
                const int32_t *src = &ksw[((((size_t)i * ks_t + j) * (base - 1)) + (v - 1)) * (n + 1)];
                LweSample *dst = &bk->ks->ks[i][j][v];
                for (int32_t q = 0; q < n; ++q) dst->a[q] = src[q];
                dst->b = src[n];
            }
        }
    // the secret key, only to show the decrypted bits beside the words
    const std::vector<int32_t> sk = load(dir, "lwe_key.i32", n);
    LweKey *lwe_key = new_LweKey(lwe_params);
    for (int32_t q = 0; q < n; ++q) lwe_key->key[q] = sk[q];

    // ---- inputs, expected outputs ----
    const std::vector<int32_t> inw = load(dir, "inputs.i32", (size_t)8 * (n + 1));
    LweSample *in = new_LweSample_array(8, lwe_params);
    for (int32_t c = 0; c < 8; ++c) {
        for (int32_t q = 0; q < n; ++q) in[c].a[q] = inw[(size_t)c * (n + 1) + q];
        in[c].b = inw[(size_t)c * (n + 1) + n];
    }
    struct Case { const char *gate; int a, b, c; };
    const Case cases[9] = {{"AND", 1, 2, -1}, {"AND", 0, 1, -1}, {"XOR", 1, 2, -1}, {"XOR", 0, 3, -1}, {"OR", 0, 3, -1},
                           {"XNOR", 1, 3, -1}, {"NAND", 4, 5, -1}, {"MUX", 1, 0, 2}, {"MUX", 3, 0, 2}};
    const std::vector<int32_t> want = load(dir, "expected.i32", (size_t)9 * (n + 1));
    const std::vector<int32_t> want_u = load(dir, "extracted.i32", (size_t)k * N + 1);

    const Torus32 MU = modSwitchToTorus32(1, 8);
    const LweParams *extract_params = &tlwe_params->extracted_lweparams;
    LweSample *temp = new_LweSample(lwe_params), *res = new_LweSample(lwe_params);
    LweSample *u1 = new_LweSample(extract_params), *u2 = new_LweSample(extract_params);
    LweBootstrappingKeyFFT *bkfft = new_LweBootstrappingKeyFFT(bk);
    LweSample *res_fft = new_LweSample(lwe_params);

    int bad = 0;
    for (int ci = 0; ci < 9; ++ci) {
        const Case &cs = cases[ci];
        const std::string g = cs.gate;
        if (g == "MUX") {
            // upstream bootsMUX with the exact bootstrap: u1 = BR(-1/8 + a + b), u2 = BR(-1/8 - a + c), KS(u1 + u2 + 1/8)
            lweNoiselessTrivial(temp, modSwitchToTorus32(-1, 8), lwe_params);
            lweAddTo(temp, &in[cs.a], lwe_params);
            lweAddTo(temp, &in[cs.b], lwe_params);
            tfhe_bootstrap_woKS(u1, bk, MU, temp);
            lweNoiselessTrivial(temp, modSwitchToTorus32(-1, 8), lwe_params);
            lweSubTo(temp, &in[cs.a], lwe_params);
            lweAddTo(temp, &in[cs.c], lwe_params);
            tfhe_bootstrap_woKS(u2, bk, MU, temp);
            lweNoiselessTrivial(temp, 0, lwe_params);      // (unused below; keeps temp defined)
            LweSample *sum = new_LweSample(extract_params);
            lweNoiselessTrivial(sum, MU, extract_params);
            lweAddTo(sum, u1, extract_params);
            lweAddTo(sum, u2, extract_params);
            lweKeySwitch(res, bk->ks, sum);
            delete_LweSample(sum);
            lweCopy(res_fft, res, lwe_params);             // (no separate FFT run for MUX)
        } else {
            // upstream boot-gates.cpp preludes
            if (g == "AND") { lweNoiselessTrivial(temp, modSwitchToTorus32(-1, 8), lwe_params); lweAddTo(temp, &in[cs.a], lwe_params); lweAddTo(temp, &in[cs.b], lwe_params); }
            else if (g == "OR") { lweNoiselessTrivial(temp, modSwitchToTorus32(1, 8), lwe_params); lweAddTo(temp, &in[cs.a], lwe_params); lweAddTo(temp, &in[cs.b], lwe_params); }
            else if (g == "NAND") { lweNoiselessTrivial(temp, modSwitchToTorus32(1, 8), lwe_params); lweSubTo(temp, &in[cs.a], lwe_params); lweSubTo(temp, &in[cs.b], lwe_params); }
            else if (g == "XOR") { lweNoiselessTrivial(temp, modSwitchToTorus32(1, 4), lwe_params); lweAddMulTo(temp, 2, &in[cs.a], lwe_params); lweAddMulTo(temp, 2, &in[cs.b], lwe_params); }
            else /* XNOR */ { lweNoiselessTrivial(temp, modSwitchToTorus32(-1, 4), lwe_params); lweSubMulTo(temp, 2, &in[cs.a], lwe_params); lweSubMulTo(temp, 2, &in[cs.b], lwe_params); }
            if (ci == 0) {
                tfhe_bootstrap_woKS(u1, bk, MU, temp);
                int du = 0;
                for (int32_t q = 0; q < k * N; ++q) du += u1->a[q] != want_u[q];
                du += u1->b != want_u[(size_t)k * N];
                std::printf("case 0 before the key switch: %d of %d words differ\n", du, k * N + 1);
            }
            tfhe_bootstrap(res, bk, MU, temp);              // EXACT path: blind rotate with exact products, extract, key switch
            tfhe_bootstrap_FFT(res_fft, bkfft, MU, temp);   // the flavour linked, for the distance report
        }
        int diff = 0;
        int64_t far = 0;
        for (int32_t q = 0; q <= n; ++q) {
            const int32_t got = q < n ? res->a[q] : res->b, gotf = q < n ? res_fft->a[q] : res_fft->b;
            diff += got != want[(size_t)ci * (n + 1) + q];
            const int64_t d = (int64_t)(int32_t)((uint32_t)got - (uint32_t)gotf);
            if ((d < 0 ? -d : d) > far) far = d < 0 ? -d : d;
        }
        const Torus32 phase = lwePhase(res, lwe_key);
        std::printf("case %d %-4s: %s (%d of %d words differ); decrypts to %d; FFT flavour within %lld units of 2^-32\n", ci, cs.gate,
                    diff ? "MISMATCH" : "equal", diff, n + 1, phase > 0 ? 1 : 0, (long long)far);
        bad += diff != 0;
    }
    if (bad) std::printf("PARITY KIT: %d case(s) differ from upstream's exact bootstrap\n", bad);
    else std::printf("PARITY KIT: all cases equal upstream's exact bootstrap\n");
    return bad ? 1 : 0;
}
