// io.cpp -- key / ciphertext files behind the tfhe_io.h entry points.
// Container: {"TFHP", u32 version, u32 kind, u32 reserved, u64 payload bytes} (24 bytes) + payload,
// little endian.
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "engine.hpp"
#include "../../include/tfhe/tfhe.h"
#include "../../include/tfhe/tfhe_io.h"

using namespace tfhe_hip;

namespace {

enum : uint32_t { KIND_PARAMS = 1, KIND_CLOUD = 2, KIND_SECRET = 3, KIND_SAMPLE = 4 };
constexpr uint32_t VERSION = 1;

struct Header { char magic[4]; uint32_t version, kind, reserved; uint64_t bytes; };
static_assert(sizeof(Header) == 24, "file header layout");

void put(FILE *F, const void *p, size_t n) {
    if (n && std::fwrite(p, 1, n, F) != n) api_fail("tfhe_io: short write");
}
void get(FILE *F, void *p, size_t n) {
    if (n && std::fread(p, 1, n, F) != n) api_fail("tfhe_io: short read (truncated or foreign file)");
}
void put_header(FILE *F, uint32_t kind, uint64_t bytes) {
    Header h{{'T', 'F', 'H', 'P'}, VERSION, kind, 0, bytes};
    put(F, &h, sizeof h);
}
uint64_t get_header(FILE *F, uint32_t kind) {
    Header h;
    get(F, &h, sizeof h);
    if (std::memcmp(h.magic, "TFHP", 4) != 0) api_fail("tfhe_io: not a libtfhe-hip file (upstream tfhe files are not interchangeable)");
    if (h.version != VERSION) api_fail("tfhe_io: unsupported file version " + std::to_string(h.version));
    if (h.kind != kind) api_fail("tfhe_io: file holds object kind " + std::to_string(h.kind) + ", expected " + std::to_string(kind));
    return h.bytes;
}

struct ParamsRecord { int32_t n, N, k, l, Bgbit, ks_t, ks_basebit, pad; double ks_stdev, bk_stdev, max_stdev; };
ParamsRecord to_record(const Params &p) { return {p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, 0, p.ks_stdev, p.bk_stdev, p.max_stdev}; }
Params from_record(const ParamsRecord &r) { return Params{r.n, r.N, r.k, r.l, r.Bgbit, r.ks_t, r.ks_basebit, r.ks_stdev, r.bk_stdev, r.max_stdev}; }

void put_params(FILE *F, const Params &p) { const ParamsRecord r = to_record(p); put(F, &r, sizeof r); }
Params get_params(FILE *F) {
    ParamsRecord r;
    get(F, &r, sizeof r);
    const Params p = from_record(r);
    if (p.n <= 0 || p.n > 4096 || p.N <= 0 || p.N > 65536 || (p.N & (p.N - 1)) || p.k < 1 || p.k > 4 || p.l < 1 || p.Bgbit < 1 ||
        p.l * p.Bgbit > 32 || p.ks_t < 1 || p.ks_basebit < 1 || p.ks_t * p.ks_basebit > 31)
        api_fail("tfhe_io: corrupt parameter record");
    // the shapes the engine can evaluate (the predicate of Engine::upload_key): a file describing anything else
    // is refused here, with the reason, instead of loading and failing at the first gate
    if (const char *why = unsupported_reason(p)) api_fail(std::string("tfhe_io: unsupported parameter set in file: ") + why);
    return p;
}

template <typename T>
void put_vec(FILE *F, const std::vector<T> &v) { put(F, v.data(), v.size() * sizeof(T)); }
// Reads in bounded chunks, growing the vector as the bytes arrive: a short or hostile file whose header promises
// gigabytes fails at its real end with at most one chunk allocated beyond its content.
template <typename T>
void get_vec(FILE *F, std::vector<T> &v, size_t count) {
    constexpr size_t CHUNK = ((size_t)16 << 20) / sizeof(T);
    v.clear();
    for (size_t done = 0; done < count;) {
        const size_t step = count - done < CHUNK ? count - done : CHUNK;
        v.resize(done + step);
        get(F, v.data() + done, step * sizeof(T));
        done += step;
    }
}

// keysets created by the loaders own their parameter bundle (and, for cloud keysets, themselves)
std::mutex g_mtx;
std::set<const void *> g_owned_cloud, g_owned_secret;

void put_cloud_payload(FILE *F, const TfheHipCloudKey &ck) { put_params(F, ck.p); put_vec(F, ck.bk); put_vec(F, ck.ksk); }
uint64_t cloud_bytes(const Params &p) { return sizeof(ParamsRecord) + (p.bk_words() + p.ksk_words()) * sizeof(Torus32); }
// `declared` = the header's payload size (a secret-keyset file carries the two secret keys after the cloud payload); the sizes the
// file's own parameter record implies must agree with it before anything is allocated
TfheHipCloudKey *get_cloud_payload(FILE *F, uint64_t declared, bool with_secret_keys) {
    std::unique_ptr<TfheHipCloudKey> ck(new TfheHipCloudKey());
    ck->p = get_params(F);
    const uint64_t secret_bytes = with_secret_keys ? (uint64_t)(ck->p.n + ck->p.k * ck->p.N) * sizeof(int32_t) : 0;
    if (declared != cloud_bytes(ck->p) + secret_bytes)
        api_fail("tfhe_io: payload size does not match the file's parameter record (corrupt file)");
    get_vec(F, ck->bk, ck->p.bk_words());
    get_vec(F, ck->ksk, ck->p.ksk_words());
    return ck.release();
}

}  // namespace

namespace tfhe_hip {
// called by the delete_* functions of shim.cpp: true if the keyset came from a loader
bool io_forget_owned_cloud(const void *ks) { std::lock_guard<std::mutex> g(g_mtx); return g_owned_cloud.erase(ks) != 0; }
bool io_forget_owned_secret(const void *ks) { std::lock_guard<std::mutex> g(g_mtx); return g_owned_secret.erase(ks) != 0; }
}  // namespace tfhe_hip

// A malformed, truncated or foreign (e.g. upstream tfhe) file is reported through
// tfhe_hip_last_error(): loaders return nullptr, the others leave their arguments untouched.
template <typename F>
auto io_guard(F &&body) -> decltype(body()) {
    try { return body(); }
    catch (const ApiError &e) { set_error(e.msg); return decltype(body())(); }
    catch (const std::bad_alloc &) { set_error("tfhe_io: out of host memory"); return decltype(body())(); }
    catch (const std::exception &e) { set_error(std::string("tfhe_io: ") + e.what()); return decltype(body())(); }
}
struct Done {};   // "void" for io_guard

extern "C" {

void export_tfheGateBootstrappingParameterSet_toFile(FILE *F, const TFheGateBootstrappingParameterSet *params) {
    io_guard([&] {
        put_header(F, KIND_PARAMS, sizeof(ParamsRecord));
        put_params(F, params_of(params));
        return Done{};
    });
}
TFheGateBootstrappingParameterSet *new_tfheGateBootstrappingParameterSet_fromFile(FILE *F) {
    return io_guard([&]() -> TFheGateBootstrappingParameterSet * {
        if (get_header(F, KIND_PARAMS) != sizeof(ParamsRecord)) api_fail("tfhe_io: corrupt parameter file");
        return &make_param_bundle(get_params(F))->set;
    });
}

void export_tfheGateBootstrappingCloudKeySet_toFile(FILE *F, const TFheGateBootstrappingCloudKeySet *keyset) {
    io_guard([&] {
        put_header(F, KIND_CLOUD, cloud_bytes(keyset->bk->p));
        put_cloud_payload(F, *keyset->bk);
        return Done{};
    });
}
static TFheGateBootstrappingCloudKeySet *load_cloud(FILE *F) {
    const uint64_t declared = get_header(F, KIND_CLOUD);
    TfheHipCloudKey *ck = get_cloud_payload(F, declared, false);
    auto *ks = new TFheGateBootstrappingCloudKeySet();
    ks->params = &make_param_bundle(ck->p)->set;
    ks->bk = ck;
    ks->bkFFT = ck;
    std::lock_guard<std::mutex> g(g_mtx);
    g_owned_cloud.insert(ks);
    return ks;
}
TFheGateBootstrappingCloudKeySet *new_tfheGateBootstrappingCloudKeySet_fromFile(FILE *F) {
    return io_guard([&] { return load_cloud(F); });
}

void export_tfheGateBootstrappingSecretKeySet_toFile(FILE *F, const TFheGateBootstrappingSecretKeySet *keyset) {
    io_guard([&] {
        const TfheHipSecretKey &sk = *keyset->lwe_key;
        put_header(F, KIND_SECRET, cloud_bytes(sk.p) + (sk.lwe_key.size() + sk.tlwe_key.size()) * sizeof(int32_t));
        put_cloud_payload(F, *keyset->cloud.bk);
        put_vec(F, sk.lwe_key);
        put_vec(F, sk.tlwe_key);
        return Done{};
    });
}
static TFheGateBootstrappingSecretKeySet *load_secret(FILE *F) {
    const uint64_t declared = get_header(F, KIND_SECRET);
    std::unique_ptr<TfheHipCloudKey> ck(get_cloud_payload(F, declared, true));
    std::unique_ptr<TfheHipSecretKey> sk(new TfheHipSecretKey());
    sk->p = ck->p;
    get_vec(F, sk->lwe_key, (size_t)ck->p.n);
    get_vec(F, sk->tlwe_key, (size_t)ck->p.k * ck->p.N);
    auto *ks = new TFheGateBootstrappingSecretKeySet();
    ks->params = &make_param_bundle(ck->p)->set;
    ks->lwe_key = sk.get();
    ks->tgsw_key = sk.release();
    ks->cloud.params = ks->params;
    ks->cloud.bk = ck.get();
    ks->cloud.bkFFT = ck.release();
    std::lock_guard<std::mutex> g(g_mtx);
    g_owned_secret.insert(ks);
    return ks;
}
TFheGateBootstrappingSecretKeySet *new_tfheGateBootstrappingSecretKeySet_fromFile(FILE *F) {
    return io_guard([&] { return load_secret(F); });
}

void export_gate_bootstrapping_ciphertext_toFile(FILE *F, const LweSample *sample,
                                                 const TFheGateBootstrappingParameterSet *params) {
    io_guard([&] {
        const int32_t words = tfhe_hip_sample_words(params);
        std::vector<Torus32> w((size_t)words);
        if (tfhe_hip_export_samples(sample, 1, params, w.data()) != 0) return Done{};     // error already set
        put_header(F, KIND_SAMPLE, (uint64_t)words * sizeof(Torus32));
        put_vec(F, w);
        return Done{};
    });
}
void import_gate_bootstrapping_ciphertext_fromFile(FILE *F, LweSample *sample,
                                                   const TFheGateBootstrappingParameterSet *params) {
    io_guard([&] {
        const int32_t words = tfhe_hip_sample_words(params);
        if (get_header(F, KIND_SAMPLE) != (uint64_t)words * sizeof(Torus32)) api_fail("tfhe_io: ciphertext of a different LWE dimension");
        std::vector<Torus32> w;
        get_vec(F, w, (size_t)words);
        (void)tfhe_hip_import_samples(sample, 1, params, w.data());     // sets the error itself
        return Done{};
    });
}

}  // extern "C"
