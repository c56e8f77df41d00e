// dist.cpp -- libpeba1-dist: the multi-GPU forms of the match for a C-ABI host (include/peba1_dist.h).
// Host logic only: slot partition, the one exchange per match, rank 0's combine.  Every gate runs in the gate
// provider (libtfhe-hip); the circuits are libpeba1-circuits'.  RCCL is opened with dlopen on first use.
#include "../../include/peba1_dist.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfhe_hip.h"

// entry points only libtfhe-hip has (a host-transport run over another provider never calls them)
extern "C" {
__attribute__((weak)) int tfhe_hip_export_samples_device_async(const LweSample *, int32_t, const TFheGateBootstrappingParameterSet *, void *);
__attribute__((weak)) int tfhe_hip_import_samples_device_async(LweSample *, int32_t, const TFheGateBootstrappingParameterSet *, const void *);
__attribute__((weak)) void *tfhe_hip_stream(void);
__attribute__((weak)) const char *tfhe_hip_last_error(void);
}

namespace {

constexpr int PARTIAL_BITS = 24;          // samples of a partial sum of squares (Math.cpp:342: max_bitsize)

thread_local std::string g_error;
int fail(const std::string &msg) { g_error = msg; return -1; }
std::string provider_error() { return tfhe_hip_last_error ? std::string(tfhe_hip_last_error()) : std::string(); }

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return false;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        Gather = reinterpret_cast<decltype(Gather)>(dlsym(handle, "ncclGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        return GetUniqueId && CommInitRank && CommDestroy && Gather && GetErrorString;
    }
};
Rccl g_rccl;

}  // namespace

struct Peba1Comm {
    int world = 1, rank = 0;
    bool rccl = false, own = false;
    ncclComm_t nccl = nullptr;
    peba1_gather_fn gather = nullptr;
    void *ctx = nullptr;
    // grow-only device buffers of the RCCL path: they must outlive the stream operations that use them
    int32_t *send = nullptr, *recv = nullptr;
    size_t send_words = 0, recv_words = 0;
};

namespace {

bool device_buffer(int32_t *&buf, size_t &have, size_t want) {
    if (have >= want) return true;
    // the stream may still be using the old buffer
    if (buf) { (void)hipStreamSynchronize(static_cast<hipStream_t>(tfhe_hip_stream())); (void)hipFree(buf); buf = nullptr; have = 0; }
    if (hipMalloc(reinterpret_cast<void **>(&buf), want * sizeof(int32_t)) != hipSuccess) return false;
    have = want;
    return true;
}

// `count` samples of every rank -> rank 0, rank-major.  RCCL: device buffers, everything on the provider's stream;
// host transport: through the callback.
int gather_samples(Peba1Comm *c, LweSample *all, const LweSample *mine, int count, const TFheGateBootstrappingParameterSet *params) {
    const size_t words = (size_t)tfhe_hip_sample_words(params) * (size_t)count;
    if (c->rccl) {
        if (!tfhe_hip_export_samples_device_async || !tfhe_hip_import_samples_device_async || !tfhe_hip_stream)
            return fail("the RCCL transport needs libtfhe-hip as the gate provider");
        if (!device_buffer(c->send, c->send_words, words)) return fail("hipMalloc of the send buffer failed");
        if (c->rank == 0 && !device_buffer(c->recv, c->recv_words, words * (size_t)c->world)) return fail("hipMalloc of the receive buffer failed");
        hipStream_t stream = static_cast<hipStream_t>(tfhe_hip_stream());
        // runs the pending gates, then gathers the slots into the buffer -- enqueued, not waited for
        if (tfhe_hip_export_samples_device_async(mine, count, params, c->send) != 0) return fail("export of the samples: " + provider_error());
        const ncclResult_t r = g_rccl.Gather(c->send, c->rank == 0 ? c->recv : nullptr, words, ncclInt32, 0, c->nccl, stream);
        if (r != ncclSuccess) return fail(std::string("ncclGather: ") + g_rccl.GetErrorString(r));
        if (c->rank == 0 && tfhe_hip_import_samples_device_async(all, count * c->world, params, c->recv) != 0)
            return fail("import of the gathered samples: " + provider_error());
        return 0;
    }
    std::vector<int32_t> send(words), recv(c->rank == 0 ? words * (size_t)c->world : 0);
    if (tfhe_hip_export_samples(mine, count, params, send.data()) != 0) return fail("export of the samples: " + provider_error());
    if (c->gather(c->ctx, send.data(), c->rank == 0 ? recv.data() : nullptr, words * sizeof(int32_t), 0) != 0)
        return fail("the host gather callback failed");
    if (c->rank == 0 && tfhe_hip_import_samples(all, count * c->world, params, recv.data()) != 0)
        return fail("import of the gathered samples: " + provider_error());
    return 0;
}

}  // namespace

extern "C" {

void peba1_dist_shard_slots(int nslots, int world, int rank, int *lo, int *hi) {
    const int base = nslots / world, rem = nslots % world;
    const int l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

const char *peba1_dist_last_error(void) { return g_error.c_str(); }
int peba1_dist_rank(const Peba1Comm *c) { return c->rank; }
int peba1_dist_world(const Peba1Comm *c) { return c->world; }

int peba1_dist_unique_id(void *id128) {
    static_assert(sizeof(ncclUniqueId) == PEBA1_DIST_ID_BYTES, "RCCL unique id size");
    if (!g_rccl.load()) return fail(std::string("cannot open RCCL: ") + (dlerror() ? dlerror() : "symbols missing"));
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

Peba1Comm *peba1_dist_init_rccl(const void *id128, int world, int rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_init_rccl: bad arguments"); return nullptr; }
    if (!g_rccl.load()) { fail(std::string("cannot open RCCL: ") + (dlerror() ? dlerror() : "symbols missing")); return nullptr; }
    if (!tfhe_hip_stream) { fail("the RCCL transport needs libtfhe-hip as the gate provider"); return nullptr; }
    (void)tfhe_hip_stream();                  // initialises the engine: the device it selected is current for RCCL
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->rccl = true; c->own = true;
    const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { fail(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete c; return nullptr; }
    return c;
}

Peba1Comm *peba1_dist_adopt_rccl(void *nccl_comm, int world, int rank) {
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_adopt_rccl: bad arguments"); return nullptr; }
    if (!g_rccl.load()) { fail("cannot open RCCL"); return nullptr; }
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->rccl = true; c->own = false;
    c->nccl = static_cast<ncclComm_t>(nccl_comm);
    return c;
}

Peba1Comm *peba1_dist_init_host(peba1_gather_fn gather, void *ctx, int world, int rank) {
    if (!gather || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_init_host: bad arguments"); return nullptr; }
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->gather = gather; c->ctx = ctx;
    return c;
}

void peba1_dist_destroy(Peba1Comm *c) {
    if (!c) return;
    if (c->rccl) {
        if (tfhe_hip_stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(tfhe_hip_stream()));
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        if (c->own && c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    }
    delete c;
}

// phase 1 of a rank: the reference's slot loop over its slots, or (PEBA1_DIST_FAST_PARTIAL) the depth-optimised
// distance circuit -- its 24th bit is the carry the reference's 23-bit accumulator drops; both combines read 23 bits
static void partial_phase(LweSample *partial, LweSample *const *a, LweSample *const *b, int nslots_local, int bitsize,
                          const TFheGateBootstrappingCloudKeySet *ck, int flags) {
    if (flags & PEBA1_DIST_FAST_PARTIAL) peba1_euclidean_distance_fast(partial, a, b, nslots_local, bitsize, ck);
    else peba1_partial_distance(partial, a, b, nslots_local, bitsize, ck);
}

int peba1_sharded_partial_packed(LweSample *const *a, LweSample *const *b, int nslots_local, int bitsize,
                                 const TFheGateBootstrappingCloudKeySet *ck, int32_t *packed, int flags) {
    if (!ck || !packed || nslots_local < 0 || (nslots_local > 0 && (!a || !b)) ||
        ((flags & PEBA1_DIST_FAST_PARTIAL) && 3 * bitsize != PARTIAL_BITS))       // that circuit writes 3 * bitsize samples
        return fail("peba1_sharded_partial_packed: bad arguments");
    LweSample *partial = new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, ck->params);
    if (!partial) return fail("allocation of the partial sum: " + provider_error());
    partial_phase(partial, a, b, nslots_local, bitsize, ck, flags);
    const int rc = tfhe_hip_export_samples(partial, PARTIAL_BITS, ck->params, packed);
    delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, partial);
    return rc == 0 ? 0 : fail("export of the partial sum: " + provider_error());
}

static int combine(LweSample *result_b, LweSample *parts, int nparts, LweSample *bound_match,
                   const TFheGateBootstrappingCloudKeySet *ck, int flags) {
    std::vector<LweSample *> ptrs((size_t)nparts);
    for (int r = 0; r < nparts; ++r) ptrs[(size_t)r] = parts + (size_t)r * PARTIAL_BITS;
    if (flags & PEBA1_DIST_FAST_COMBINE) peba1_combine_and_compare_fast(result_b, ptrs.data(), nparts, bound_match, ck);
    else peba1_combine_and_compare(result_b, ptrs.data(), nparts, bound_match, ck);
    return 0;
}

int peba1_sharded_combine_packed(LweSample *result_b, const int32_t *packed, int nparts, LweSample *bound_match,
                                 const TFheGateBootstrappingCloudKeySet *ck, int flags) {
    if (!result_b || !packed || nparts < 1 || !bound_match) return fail("peba1_sharded_combine_packed: bad arguments");
    LweSample *parts = new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * nparts, ck->params);
    if (!parts) return fail("allocation of the partial sums: " + provider_error());
    int rc = tfhe_hip_import_samples(parts, PARTIAL_BITS * nparts, ck->params, packed);
    if (rc == 0) rc = combine(result_b, parts, nparts, bound_match, ck, flags);
    else fail("import of the partial sums: " + provider_error());
    delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * nparts, parts);
    return rc;
}

int peba1_dist_gather_samples(Peba1Comm *c, LweSample *all, const LweSample *mine, int count,
                              const TFheGateBootstrappingParameterSet *params) {
    if (!c || !mine || count < 1 || !params || (c->rank == 0 && !all)) return fail("peba1_dist_gather_samples: bad arguments");
    return gather_samples(c, all, mine, count, params);
}

int peba1_sharded_function_f(Peba1Comm *c, LweSample *result_b, LweSample *const *a, LweSample *const *b,
                             int nslots_local, LweSample *bound_match, int bitsize,
                             const TFheGateBootstrappingCloudKeySet *ck, int flags) {
    if (!c || !ck || nslots_local < 0 || (nslots_local > 0 && (!a || !b)) || (c->rank == 0 && (!result_b || !bound_match)))
        return fail("peba1_sharded_function_f: bad arguments");
    if ((flags & PEBA1_DIST_FAST_PARTIAL) && 3 * bitsize != PARTIAL_BITS)
        return fail("peba1_sharded_function_f: PEBA1_DIST_FAST_PARTIAL needs bitsize 8 (a 24-sample partial sum)");
    // phase 1, every rank: the reference's slot loop over this rank's slots (recorded; the export runs it)
    LweSample *partial = new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, ck->params);
    if (!partial) return fail("allocation of the partial sum: " + provider_error());
    partial_phase(partial, a, b, nslots_local, bitsize, ck, flags);
    // phase 2: ONE exchange, 24 ciphertexts per rank
    LweSample *parts = c->rank == 0 ? new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * c->world, ck->params) : nullptr;
    int rc = gather_samples(c, parts, partial, PARTIAL_BITS, ck->params);
    delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, partial);
    // phase 3, rank 0: add the partial sums, compare with the bound (recorded; runs at the caller's next decrypt /
    // export / flush, ordered behind the import on the provider's stream)
    if (rc == 0 && c->rank == 0) rc = combine(result_b, parts, c->world, bound_match, ck, flags);
    if (parts) delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * c->world, parts);
    return rc;
}

}  // extern "C"
