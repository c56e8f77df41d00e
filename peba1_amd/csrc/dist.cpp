// dist.cpp -- libpeba1-dist: the multi-GPU forms of the match for a C-ABI host (include/peba1_dist.h).
// Host logic only: slot partition, the one exchange per match, rank 0's combine.  Every gate runs in the gate
// provider (libtfhe-hip); the circuits are libpeba1-circuits'.  RCCL is opened with dlopen on first use.
#include "../../include/peba1_dist.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
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
__attribute__((weak)) int tfhe_hip_stream_sync(void);
__attribute__((weak)) int tfhe_hip_wait_event(void *, const char *);
__attribute__((weak)) int tfhe_hip_get_device(void);
__attribute__((weak)) int tfhe_hip_set_tuning(const char *, int64_t);
__attribute__((weak)) void tfhe_hip_set_diag_label(const char *);
__attribute__((weak)) int tfhe_hip_flush_async(void);
__attribute__((weak)) int tfhe_hip_wait(void);
__attribute__((weak)) void tfhe_hip_set_deferred(int);
__attribute__((weak)) int tfhe_hip_get_deferred(void);
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
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;      // optional (NCCL >= 2.18)
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Bcast)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
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
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(handle, "ncclAllGather"));
        Bcast = reinterpret_cast<decltype(Bcast)>(dlsym(handle, "ncclBcast"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(handle, "ncclGetVersion"));     // evidence only: may be absent
        CommSplit = reinterpret_cast<decltype(CommSplit)>(dlsym(handle, "ncclCommSplit"));        // may be absent: status_comm()
        return GetUniqueId && CommInitRank && CommDestroy && Gather && AllGather && Bcast && GetErrorString;
    }
};
Rccl g_rccl;

}  // namespace

struct Peba1Comm {
    int world = 1, rank = 0;
    bool rccl = false, own = false;
    ncclComm_t nccl = nullptr;
    peba1_gather_fn gather = nullptr;
    peba1_bcast_fn bcast = nullptr;
    void *ctx = nullptr;
    // grow-only device buffers of the RCCL path: they must outlive the stream operations that use them
    int32_t *send = nullptr, *recv = nullptr;
    size_t send_words = 0, recv_words = 0;
    // status exchange of the RCCL path: [0] this rank's word, [1 .. world] every rank's (device, and a pinned host mirror);
    // it runs on a stream of its own so that reading the words back does not wait for the gates in flight on the
    // provider's stream (exchange_status_rccl)
    int32_t *st_dev = nullptr, *st_host = nullptr;
    hipStream_t st_stream = nullptr;
    hipEvent_t st_event = nullptr;
    // ... and on a COMMUNICATOR of its own: split off the data communicator when this one is made
    // (ncclCommSplit, every rank in one colour).  Two communicators are independent by contract, whatever streams their
    // collectives are enqueued on; one communicator used from two streams relies on the library chaining its kernels.
    // Null where the loaded RCCL has no ncclCommSplit or the split failed: the status words then ride on the data
    // communicator AND the provider's stream -- one communicator, one stream, trivially ordered, at the price of waiting
    // for the gates in flight.
    ncclComm_t st_nccl = nullptr;
    int inject_failures = 0;             // test hook (peba1_dist_inject_failure): local failures still to report
    // what this communicator has done (peba1_dist_counters): status exchanges, gathers, broadcasts, payload bytes sent
    uint64_t n_status = 0, n_gather = 0, n_bcast = 0, bytes_sent = 0;
    // the ORDER it has issued collectives in (peba1_dist_sequence): how many, and a rolling hash of (kind, words per rank,
    // root) -- equal on every rank of a job iff every rank issued the same collectives in the same order
    uint64_t seq_count = 0, seq_hash = 0xcbf29ce484222325ull;
    void issued(int kind, uint64_t words, int root) {
        for (uint64_t v : {(uint64_t)kind, words, (uint64_t)(root + 1)}) { seq_hash ^= v; seq_hash *= 0x100000001b3ull; }
        ++seq_count;
    }
};

namespace {

// this library calls the HIP runtime itself (exchange buffers, the status words, RCCL): on the provider's device, whatever
// the calling thread had current -- and the caller gets its own device back on return
class DeviceScope {
public:
    DeviceScope() {
        if (tfhe_hip_stream) (void)tfhe_hip_stream();                 // initialises the provider if nothing has yet
        if (!tfhe_hip_get_device) return;
        const int want = tfhe_hip_get_device();
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev_ = cur;
    }
    ~DeviceScope() { if (prev_ >= 0) (void)hipSetDevice(prev_); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
private:
    int prev_ = -1;
};

// every host wait of this library goes through the provider's bounded wait (tfhe_hip.h "bounded host waits")
void stream_sync() {
    if (tfhe_hip_stream_sync) (void)tfhe_hip_stream_sync();
    else if (tfhe_hip_stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(tfhe_hip_stream()));
}

bool device_buffer(int32_t *&buf, size_t &have, size_t want) {
    if (have >= want) return true;
    // the stream may still be using the old buffer
    if (buf) { stream_sync(); (void)hipFree(buf); buf = nullptr; have = 0; }
    if (hipMalloc(reinterpret_cast<void **>(&buf), want * sizeof(int32_t)) != hipSuccess) return false;
    have = want;
    return true;
}

// A communicator of several ranks bounds every host wait of the provider (a peer that never arrives must end the job
// with a message, not with the scheduler's time limit) and says who is waiting.
void arm_deadline(Peba1Comm *c) {
    if (c->world <= 1) return;
    long long secs = 600;
    if (const char *env = std::getenv("PEBA1_DIST_TIMEOUT_S")) secs = std::atoll(env);
    if (tfhe_hip_set_tuning && secs > 0) (void)tfhe_hip_set_tuning("sync_deadline_ms", (int64_t)secs * 1000);
    if (tfhe_hip_set_diag_label) {
        const std::string label = "rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) + " (libpeba1-dist)";
        tfhe_hip_set_diag_label(label.c_str());
    }
}

// Failure containment around a collective: a rank that failed locally -- an allocation,
// the export that runs its pending gates -- must not leave the others inside the collective.  Every rank therefore
// ENTERS the exchange whatever happened to it, carrying a status word:
//   RCCL: one ncclAllGather of the status words on the provider's stream, read back by every rank (a bounded host
//     wait); if any rank reports a failure, every rank skips the data gather and returns -1 naming that rank;
//   host transport (gather only): the status word rides in front of the payload; rank 0 checks all of them and
//     returns -1 naming the rank, the failed rank returns -1 with its own message.
// Returns the first failed rank, -1 if none, -2 if the exchange itself failed (g_error set).
int exchange_status_rccl(Peba1Comm *c, int local) {
    if (!c->st_dev) {
        if (hipMalloc(reinterpret_cast<void **>(&c->st_dev), (size_t)(c->world + 1) * sizeof(int32_t)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&c->st_host), (size_t)(c->world + 1) * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithFlags(&c->st_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->st_event, hipEventDisableTiming) != hipSuccess) {
            // cannot even report: the peers run into their deadline with a message of their own
            fail("allocation of the status words failed");
            return -2;
        }
    }
    // On a stream AND a communicator of their own: the words say what the HOST of every rank knows after
    // it has enqueued its export -- they do not depend on the gates in flight, so reading them back must not wait for
    // those, nor for the previous call's data collective.  Without a communicator of their own (Peba1Comm::st_nccl) they
    // take the data communicator on the provider's stream: ordered by the stream, behind the gates.
    const bool own = c->st_nccl != nullptr;
    hipStream_t stream = own ? c->st_stream : static_cast<hipStream_t>(tfhe_hip_stream());
    constexpr int32_t UNSET = INT32_MIN;      // a word still holding this after the wait was never received
    c->st_host[0] = local;
    for (int r = 0; r < c->world; ++r) c->st_host[1 + r] = UNSET;
    if (hipMemcpyAsync(c->st_dev, c->st_host, sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess) { fail("upload of the status word failed"); return -2; }
    const ncclResult_t r = g_rccl.AllGather(c->st_dev, c->st_dev + 1, 1, ncclInt32, own ? c->st_nccl : c->nccl, stream);
    if (r != ncclSuccess) { fail(std::string("ncclAllGather of the status words: ") + g_rccl.GetErrorString(r)); return -2; }
    c->issued(0, 1, -1);
    if (hipMemcpyAsync(c->st_host + 1, c->st_dev + 1, (size_t)c->world * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipEventRecord(c->st_event, stream) != hipSuccess) {
        fail("download of the status words failed");
        return -2;
    }
    ++c->n_status;
    // bounded: a peer that never arrives ends the process with a message (the provider's deadline, its exit code).  A wait
    // that FAILED is a failed exchange: the words were never received and must not read as "no rank failed"
    const bool waited = tfhe_hip_wait_event ? tfhe_hip_wait_event(c->st_event, "status words of a collective") == 0
                                            : hipEventSynchronize(c->st_event) == hipSuccess;
    if (!waited) { fail("the wait for the status words failed: " + provider_error()); return -2; }
    for (int k = 0; k < c->world; ++k)
        if (c->st_host[1 + k] == UNSET) { fail("status word of rank " + std::to_string(k) + " was not received"); return -2; }
    for (int k = 0; k < c->world; ++k)
        if (c->st_host[1 + k] != 0) return k;
    return -1;
}

// `count` samples of every rank -> rank 0, rank-major.  RCCL: device buffers, everything on the provider's stream;
// host transport: through the callback.  local / why: what already went wrong on this rank (0 / "" = nothing).
int gather_samples(Peba1Comm *c, LweSample *all, const LweSample *mine, int count, const TFheGateBootstrappingParameterSet *params,
                   int local = 0, std::string why = std::string()) {
    const size_t words = (size_t)tfhe_hip_sample_words(params) * (size_t)count;
    auto note = [&](const std::string &msg) { if (local == 0) { local = -1; why = msg; } };
    if (c->inject_failures > 0) { --c->inject_failures; note("injected failure (peba1_dist_inject_failure)"); }
    if (c->rank == 0 && !all) note("no destination array on rank 0");
    if (c->rccl) {
        if (!tfhe_hip_export_samples_device_async || !tfhe_hip_import_samples_device_async || !tfhe_hip_stream)
            return fail("the RCCL transport needs libtfhe-hip as the gate provider");
        DeviceScope on_provider_device;
        if (local == 0 && !device_buffer(c->send, c->send_words, words)) note("hipMalloc of the send buffer failed");
        if (local == 0 && c->rank == 0 && !device_buffer(c->recv, c->recv_words, words * (size_t)c->world)) note("hipMalloc of the receive buffer failed");
        hipStream_t stream = static_cast<hipStream_t>(tfhe_hip_stream());
        // runs the pending gates, then gathers the slots into the buffer -- enqueued, not waited for
        if (local == 0 && tfhe_hip_export_samples_device_async(mine, count, params, c->send) != 0) note("export of the samples: " + provider_error());
        const int bad = exchange_status_rccl(c, local);
        if (bad == -2) return -1;
        if (bad >= 0)
            return fail(bad == c->rank ? "rank " + std::to_string(bad) + " (this rank) failed before the gather: " + why
                                       : "rank " + std::to_string(bad) + " reported a failure before the gather; no rank entered it");
        const ncclResult_t r = g_rccl.Gather(c->send, c->rank == 0 ? c->recv : nullptr, words, ncclInt32, 0, c->nccl, stream);
        if (r != ncclSuccess) return fail(std::string("ncclGather: ") + g_rccl.GetErrorString(r));
        ++c->n_gather; c->bytes_sent += words * sizeof(int32_t);
        c->issued(1, words, 0);
        if (c->rank == 0 && tfhe_hip_import_samples_device_async(all, count * c->world, params, c->recv) != 0)
            return fail("import of the gathered samples: " + provider_error());
        return 0;
    }
    // host transport: [status word][payload] per rank
    std::vector<int32_t> send(1 + words, 0), recv(c->rank == 0 ? (1 + words) * (size_t)c->world : 0);
    if (local == 0 && tfhe_hip_export_samples(mine, count, params, send.data() + 1) != 0) note("export of the samples: " + provider_error());
    send[0] = local;
    if (c->gather(c->ctx, send.data(), c->rank == 0 ? recv.data() : nullptr, (1 + words) * sizeof(int32_t), 0) != 0)
        return fail("the host gather callback failed");
    ++c->n_gather; ++c->n_status; c->bytes_sent += (1 + words) * sizeof(int32_t);      // the status word rides in front
    c->issued(1, words, 0);
    if (local != 0) return fail("rank " + std::to_string(c->rank) + " (this rank) failed before the gather: " + why);
    if (c->rank != 0) return 0;
    for (int k = 0; k < c->world; ++k)
        if (recv[(size_t)k * (1 + words)] != 0)
            return fail("rank " + std::to_string(k) + " reported a failure before the gather; its contribution is void");
    std::vector<int32_t> packed(words * (size_t)c->world);
    for (int k = 0; k < c->world; ++k)
        std::memcpy(packed.data() + (size_t)k * words, recv.data() + (size_t)k * (1 + words) + 1, words * sizeof(int32_t));
    if (tfhe_hip_import_samples(all, count * c->world, params, packed.data()) != 0)
        return fail("import of the gathered samples: " + provider_error());
    return 0;
}

// `count` samples of `root` -> every rank.  The status exchange of the RCCL path is the all-gather above; over the host
// transport the status word of the ROOT rides in front of the payload (a root that failed broadcasts a failure), a failed
// receiver still takes part and returns -1 itself.
int broadcast_samples(Peba1Comm *c, LweSample *samples, int count, const TFheGateBootstrappingParameterSet *params, int root) {
    const size_t words = (size_t)tfhe_hip_sample_words(params) * (size_t)count;
    int local = 0;
    std::string why;
    auto note = [&](const std::string &msg) { if (local == 0) { local = -1; why = msg; } };
    if (c->inject_failures > 0) { --c->inject_failures; note("injected failure (peba1_dist_inject_failure)"); }
    if (c->rccl) {
        if (!tfhe_hip_export_samples_device_async || !tfhe_hip_import_samples_device_async || !tfhe_hip_stream)
            return fail("the RCCL transport needs libtfhe-hip as the gate provider");
        DeviceScope on_provider_device;
        if (local == 0 && !device_buffer(c->send, c->send_words, words)) note("hipMalloc of the broadcast buffer failed");
        hipStream_t stream = static_cast<hipStream_t>(tfhe_hip_stream());
        if (local == 0 && c->rank == root && tfhe_hip_export_samples_device_async(samples, count, params, c->send) != 0)
            note("export of the samples: " + provider_error());
        const int bad = exchange_status_rccl(c, local);
        if (bad == -2) return -1;
        if (bad >= 0)
            return fail(bad == c->rank ? "rank " + std::to_string(bad) + " (this rank) failed before the broadcast: " + why
                                       : "rank " + std::to_string(bad) + " reported a failure before the broadcast; no rank entered it");
        const ncclResult_t r = g_rccl.Bcast(c->send, words, ncclInt32, root, c->nccl, stream);
        if (r != ncclSuccess) return fail(std::string("ncclBcast: ") + g_rccl.GetErrorString(r));
        ++c->n_bcast; if (c->rank == root) c->bytes_sent += words * sizeof(int32_t);
        c->issued(2, words, root);
        if (c->rank != root && tfhe_hip_import_samples_device_async(samples, count, params, c->send) != 0)
            return fail("import of the broadcast samples: " + provider_error());
        return 0;
    }
    if (!c->bcast) return fail("the host transport has no broadcast: give one to peba1_dist_set_host_bcast");
    std::vector<int32_t> buf(1 + words, 0);
    if (c->rank == root) {
        if (local == 0 && tfhe_hip_export_samples(samples, count, params, buf.data() + 1) != 0) note("export of the samples: " + provider_error());
        buf[0] = local;
    }
    if (c->bcast(c->ctx, buf.data(), (1 + words) * sizeof(int32_t), root) != 0) return fail("the host broadcast callback failed");
    ++c->n_bcast; ++c->n_status; if (c->rank == root) c->bytes_sent += (1 + words) * sizeof(int32_t);
    c->issued(2, words, root);
    if (local != 0) return fail("rank " + std::to_string(c->rank) + " (this rank) failed before the broadcast: " + why);
    if (buf[0] != 0) return fail("rank " + std::to_string(root) + " (the root) reported a failure before the broadcast; its payload is void");
    if (c->rank != root && tfhe_hip_import_samples(samples, count, params, buf.data() + 1) != 0)
        return fail("import of the broadcast samples: " + provider_error());
    return 0;
}

}  // namespace

extern "C" {

void peba1_dist_set_host_bcast(Peba1Comm *c, peba1_bcast_fn bcast) { if (c) c->bcast = bcast; }

int peba1_dist_broadcast_samples(Peba1Comm *c, LweSample *samples, int count, const TFheGateBootstrappingParameterSet *params,
                                 int root) {
    if (!c || !samples || count < 1 || !params || root < 0 || root >= c->world) return fail("peba1_dist_broadcast_samples: bad arguments");
    return broadcast_samples(c, samples, count, params, root);
}

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

// the status words' own communicator (Peba1Comm::st_nccl): a collective over the data communicator -- every rank makes its
// Peba1Comm at the same point, so every rank is here together.  Failure is not an error: the fallback is documented there.
static void status_comm(Peba1Comm *c) {
    if (!g_rccl.CommSplit || std::getenv("PEBA1_DIST_NO_STATUS_COMM")) return;
    ncclComm_t split = nullptr;
    if (g_rccl.CommSplit(c->nccl, 0, c->rank, &split, nullptr) == ncclSuccess && split) c->st_nccl = split;
}

Peba1Comm *peba1_dist_init_rccl(const void *id128, int world, int rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_init_rccl: bad arguments"); return nullptr; }
    if (!g_rccl.load()) { fail(std::string("cannot open RCCL: ") + (dlerror() ? dlerror() : "symbols missing")); return nullptr; }
    if (!tfhe_hip_stream) { fail("the RCCL transport needs libtfhe-hip as the gate provider"); return nullptr; }
    DeviceScope on_provider_device;           // initialises the engine: the device it selected is current for RCCL
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->rccl = true; c->own = true;
    const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { fail(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete c; return nullptr; }
    status_comm(c);
    arm_deadline(c);
    return c;
}

Peba1Comm *peba1_dist_adopt_rccl(void *nccl_comm, int world, int rank) {
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_adopt_rccl: bad arguments"); return nullptr; }
    if (!g_rccl.load()) { fail("cannot open RCCL"); return nullptr; }
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->rccl = true; c->own = false;
    c->nccl = static_cast<ncclComm_t>(nccl_comm);
    DeviceScope on_provider_device;
    status_comm(c);
    arm_deadline(c);
    return c;
}

Peba1Comm *peba1_dist_init_host(peba1_gather_fn gather, void *ctx, int world, int rank) {
    if (!gather || world < 1 || rank < 0 || rank >= world) { fail("peba1_dist_init_host: bad arguments"); return nullptr; }
    auto *c = new Peba1Comm();
    c->world = world; c->rank = rank; c->gather = gather; c->ctx = ctx;
    arm_deadline(c);
    return c;
}

void peba1_dist_destroy(Peba1Comm *c) {
    if (!c) return;
    if (c->rccl) {
        DeviceScope on_provider_device;
        stream_sync();
        // (every status exchange was waited for when it was made: nothing is in flight on the status stream; the wait is a formality)
        if (c->st_stream) (void)hipStreamSynchronize(c->st_stream);
        if (c->st_nccl) (void)g_rccl.CommDestroy(c->st_nccl);           // the split-off communicator is ours even when the parent is adopted
        if (c->own && c->nccl) (void)g_rccl.CommDestroy(c->nccl);
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        if (c->st_dev) (void)hipFree(c->st_dev);
        if (c->st_host) (void)hipHostFree(c->st_host);
        if (c->st_event) (void)hipEventDestroy(c->st_event);
        if (c->st_stream) (void)hipStreamDestroy(c->st_stream);
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
    // (a missing destination on rank 0 is reported THROUGH the exchange, so that the other ranks are not left in it)
    if (!c || !mine || count < 1 || !params) return fail("peba1_dist_gather_samples: bad arguments");
    return gather_samples(c, all, mine, count, params);
}

int peba1_sharded_function_f(Peba1Comm *c, LweSample *result_b, LweSample *const *a, LweSample *const *b,
                             int nslots_local, LweSample *bound_match, int bitsize,
                             const TFheGateBootstrappingCloudKeySet *ck, int flags) {
    if (!c || !ck || nslots_local < 0 || (nslots_local > 0 && (!a || !b)) || (c->rank == 0 && (!result_b || !bound_match)))
        return fail("peba1_sharded_function_f: bad arguments");
    if ((flags & PEBA1_DIST_FAST_PARTIAL) && 3 * bitsize != PARTIAL_BITS)
        return fail("peba1_sharded_function_f: PEBA1_DIST_FAST_PARTIAL needs bitsize 8 (a 24-sample partial sum)");
    // phase 1, every rank: the reference's slot loop over this rank's slots (recorded; the export runs it).  From here
    // on nothing returns before the exchange: a local failure travels through it (gather_samples)
    int local = 0;
    std::string why;
    LweSample *partial = new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, ck->params);
    if (!partial) { local = -1; why = "allocation of the partial sum: " + provider_error(); }
    else partial_phase(partial, a, b, nslots_local, bitsize, ck, flags);
    // phase 2: ONE exchange, 24 ciphertexts per rank
    LweSample *parts = c->rank == 0 ? new_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * c->world, ck->params) : nullptr;
    if (c->rank == 0 && !parts && local == 0) { local = -1; why = "allocation of the gathered partial sums: " + provider_error(); }
    int rc = gather_samples(c, parts, partial, PARTIAL_BITS, ck->params, local, why);
    // (the exported slots may be released at once: a later host write of a recycled slot waits for the stream-ordered
    // export first -- engine.cpp write_slot)
    if (partial) delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS, partial);
    // phase 3, rank 0: add the partial sums, compare with the bound (recorded; runs at the caller's next decrypt /
    // export / flush, ordered behind the import on the provider's stream)
    if (rc == 0 && c->rank == 0) rc = combine(result_b, parts, c->world, bound_match, ck, flags);
    if (parts) delete_gate_bootstrapping_ciphertext_array(PARTIAL_BITS * c->world, parts);
    return rc;
}

int peba1_dist_rccl_version(void) {
    int v = 0;
    if (!g_rccl.load() || !g_rccl.GetVersion || g_rccl.GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

void peba1_dist_counters(const Peba1Comm *c, uint64_t out4[4]) {
    if (!c || !out4) return;
    out4[0] = c->n_status; out4[1] = c->n_gather; out4[2] = c->n_bcast; out4[3] = c->bytes_sent;
}

int peba1_dist_transport(const Peba1Comm *c) { return c && c->rccl ? 1 : 0; }

int peba1_dist_status_channel(const Peba1Comm *c) { return !c || !c->rccl ? 0 : c->st_nccl ? 2 : 1; }

void peba1_dist_sequence(const Peba1Comm *c, uint64_t out2[2]) {
    if (!c || !out2) return;
    out2[0] = c->seq_count; out2[1] = c->seq_hash;
}

void peba1_dist_inject_failure(Peba1Comm *c, int count) { if (c) c->inject_failures = count > 0 ? count : 0; }

int peba1_dist_set_timeout(Peba1Comm *c, double seconds) {
    if (!c || seconds < 0) return fail("peba1_dist_set_timeout: bad arguments");
    if (!tfhe_hip_set_tuning) return fail("the gate provider has no bounded waits (not libtfhe-hip)");
    return tfhe_hip_set_tuning("sync_deadline_ms", (int64_t)(seconds * 1000.0)) == 0 ? 0 : fail("sync_deadline_ms refused");
}

// 1-to-N identification (BASELINE configs[3]; the loop a server puts around /root/reference/src/main.cpp:533-542, one
// Function_f per enrolled client): this rank's m_local matches, recorded `group` at a time -- a flush runs the pending
// gates of the whole group level by level, so the narrow tail levels of one match are filled by the others, and
// releases their slots: device memory is bounded by `group`, not by m_local.  Pipelined: a group's launches are
// enqueued and the next group is recorded while the device works.  Only the match-bit ciphertexts are kept; one gather
// brings them to rank 0.
int peba1_identify(Peba1Comm *c, LweSample *all, LweSample *mine, LweSample *const *probe, LweSample *const *templates,
                   int m_local, int nslots, LweSample *bound_match, int bitsize, const TFheGateBootstrappingCloudKeySet *ck,
                   int group, int flags) {
    if (!mine || !probe || !templates || m_local < 1 || nslots < 1 || !bound_match || bitsize < 1 || !ck || group < 1)
        return fail("peba1_identify: bad arguments");
    const int was_deferred = tfhe_hip_get_deferred ? tfhe_hip_get_deferred() : -1;
    if (tfhe_hip_set_deferred) tfhe_hip_set_deferred(1);            // record; the flushes below run the batches
    int local = 0;
    std::string why;
    for (int first = 0; first < m_local && local == 0; first += group) {
        const int cnt = m_local - first < group ? m_local - first : group;
        for (int m = first; m < first + cnt; ++m) {
            LweSample *rb = new_gate_bootstrapping_ciphertext_array(3 * bitsize, ck->params);
            if (!rb) { local = -1; why = "allocation of a result array: " + provider_error(); break; }
            LweSample *const *tmpl = templates + (size_t)m * (size_t)nslots;
            if (flags & PEBA1_IDENTIFY_FAST) peba1_function_f_fast(rb, probe, tmpl, nslots, bound_match, bitsize, ck);
            else peba1_function_f(rb, probe, tmpl, nslots, bound_match, bitsize, ck);
            bootsCOPY(mine + m, rb, ck);                            // re-points a handle: no data moves
            delete_gate_bootstrapping_ciphertext_array(3 * bitsize, rb);
        }
        // the launches of this group are enqueued and the host goes on recording the next group; the next flush (or the
        // final wait / export) completes this one
        if (local == 0 && tfhe_hip_flush_async && tfhe_hip_flush_async() < 0) { local = -1; why = "flush: " + provider_error(); }
    }
    int rc = 0;
    if (c && (c->world > 1 || all)) {
        rc = gather_samples(c, all, mine, m_local, ck->params, local, why);
    } else if (local != 0) {
        // (no wait here: the last group stays in flight like the others -- a decrypt / export of the match bits, or the next
        // call's first flush, completes it; a server streaming probe after probe keeps the device busy across calls)
        rc = fail(why);
    }
    if (tfhe_hip_set_deferred && was_deferred == 0) tfhe_hip_set_deferred(0);
    return rc;
}

}  // extern "C"
