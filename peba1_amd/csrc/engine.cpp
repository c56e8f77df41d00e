// engine.cpp -- device memory, key upload and batched level execution.
#include "engine.hpp"

#include <algorithm>
#include <utility>
#include <vector>
#include <chrono>
#include <ctime>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "br_forms.hpp"
#include "ntt_field.hpp"

namespace tfhe_hip {

// ---- error channel ---------------------------------------------------------
static std::string g_last_error;
static std::mutex g_err_mtx;

void set_error(const std::string &msg) {
    std::lock_guard<std::mutex> g(g_err_mtx);
    g_last_error = msg;
}
const std::string &last_error_ref() { return g_last_error; }

[[noreturn]] void fatal(const std::string &msg) {
    set_error(msg);
    std::fprintf(stderr, "libtfhe-hip: fatal: %s\n", msg.c_str());
    std::abort();
}

[[noreturn]] void api_fail(const std::string &msg) { throw ApiError{msg}; }

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) fatal(std::string(what) + ": " + hipGetErrorString(e));
}

// Device memory whose exhaustion is RECOVERABLE: the slot pool's growth, the per-flush scratch and
// the device image of a key (a server holds one per enrolled client's key set).
// A library living inside a matching server must not abort() because one request recorded more than the card holds:
// hipErrorOutOfMemory becomes an ApiError -- the call that needed the memory has no effect (the gates recorded so far
// stay recorded; a flush returns -1), tfhe_hip_last_error() says what could not be allocated, and the caller may free
// ciphertext arrays and go on.  Anything else hipMalloc can return is still fatal.  `g_alloc_cap` (test hook
// tfhe_hip_test_set_alloc_cap) makes allocations above a total fail the same way, so that the path has a test.
static long long g_alloc_cap = 0, g_alloc_total = 0;
void set_alloc_cap(long long bytes) { g_alloc_cap = bytes > 0 ? bytes : 0; }
static void *recoverable_alloc(size_t bytes, const char *what) {
    void *p = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    if (!(g_alloc_cap > 0 && g_alloc_total + (long long)bytes > g_alloc_cap)) e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();                 // the sticky error of the failed call must not fail the next launch check
        api_fail(std::string("out of device memory: ") + what + " needs " + std::to_string(bytes >> 20) + " MiB more (" +
                 std::to_string(g_alloc_total >> 20) + " MiB held by key images, the slot pool and the flush scratch); the call "
                 "had no effect -- free ciphertext arrays or key sets, flush less at a time, or lower TFHE_HIP_POOL_SLOTS");
    }
    hip_check(e, what);
    g_alloc_total += (long long)bytes;
    return p;
}
static void recoverable_free(void *p, size_t bytes) {
    if (!p) return;
    (void)hipFree(p);
    g_alloc_total -= (long long)bytes;
}

// HIP's current device is per host thread and starts at 0: a worker thread of a multi-GPU server (rank r drives GPU r)
// that calls in without ever having selected a device -- or after selecting another one for its own work -- would launch
// on the engine's stream with the wrong device current.  Every method below that calls the HIP runtime therefore binds
// the calling thread to the engine's device for its duration and gives the caller's device back on return (ADVICE r4:
// a once-per-thread bind both changed the caller's device for good and missed a later hipSetDevice by the caller).
// The boots* recording calls never reach the runtime and pay nothing for this.
namespace {
class DeviceScope {
public:
    explicit DeviceScope(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev_ = cur;
    }
    ~DeviceScope() { if (prev_ >= 0) (void)hipSetDevice(prev_); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
private:
    int prev_ = -1;
};
}  // namespace
#define ENGINE_DEVICE_SCOPE() DeviceScope device_scope_(Engine::get().device())

// ---- slot pool ---------------------------------------------------------------
// The pool is one contiguous device array (kernels index it by slot id) that starts small and doubles on demand up to
// `capacity` slots: a process that evaluates a few gates holds 0.2 GB, one that records whole matches grows to what its
// widest flush pins (5.3 GB were allocated up front whatever the caller did).
SlotPool::SlotPool(int ct_words, int ct_stride, size_t capacity) : words_(ct_words), stride_(ct_stride), max_cap_(capacity) {
    free_.reserve(std::min<size_t>(max_cap_, (size_t)1 << 17));
    grow(std::min<size_t>(max_cap_, (size_t)1 << 16));
}
SlotPool::~SlotPool() {
    ENGINE_DEVICE_SCOPE();
    recoverable_free(data_, cap_ * (size_t)stride_ * sizeof(int32_t));
}
void SlotPool::grow(size_t new_cap) {
    ENGINE_DEVICE_SCOPE();
    // (throws ApiError when the card is full: nothing has changed yet, the pool keeps its size and its contents)
    int32_t *fresh = static_cast<int32_t *>(recoverable_alloc(new_cap * (size_t)stride_ * sizeof(int32_t), "growing the ciphertext slot pool"));
    if (data_) {
        // growth happens while recording (host side).  A pipelined flush may still be in flight -- its launches carry
        // the old pointer -- and a caller's own stream may still read an exported buffer: wait for the whole device
        // once (every launch enqueued so far has then finished with the old buffer), copy, release
        Engine::get().sync_stream("sync before pool growth");      // bounded when a deadline is set (a collective may sit there)
        hip_check(hipDeviceSynchronize(), "sync before pool growth");
        hip_check(hipMemcpy(fresh, data_, cap_ * (size_t)stride_ * sizeof(int32_t), hipMemcpyDeviceToDevice), "copy slot pool");
        recoverable_free(data_, cap_ * (size_t)stride_ * sizeof(int32_t));
    }
    data_ = fresh;
    ref_.resize(new_cap, 0);
    level.resize(new_cap, 0);
    pending.resize(new_cap, 0);
    for (size_t i = new_cap; i-- > cap_;) free_.push_back((int32_t)i);
    cap_ = new_cap;
}
int32_t SlotPool::alloc() {
    if (free_.empty()) {
        if (cap_ >= max_cap_)
            api_fail("ciphertext slot pool exhausted (" + std::to_string(max_cap_) +
                     " slots); raise TFHE_HIP_POOL_SLOTS or free ciphertext arrays");
        grow(std::min(max_cap_, cap_ * 2));
    }
    const int32_t s = free_.back();
    free_.pop_back();
    ref_[s] = 1;
    level[s] = 0;
    pending[s] = 0;
    return s;
}
void SlotPool::release(int32_t s) {
    if (s < 0) return;
    if (--ref_[s] == 0) free_.push_back(s);
}

// ---- engine ------------------------------------------------------------------
Engine &Engine::get() {
    static Engine e;
    return e;
}

void Engine::set_device(int d) {
    if (inited_.load() && d != device_) fatal("tfhe_hip_set_device after the engine was initialised");
    device_ = d;
}

void Engine::ensure_init() {
    if (inited_.load(std::memory_order_acquire)) return;
    static std::mutex init_mtx;
    std::lock_guard<std::mutex> g(init_mtx);
    if (inited_.load(std::memory_order_relaxed)) return;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        fatal("no HIP device available: libtfhe-hip evaluates gates on the GPU only (there is no CPU fallback)");
    if (const char *env = std::getenv("TFHE_HIP_DEVICE")) device_ = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_KS_BLOCKS")) ks_target_blocks = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_KS_MAX_SPLITS")) ks_max_splits = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_KS_TILE")) ks_tile = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_KS_INDEX")) ks_index = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_KS_SPLIT_TIES")) ks_split_ties = std::atoi(env);
    DeviceScope bind(device_);               // (an invalid device shows at the first runtime call below)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) hip_check(hipSetDevice(device_), "hipSetDevice");
    }
    {
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
        cu_count_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    {
        int least = 0, greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "priority range");
        hip_check(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest), "hipStreamCreate");
    }
    if (const char *env = std::getenv("TFHE_HIP_BR8_MAX")) br8_max_rotations = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_BR_TAIL8")) br_tail8 = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_BR_VARIANT")) br_variant = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_BR_TABLE")) br_digit_table = std::atoi(env);
    if (const char *env = std::getenv("TFHE_HIP_SYNC_DEADLINE_MS")) sync_deadline_ms = std::atoll(env);
    // a caller that cannot reach tfhe_hip_set_kernel_timing (the reference's unmodified program) asks for the per-level
    // times by naming the trace file
    if (std::getenv("TFHE_HIP_TRACE_TIMES")) kernel_timing = true;
    for (auto &e : ev_) hip_check(hipEventCreate(&e), "hipEventCreate");
    inited_.store(true, std::memory_order_release);
}

// ---- host waits ----------------------------------------------------------------
// Every host wait on the engine's stream goes through here.  Without a deadline it is hipStreamSynchronize.  With one
// (tuning "sync_deadline_ms" / TFHE_HIP_SYNC_DEADLINE_MS; libpeba1-dist sets it for communicators of more than one
// rank) the stream is polled, and a wait that outlasts the deadline -- a collective whose peer never arrived, a kernel
// that never ends -- prints what was waited for, by whom, and ends the process with TFHE_HIP_EXIT_DEADLINE: the job fails
// with a message instead of sitting in its scheduler's time limit.  No retry, no re-exec: the device state is unknown.
[[noreturn]] static void deadline_expired(const char *what, long long ms, const std::string &label) {
    std::fprintf(stderr, "libtfhe-hip: fatal: '%s' did not complete within %lld ms%s%s; the engine's stream is stuck behind a "
                         "collective or a kernel that does not finish -- exiting with code %d\n",
                 what, ms, label.empty() ? "" : " on ", label.c_str(), TFHE_HIP_EXIT_DEADLINE);
    std::fflush(stderr);
    _exit(TFHE_HIP_EXIT_DEADLINE);       // not exit(): runtime destructors would wait for the same stream
}

template <typename Query>
static void bounded_wait(Query &&query, const char *what, long long deadline_ms, const std::string &label) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = query();
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) hip_check(e, what);
        if ((spins & 63u) == 63u || spins > 4096u) {
            const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms >= deadline_ms) deadline_expired(what, deadline_ms, label);
        }
        if (spins > 4096u) {                 // a long wait: stop burning the core the recorder may want
            struct timespec ts = {0, 50 * 1000};
            nanosleep(&ts, nullptr);
        }
    }
}

void Engine::sync_stream(const char *what) {
    ENGINE_DEVICE_SCOPE();
    if (sync_deadline_ms <= 0) hip_check(hipStreamSynchronize(stream_), what);
    else bounded_wait([&] { return hipStreamQuery(stream_); }, what, sync_deadline_ms, diag_label);
    io_pending_ = false;                     // whatever was enqueued without a host wait has completed too
}

// Stream-ordered transfers that returned without a host wait (write_/read_slots_packed with wait = false) leave an
// event behind; host accesses of slot memory that do not go through the stream wait for it first (a
// decrypt of a sample imported behind an ncclGather read the slot before the scatter had written it).
void Engine::note_async_io() {
    if (!io_event_) hip_check(hipEventCreateWithFlags(&io_event_, hipEventDisableTiming), "io event");
    hip_check(hipEventRecord(io_event_, stream_), "io event record");
    io_pending_ = true;
}
// a caller's own event (libpeba1-dist: the status words of a collective, on a stream of their own), bounded like every
// other host wait of the library.  Takes no engine state: safe beside a flush in flight.
void Engine::wait_event(hipEvent_t ev, const char *what, long long deadline_ms, const std::string &label) {
    ENGINE_DEVICE_SCOPE();
    if (deadline_ms <= 0) hip_check(hipEventSynchronize(ev), what);
    else bounded_wait([&] { return hipEventQuery(ev); }, what, deadline_ms, label);
}

bool Engine::pci_bus_id(char *out, int len) {
    ENGINE_DEVICE_SCOPE();
    return hipDeviceGetPCIBusId(out, len, device_) == hipSuccess;
}

void Engine::sync_io() {
    ENGINE_DEVICE_SCOPE();
    if (!io_pending_) return;
    if (sync_deadline_ms <= 0) hip_check(hipEventSynchronize(io_event_), "stream-ordered transfer");
    else bounded_wait([&] { return hipEventQuery(io_event_); }, "stream-ordered transfer", sync_deadline_ms, diag_label);
    io_pending_ = false;
}

void *Engine::scratch(size_t idx, size_t bytes) {
    if (scratch_ptr_.size() <= idx) { scratch_ptr_.resize(idx + 1, nullptr); scratch_size_.resize(idx + 1, 0); }
    if (scratch_size_[idx] < bytes) {
        sync_stream("sync before scratch realloc");
        // the old buffer goes first (the two need not fit side by side); if the new one cannot be had the entry is empty,
        // which the next call sees as size 0 -- and the ApiError leaves the caller's call without effect
        recoverable_free(scratch_ptr_[idx], scratch_size_[idx]);
        scratch_ptr_[idx] = nullptr;
        scratch_size_[idx] = 0;
        size_t cap = bytes + bytes / 2 + 4096;
        try {
            scratch_ptr_[idx] = recoverable_alloc(cap, "the scratch buffers of a flush");
        } catch (const ApiError &) {
            if (cap == bytes + 4096) throw;
            cap = bytes + 4096;                  // without the growth margin
            scratch_ptr_[idx] = recoverable_alloc(cap, "the scratch buffers of a flush");
        }
        scratch_size_[idx] = cap;
    }
    return scratch_ptr_[idx];
}

static uint32_t bitrev32(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// twiddles [prime][fwd,inv][N]: psi^{+-brv(i)} * R mod P, followed by the radix-4 quads
// [prime][fwd,inv][N/2][4]: entry e = 2^s + t (stage pair (s, s+1), block t) holds, with
// w1 = W[e], w2 = W[2e], w3 = W[2e+1]: {w2, w3, w1 w2, P - w1 w3}, all times R mod P
static std::vector<uint32_t> make_twiddles(int N, uint32_t scale_out[2]) {
    int logn = 0;
    while ((1 << logn) < N) ++logn;
    std::vector<uint32_t> tw((size_t)24 * N);
    for (int q = 0; q < 2; ++q) {
        const uint64_t P = NTT_P[q];
        const uint64_t psi = powmod_c(NTT_GEN[q], (P - 1) / (uint64_t)(2 * N), P);
        const uint64_t ipsi = powmod_c(psi, P - 2, P);
        const uint64_t R = NTT_R[q];
        std::vector<uint64_t> plain[2] = {std::vector<uint64_t>(N), std::vector<uint64_t>(N)};
        uint64_t a = 1, b = 1;
        for (int i = 0; i < N; ++i) {
            const uint32_t j = bitrev32((uint32_t)i, logn);
            plain[0][j] = a;
            plain[1][j] = b;
            a = a * psi % P;
            b = b * ipsi % P;
        }
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<uint64_t> &W = plain[dir];
            uint32_t *t2 = &tw[(size_t)(q * 2 + dir) * N];
            uint32_t *t4 = &tw[(size_t)4 * N + (size_t)(q * 2 + dir) * 2 * N];
            for (int i = 0; i < N; ++i) t2[i] = (uint32_t)(W[i] * R % P);
            for (int e = 1; e < N / 2; ++e) {
                const uint64_t w1 = W[e], w2 = W[2 * e], w3 = W[2 * e + 1];
                t4[4 * e + 0] = (uint32_t)(w2 * R % P);
                t4[4 * e + 1] = (uint32_t)(w3 * R % P);
                t4[4 * e + 2] = (uint32_t)(w1 * w2 % P * R % P);
                t4[4 * e + 3] = (uint32_t)((P - w1 * w3 % P) % P * R % P);
            }
        }
        // Sub-transform tables of the split kernels (kernels.hip blind_rotate_split_kernel): after
        // stage 0 the two halves of the index range are independent N/2-point transforms whose stage
        // s' twiddle of block t' is W[(2 + h) 2^s' + t'].  Block h (6N words at 12N + 6N h) is laid
        // out exactly like the table of a stand-alone N/2-point transform, so the same code walks it.
        for (int h = 0; h < 2; ++h) {
            const int M = N / 2;
            uint32_t *blk = &tw[(size_t)12 * N + (size_t)h * 12 * M];
            for (int dir = 0; dir < 2; ++dir) {
                const std::vector<uint64_t> &W = plain[dir];
                uint32_t *t2 = blk + (size_t)(q * 2 + dir) * M;
                uint32_t *t4 = blk + (size_t)4 * M + (size_t)(q * 2 + dir) * 2 * M;
                auto full = [&](int e) {                 // e = 2^s' + t' -> (2 + h) 2^s' + t'
                    int top = 1;
                    while (top * 2 <= e) top *= 2;
                    return (2 + h) * top + (e - top);
                };
                t2[0] = 0;
                for (int e = 1; e < M; ++e) t2[e] = (uint32_t)(W[full(e)] * R % P);
                for (int e = 1; e < M / 2; ++e) {
                    const int f = full(e);
                    const uint64_t w1 = W[f], w2 = W[2 * f], w3 = W[2 * f + 1];
                    t4[4 * e + 0] = (uint32_t)(w2 * R % P);
                    t4[4 * e + 1] = (uint32_t)(w3 * R % P);
                    t4[4 * e + 2] = (uint32_t)(w1 * w2 % P * R % P);
                    t4[4 * e + 3] = (uint32_t)((P - w1 * w3 % P) % P * R % P);
                }
            }
        }
        // image scale: N^-1 (the inverse NTT is unscaled) times R (so that the
        // Montgomery reduction of sum x*img leaves sum x*bk / N)
        scale_out[q] = (uint32_t)(powmod_c((uint64_t)N, P - 2, P) * R % P);
    }
    return tw;
}

static DevParams make_dev_params(const Params &p) {
    DevParams d;
    d.n = p.n; d.N = p.N; d.k = p.k; d.l = p.l; d.Bgbit = p.Bgbit; d.ks_t = p.ks_t; d.ks_basebit = p.ks_basebit;
    d.kpl = p.kpl(); d.ct_stride = p.ct_stride(); d.u_stride = p.u_stride();
    d.decomp_offset = p.decomp_offset(); d.ks_prec_offset = p.ks_prec_offset();
    d.mu = 1 << 29;
    d.br_variant = 0;
    d.digit_table = 1;
    d.clock_acc = nullptr;
    d.wg_times = nullptr;
    return d;
}

// why the kernels cannot evaluate gates under `p` exactly, or null.  Also applied by the file loaders (io.cpp).
const char *unsupported_reason(const Params &p) {
    if ((p.N != 1024 && p.N != 2048) || p.k != 1) return "the blind-rotate kernels are built for N = 1024 or 2048 and k = 1";
    if (p.n < 1 || p.n > 1024 || p.ct_stride() / 4 > 320) return "LWE dimension n must be in [1, 1024]";
    if (p.l < 1 || p.Bgbit < 1 || p.l * p.Bgbit > 32 || p.Bgbit > 12) return "gadget digits must fit 12 bits and l * Bgbit <= 32";
    if (p.ks_t < 1 || p.ks_basebit < 1 || p.ks_t * p.ks_basebit > 31 || p.ks_basebit > 8) return "key-switch digits out of range";
    // exactness of the CRT range: (k+1) l N (Bg/2) 2^31 must stay below P0*P1/2
    const double bound = (double)(p.k + 1) * p.l * p.N * (double)(1u << (p.Bgbit - 1)) * 2147483648.0;
    if (bound >= (double)CRT_EXACT_LIMIT) return "gadget parameters exceed the exact range of the two-prime NTT";
    // at least one kernel form must keep its lazy-arithmetic bounds for this (l, Bgbit) (br_forms.hpp)
    for (int f = 0; f < BR_FORM_COUNT; ++f)
        for (int t = 0; t < 3; ++t)
            if (br_form_admissible(f, p.N, p.l, p.Bgbit, t)) return nullptr;
    return "gadget (l, Bgbit) outside the exact range of every blind-rotate kernel form (br_forms.hpp)";
}

DeviceKeyImage *Engine::upload_key(const TfheHipCloudKey &ck) {
    ENGINE_DEVICE_SCOPE();
    ensure_init();
    const Params &p = ck.p;
    // a parameter set the kernels cannot run exactly is refused (the call that needed the key has no effect and
    // tfhe_hip_last_error() says why): a custom tuple or a loaded file may carry anything
    if (const char *why = unsupported_reason(p)) api_fail(why);
    auto *img = new DeviceKeyImage();
    img->dp = make_dev_params(p);
    for (int f = 0; f < BR_FORM_COUNT; ++f)
        for (int t = 0; t < 3; ++t) img->form_ok[f][t] = br_form_admissible(f, p.N, p.l, p.Bgbit, t);
    uint32_t scale[2];
    const std::vector<uint32_t> tw = make_twiddles(p.N, scale);
    const size_t bk_words = p.bk_words();
    const int base = 1 << p.ks_basebit, stride = p.ct_stride();
    const size_t rows = (size_t)p.k * p.N * p.ks_t;
    int32_t *raw = nullptr;
    try {
        // every buffer first (a card without room for them fails here, recoverably: the keyset is not made and
        // tfhe_hip_last_error() says what was needed), then the uploads and the transform
        img->tw_bytes = tw.size() * 4;
        img->tw = static_cast<uint32_t *>(recoverable_alloc(img->tw_bytes, "the twiddle tables of a key"));
        img->bk_img_bytes = bk_words * 2 * 4;
        img->bk_img = static_cast<uint32_t *>(recoverable_alloc(img->bk_img_bytes, "the bootstrapping-key image"));
        img->ksk_bytes = (rows * (base - 1) + 1) * (size_t)stride * 4;
        img->ksk = static_cast<int32_t *>(recoverable_alloc(img->ksk_bytes, "the key-switching key"));
        raw = static_cast<int32_t *>(recoverable_alloc(bk_words * 4, "the staging copy of the bootstrapping key"));
    } catch (const ApiError &) {
        recoverable_free(img->tw, img->tw_bytes);
        recoverable_free(img->bk_img, img->bk_img_bytes);
        recoverable_free(img->ksk, img->ksk_bytes);
        delete img;
        throw;
    }
    hip_check(hipMemcpy(img->tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice), "upload twiddles");

    // BK: upload raw, transform on device
    hip_check(hipMemcpy(raw, ck.bk.data(), bk_words * 4, hipMemcpyHostToDevice), "upload bk");
    launch_bk_transform(stream_, img->dp, raw, img->bk_img, img->tw, p.n * p.kpl(), p.k + 1, scale);
    hip_check(hipGetLastError(), "bk_transform launch");
    sync_stream("bk_transform");
    recoverable_free(raw, bk_words * 4);

    // KSK: drop the all-zero digit-0 rows, pad rows to ct_stride
    std::vector<int32_t> compact((rows * (base - 1) + 1) * stride, 0);   // + one row of zeros (digit 0)
    for (size_t r = 0; r < rows; ++r)
        for (int v = 1; v < base; ++v)
            std::memcpy(&compact[(r * (base - 1) + (v - 1)) * stride], &ck.ksk[(r * base + v) * (size_t)(p.n + 1)],
                        (size_t)(p.n + 1) * 4);
    hip_check(hipMemcpy(img->ksk, compact.data(), compact.size() * 4, hipMemcpyHostToDevice), "upload ksk");

    img->key.bk_img = img->bk_img;
    img->key.ksk = img->ksk;
    img->key.ksk_zero = img->ksk + rows * (base - 1) * stride;
    img->key.tw = img->tw;
    return img;
}

void Engine::free_key(DeviceKeyImage *img) {
    ENGINE_DEVICE_SCOPE();
    if (!img) return;
    if (inited_) sync_stream("sync before key free");
    recoverable_free(img->bk_img, img->bk_img_bytes);
    recoverable_free(img->ksk, img->ksk_bytes);
    recoverable_free(img->tw, img->tw_bytes);
    delete img;
}

SlotPool *Engine::find_pool(const Params &p) const {
    for (SlotPool *pl : pools_)
        if (pl->ct_stride() == p.ct_stride() && pl->ct_words() == p.ct_words()) return pl;
    return nullptr;
}

SlotPool *Engine::pool_for(const Params &p) {
    ENGINE_DEVICE_SCOPE();
    ensure_init();
    if (SlotPool *pl = find_pool(p)) return pl;
    // what the pool may GROW to (it starts at 65,536 slots and doubles on demand): 4,194,304 slots = 10.6 GB at n = 630 of
    // 288 GB of HBM -- eight 128-slot matches recorded while the eight of the flush in flight are still pinned
    size_t cap = 1u << 22;
    if (const char *env = std::getenv("TFHE_HIP_POOL_SLOTS")) cap = (size_t)std::atoll(env);
    // slot ids are packed into 29-bit fields by the recorder's table of pending gates (shim.cpp gate_key)
    if (cap < 8 || cap > MAX_POOL_SLOTS) {
        set_error("TFHE_HIP_POOL_SLOTS out of range [8, 2^29]; clamped");
        cap = cap < 8 ? 8 : MAX_POOL_SLOTS;
    }
    auto *pl = new SlotPool(p.ct_words(), p.ct_stride(), cap);
    // shared read-only slots: the trivial samples (0, -1/8) -- also what a fresh sample is -- and (0, +1/8)
    std::vector<Torus32> z(p.n, 0);
    pl->const_slot[0] = pl->alloc();
    write_slot(pl, pl->const_slot[0], z.data(), -(1 << 29));
    pl->const_slot[1] = pl->alloc();
    write_slot(pl, pl->const_slot[1], z.data(), 1 << 29);
    pools_.push_back(pl);
    return pl;
}

void Engine::write_slot(SlotPool *pool, int32_t slot, const Torus32 *a, Torus32 b) {
    ENGINE_DEVICE_SCOPE();
    // A blocking copy on the null stream is not ordered against the engine's non-blocking stream.  Slots a flush in
    // flight touches stay pinned, so a fresh slot is never one of those; but a slot freed right after a stream-ordered
    // export may still be waiting for its gather kernel: wait for such transfers, not for the whole stream
    // (a pipelined flush keeps running under the recording that calls this).
    sync_io();
    std::vector<int32_t> tmp(pool->ct_stride(), 0);
    std::memcpy(tmp.data(), a, (size_t)(pool->ct_words() - 1) * 4);
    tmp[pool->ct_words() - 1] = b;
    hip_check(hipMemcpy(pool->data() + (size_t)slot * pool->ct_stride(), tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice),
              "write_slot");
}

void Engine::read_slot(SlotPool *pool, int32_t slot, Torus32 *a, Torus32 *b) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();
    // on the engine's stream, then waited for: ordered behind everything enqueued there -- a flush, and a stream-ordered
    // import that nobody waited for (the scatter behind an ncclGather of tfhe_hip_import_samples_device_async)
    std::vector<int32_t> tmp(pool->ct_stride());
    hip_check(hipMemcpyAsync(tmp.data(), pool->data() + (size_t)slot * pool->ct_stride(), tmp.size() * 4, hipMemcpyDeviceToHost, stream_),
              "read_slot");
    sync_stream("read_slot");
    std::memcpy(a, tmp.data(), (size_t)(pool->ct_words() - 1) * 4);
    *b = tmp[pool->ct_words() - 1];
}

// Slot lists travel through a pinned host ring, so that the stream-ordered (no host wait) forms below never read
// host memory the caller has already released.
int32_t *Engine::stage_slots(const int32_t *slots, int count) {
    constexpr size_t RING = 1 << 16;                         // words; one transfer may use at most a quarter
    if (!slot_ring_) hip_check(hipHostMalloc(reinterpret_cast<void **>(&slot_ring_), RING * 4, hipHostMallocDefault), "pinned slot ring");
    if ((size_t)count > RING / 4) {                          // large lists: synchronous staging through the stream
        sync_stream("sync before large slot list");
        slot_ring_pos_ = 0;
        if ((size_t)count > RING) api_fail("too many samples in one packed transfer");
    }
    if (slot_ring_pos_ + (size_t)count > RING) {             // wrap: everything queued so far must have read its list
        sync_stream("sync at slot ring wrap");
        slot_ring_pos_ = 0;
    }
    int32_t *h = slot_ring_ + slot_ring_pos_;
    std::memcpy(h, slots, (size_t)count * 4);
    slot_ring_pos_ += (size_t)count;
    return h;
}

void Engine::write_slots_packed(SlotPool *pool, const int32_t *slots, int count, const Torus32 *words, bool on_device, bool wait) {
    ENGINE_DEVICE_SCOPE();
    if (count <= 0) return;
    const size_t wbytes = (size_t)count * pool->ct_words() * 4;
    int32_t *dslots = static_cast<int32_t *>(scratch(3, (size_t)count * 4));
    hip_check(hipMemcpyAsync(dslots, stage_slots(slots, count), (size_t)count * 4, hipMemcpyHostToDevice, stream_), "upload slot list");
    const int32_t *src = words;
    if (!on_device) {
        int32_t *dw = static_cast<int32_t *>(scratch(4, wbytes));
        hip_check(hipMemcpyAsync(dw, words, wbytes, hipMemcpyHostToDevice, stream_), "upload packed words");
        src = dw;
    }
    launch_scatter_slots(stream_, pool->data(), pool->ct_stride(), pool->ct_words(), dslots, count, src);
    if (wait || !on_device) sync_stream("scatter slots");
    else note_async_io();
}

void Engine::read_slots_packed(SlotPool *pool, const int32_t *slots, int count, Torus32 *words, bool on_device, bool wait) {
    ENGINE_DEVICE_SCOPE();
    if (count <= 0) return;
    const size_t wbytes = (size_t)count * pool->ct_words() * 4;
    // (a list in flight from an earlier stream-ordered call is read by its kernel before this copy lands: one stream)
    int32_t *dslots = static_cast<int32_t *>(scratch(3, (size_t)count * 4));
    hip_check(hipMemcpyAsync(dslots, stage_slots(slots, count), (size_t)count * 4, hipMemcpyHostToDevice, stream_), "upload slot list");
    int32_t *dst = on_device ? words : static_cast<int32_t *>(scratch(4, wbytes));
    launch_gather_slots(stream_, pool->data(), pool->ct_stride(), pool->ct_words(), dslots, count, dst);
    if (!on_device) hip_check(hipMemcpyAsync(words, dst, wbytes, hipMemcpyDeviceToHost, stream_), "download packed words");
    if (wait || !on_device) sync_stream("gather slots");
    else note_async_io();
}

hipEvent_t Engine::next_timing_event() {
    if (timing_used_ == timing_events_.size()) {
        hipEvent_t e;
        hip_check(hipEventCreate(&e), "timing event");
        timing_events_.push_back(e);
    }
    return timing_events_[timing_used_++];
}

bool Engine::launch_br(const DeviceKeyImage *key, const int32_t *pool, const RotDesc *rots, int count, int32_t *u_buf,
                       int32_t *acc_dbg, hipStream_t stream) {
    ENGINE_DEVICE_SCOPE();
    if (!stream) stream = stream_;
    tail_event_ = nullptr;                   // what an earlier launch left is not this one's
    tail_count_ = 0;
    DevParams dp = key->dp;
    // the form the tunings ask for ...
    int form;
    if (br_variant == 4 && dp.N == 1024) form = BR_FORM_WAVE2;
    else if (br_variant == 2 || dp.N == 2048) form = BR_FORM_SPLIT;
    // launches that leave CUs with at most one workgroup: the 8-wave form (a second wave per SIMD)
    else if (br8_max_rotations > 0 && count <= std::min(br8_max_rotations, cu_count_) && dp.l >= 2 && !wg_times_dbg_)
        form = BR_FORM_WAVE8;
    else form = BR_FORM_WIDE4;
    // the workgroup-time probe reads stamps only the 4-wave kernel writes
    if (wg_times_dbg_ && dp.N == 1024) form = BR_FORM_WIDE4;
    // ... if its magnitude bounds hold for this key's gadget (br_forms.hpp; every built-in set passes in every form of its
    // ring); else the same form with smaller or no digit tables, else the next form of the order
    int tables = br_digit_table < 0 || br_digit_table > 2 ? 0 : br_digit_table;
    if (!key->form_ok[form][tables]) {
        static const int order[BR_FORM_COUNT] = {BR_FORM_WIDE4, BR_FORM_SPLIT, BR_FORM_WAVE2, BR_FORM_WAVE8};
        int pick_f = -1, pick_t = 0;
        for (int k = -1; k < BR_FORM_COUNT && pick_f < 0; ++k) {
            const int f = k < 0 ? form : order[k];
            if (f == BR_FORM_WAVE8 && count > cu_count_) continue;          // its LDS allows one workgroup per CU only
            for (int t : {tables, 2, 0})
                if (key->form_ok[f][t]) { pick_f = f; pick_t = t; break; }
        }
        if (pick_f < 0) fatal("no admissible blind-rotate form (upload_key should have refused this key)");
        form = pick_f; tables = pick_t;
    }
    dp.digit_table = tables;
    if (kernel_timing) {
        if (!clock_acc_) {
            hip_check(hipMalloc(&clock_acc_, 2 * sizeof(unsigned long long)), "hipMalloc(clock sums)");
            hip_check(hipMemset(clock_acc_, 0, 2 * sizeof(unsigned long long)), "hipMemset(clock sums)");
        }
        dp.clock_acc = clock_acc_;
    }
    if (form == BR_FORM_WAVE2) {
        launch_blind_rotate2(stream, dp, key->key, pool, rots, count, u_buf, acc_dbg);
        return false;
    }
    if (form == BR_FORM_SPLIT) {
        launch_blind_rotate_split(stream, dp, key->key, pool, rots, count, u_buf, acc_dbg);
        return false;
    }
    dp.wg_times = wg_times_dbg_;
    if (form == BR_FORM_WAVE8) {
        launch_blind_rotate8(stream, dp, key->key, pool, rots, count, u_buf, acc_dbg);
        return true;
    }
    // the last, at most half-filled round of a wide launch on the 8-wave form (descriptors carry their own output
    // index, so a level splits anywhere; the debug probes index by workgroup and keep one launch)
    const int round = 2 * cu_count_, tail = count % round;
    if (br_tail8 && count > round && tail > 0 && tail <= std::min(br8_max_rotations, cu_count_) && dp.l >= 2 &&
        !acc_dbg && !wg_times_dbg_ && key->form_ok[BR_FORM_WAVE8][tables]) {
        launch_blind_rotate4(stream, dp, key->key, pool, rots, count - tail, u_buf, nullptr);
        tail_count_ = tail;
        // the event between the two launches belongs to execute()'s per-flush set (reset there); the raw test paths and
        // probes have no reader for it and must not grow the set
        if (kernel_timing && in_execute_) { tail_event_ = next_timing_event(); hip_check(hipEventRecord(tail_event_, stream), "event"); }
        launch_blind_rotate8(stream, dp, key->key, pool, rots + (count - tail), tail, u_buf, nullptr);
        return false;
    }
    launch_blind_rotate4(stream, dp, key->key, pool, rots, count, u_buf, acc_dbg);
    return false;
}

void Engine::launch_ks(const DeviceKeyImage *key, const int32_t *u_buf, const KsDesc *descs, int count, int32_t *pool,
                       hipStream_t stream) {
    ENGINE_DEVICE_SCOPE();
    if (count <= 0) return;
    if (!stream) stream = stream_;
    const DevParams &dp = key->dp;
    const int nin = dp.k * dp.N;
    // tiles of 24 or 32 gates exist in the index form only (kernels.hip keyswitch_index_kernel)
    const int ks_tile = (this->ks_tile > 16 && !ks_index) ? 16 : this->ks_tile;
    // tiled kernel: wide launches, ranges of at most 64 input coefficients
    const bool tiled = ks_tile > 0 && count >= 2 * ks_tile && dp.ks_t == 8 && dp.ks_basebit == 2 && ks_max_splits > 1 &&
                       (nin + ks_max_splits - 1) / ks_max_splits <= 64;
    const int chunk = tiled ? 8192 : count;              // bounds the partial-sum buffer (0.66 GB at P128)
    for (int done = 0; done < count; done += chunk) {
        const int cnt = std::min(chunk, count - done);
        int splits = 1;
        if (tiled && cnt >= 2 * ks_tile) {
            // the number of coefficient ranges is free between nin/64 and ks_max_splits: take the
            // one whose grid (tiles x ranges) fills whole rounds of the workgroups the chip holds,
            // e.g. 36 tiles x 28 ranges = 1008 of 1024 slots in two rounds instead of 36 x 32 = 1152 in three
            const int threads = ((dp.ct_stride / 4 + 63) / 64) * 64;
            // index form: no LDS strips; 120 VGPRs at tile 16 (four waves per SIMD), 162 at 24 (three), 204 at 32 (two);
            // strip form: ~235 VGPRs (two waves per SIMD) and 16 x threads x 16 bytes of strips
            int per_cu = std::max(1, (ks_tile == 16 ? 16 : ks_tile == 24 ? 12 : 8) / (threads / 64));
            if (!ks_index) {
                const size_t lds = (size_t)16 * threads * 16 + (size_t)ks_tile * 65 * 4;
                per_cu = std::max(1, std::min((int)((160 * 1024) / lds), 8 / (threads / 64)));
            }
            const long long slots = (long long)cu_count_ * per_cu;
            const long long tiles = (cnt + ks_tile - 1) / ks_tile;
            const int lo = std::max(2, (nin + 63) / 64);
            double best = -1.0;
            for (int sp = lo; sp <= ks_max_splits; ++sp) {
                const long long blocks = tiles * sp, rounds = (blocks + slots - 1) / slots;
                const double eff = (double)blocks / (double)(rounds * slots);
                // ties: fewer, longer ranges (less partial-sum traffic) or more, shorter ones (ks_split_ties)
                if (ks_split_ties ? eff >= best - 1e-9 : eff > best + 1e-9) { best = eff; splits = sp; }
            }
        } else {
            while (splits < ks_max_splits && cnt * splits * 2 <= ks_target_blocks) splits *= 2;
        }
        int32_t *partial = nullptr;
        if (splits > 1) partial = static_cast<int32_t *>(scratch(10, (size_t)cnt * splits * dp.ct_stride * 4));
        launch_keyswitch(stream, dp, key->key, u_buf, descs + done, cnt, pool, splits, partial, tiled ? ks_tile : 0, ks_index != 0);
    }
}

// wait = false: the launches are enqueued and the call returns; the flush is "in flight" until wait_flight() (called by
// the next execute() before it touches the descriptor buffers, by every host read of a slot, by the statistics).  The
// host work of the NEXT flush -- recording, dead-gate elimination, levelling, building its plan -- then overlaps this
// one's execution (shim.cpp flush_locked).
void Engine::execute(const DeviceKeyImage *key, SlotPool *pool, LevelPlan &&plan_in, bool wait) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();                                   // at most one flush in flight: its descriptors and scratch are in use
    flight_t0_ = std::chrono::steady_clock::now();
    // Every device buffer of the flush is sized here, before anything is enqueued and before the plan changes hands:
    // scratch() may reallocate, which must not happen under a running launch -- and it may throw (out of device memory:
    // recoverable_alloc), in which case nothing has run, the caller's recorded gates are still pending and its flush
    // returns -1 (shim.cpp flush_locked releases nothing before this call returns)
    RotDesc *drots = static_cast<RotDesc *>(scratch(0, plan_in.rots.size() * sizeof(RotDesc) + 16));
    KsDesc *dks = static_cast<KsDesc *>(scratch(1, plan_in.kss.size() * sizeof(KsDesc) + 16));
    NotDesc *dnots = static_cast<NotDesc *>(scratch(2, plan_in.nots.size() * sizeof(NotDesc) + 16));
    // extract buffer and key-switch partial sums, sized for the widest level
    int32_t *u_buf = static_cast<int32_t *>(scratch(5, (size_t)(plan_in.max_rots + 1) * key->dp.u_stride * 4));
    (void)scratch(10, (size_t)std::min(plan_in.max_rots + 1, 8192) * ks_max_splits * key->dp.ct_stride * 4);
    flight_plan_ = std::move(plan_in);               // owns the host descriptors until the uploads have certainly happened
    const LevelPlan &plan = flight_plan_;
    const int levels = plan.levels;
    if (!plan.rots.empty())
        hip_check(hipMemcpyAsync(drots, plan.rots.data(), plan.rots.size() * sizeof(RotDesc), hipMemcpyHostToDevice, stream_), "upload rots");
    if (!plan.kss.empty())
        hip_check(hipMemcpyAsync(dks, plan.kss.data(), plan.kss.size() * sizeof(KsDesc), hipMemcpyHostToDevice, stream_), "upload ks");
    if (!plan.nots.empty())
        hip_check(hipMemcpyAsync(dnots, plan.nots.data(), plan.nots.size() * sizeof(NotDesc), hipMemcpyHostToDevice, stream_), "upload nots");

    timing_used_ = 0;                                    // timing events used: base, then 2-3 (one more with a tail launch) per level
    auto timing_event = [&]() { return next_timing_event(); };
    std::vector<Timed> &timed = flight_timed_;
    timed.clear();
    hipEvent_t &base = flight_base_;
    base = nullptr;
    if (kernel_timing) {
        base = timing_event();
        hip_check(hipEventRecord(base, stream_), "event");
        timed.reserve((size_t)levels + 1);
    }
    hipEvent_t shared_end = nullptr;
    for (int L = 0; L <= levels; ++L) {
        const size_t gg = L > 0 ? (size_t)(L - 1) : 0;       // gate index
        const int nrot = L > 0 ? plan.rot_off[gg + 1] - plan.rot_off[gg] : 0;
        const int nks = L > 0 ? plan.ks_off[gg + 1] - plan.ks_off[gg] : 0;
        const int nnot = plan.not_off[(size_t)L + 1] - plan.not_off[(size_t)L];
        if (nrot == 0 && nks == 0 && nnot == 0) continue;
        Timed t{nullptr, nullptr, nullptr, false, nrot};
        // a level's start event is the previous level's end event where nothing was enqueued in between (no NOT launch
        // behind the key switch): two events per level instead of three -- an event costs the stream a few
        // microseconds, 1,131 of them 4-15 ms of a match
        if (kernel_timing) {
            if (shared_end) t.e0 = shared_end;
            else { t.e0 = timing_event(); hip_check(hipEventRecord(t.e0, stream_), "event"); }
            shared_end = nullptr;
        }
        if (nrot) {
            in_execute_ = true;
            t.wide8 = launch_br(key, pool->data(), drots + plan.rot_off[gg], nrot, u_buf, nullptr, stream_);
            in_execute_ = false;
            if (t.wide8) { ++stats.br8_launches; stats.br8_rotations += (uint64_t)nrot; }
            if (tail_count_) {                       // a second launch, of the 8-wave kernel
                t.em = tail_event_;
                t.tail = tail_count_;
                ++stats.br8_launches; ++stats.br_launches;
                stats.br8_rotations += (uint64_t)tail_count_;
            }
        }
        if (kernel_timing) { t.e1 = timing_event(); hip_check(hipEventRecord(t.e1, stream_), "event"); }
        if (nks) launch_ks(key, u_buf, dks + plan.ks_off[gg], nks, pool->data(), stream_);
        if (kernel_timing) {
            t.e2 = timing_event(); hip_check(hipEventRecord(t.e2, stream_), "event"); timed.push_back(t);
            if (nnot == 0) shared_end = t.e2;
        }
        launch_not(stream_, key->dp, dnots + plan.not_off[(size_t)L], nnot, pool->data());
        stats.blind_rotates += (uint64_t)nrot;
        stats.keyswitches += (uint64_t)nks;
        stats.linear_ops += (uint64_t)nnot;
        if (nrot) ++stats.br_launches;
    }
    if (const char *trace = std::getenv("TFHE_HIP_TRACE_LEVELS")) {   // diagnostic: rotations per level
        if (FILE *f = std::fopen(trace, "a")) {
            std::fprintf(f, "flush levels=%d lanes=1\n", levels);
            for (int L = 1; L <= levels; ++L)
                std::fprintf(f, "%d 0 %d\n", L, plan.rot_off[(size_t)L] - plan.rot_off[(size_t)L - 1]);
            std::fclose(f);
        }
    }
    hip_check(hipGetLastError(), "kernel launch");
    flight_levels_ = levels;
    in_flight_ = true;
    if (wait) wait_flight();
}

void Engine::wait_flight() {
    ENGINE_DEVICE_SCOPE();
    if (!in_flight_) return;
    in_flight_ = false;
    const int levels = flight_levels_;
    std::vector<Timed> &timed = flight_timed_;
    hipEvent_t base = flight_base_;
    sync_stream("level execution");
    if (kernel_timing && base) {
        // durations per launch, and the union of the blind-rotate intervals
        std::vector<std::pair<float, float>> br;
        br.reserve(timed.size());
        // diagnostic: start, blind-rotate and key-switch time of every level (TFHE_HIP_TRACE_TIMES = file)
        FILE *tf = nullptr;
        if (const char *trace = std::getenv("TFHE_HIP_TRACE_TIMES")) tf = std::fopen(trace, "a");
        // "flush levels=L start_ms=S": S = host time of the flush's start since the process's first flush (the gaps between
        // flushes are the caller's own time: recording, encrypting, decrypting)
        static const auto trace_t0 = flight_t0_;
        if (tf) std::fprintf(tf, "flush levels=%d start_ms=%.3f\n", levels,
                             std::chrono::duration<double, std::milli>(flight_t0_ - trace_t0).count());
        for (const Timed &t : timed) {
            float a = 0, b = 0, c = 0;
            hip_check(hipEventElapsedTime(&a, base, t.e0), "elapsed");
            hip_check(hipEventElapsedTime(&b, base, t.e1), "elapsed");
            hip_check(hipEventElapsedTime(&c, base, t.e2), "elapsed");
            if (tf) std::fprintf(tf, "%d %.4f %.4f %.4f %d\n", t.nrot, a, b - a, c - b, t.wide8 ? 1 : 0);
            stats.ms_blind_rotate += b - a;
            if (t.wide8) stats.ms_blind_rotate8 += b - a;
            if (t.em) {
                float m = 0;
                hip_check(hipEventElapsedTime(&m, base, t.em), "elapsed");
                stats.ms_blind_rotate8 += b - m;
            }
            stats.ms_keyswitch += c - b;
            if (b > a) br.emplace_back(a, b);
        }
        std::sort(br.begin(), br.end());
        float cur_a = 0, cur_b = -1;
        for (const auto &iv : br) {
            if (cur_b < cur_a || iv.first > cur_b) {
                if (cur_b >= cur_a) stats.ms_blind_rotate_busy += cur_b - cur_a;
                cur_a = iv.first; cur_b = iv.second;
            } else {
                cur_b = std::max(cur_b, iv.second);
            }
        }
        if (cur_b >= cur_a) stats.ms_blind_rotate_busy += cur_b - cur_a;
        if (tf) {
            std::fprintf(tf, "wall_ms=%.3f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - flight_t0_).count());
            std::fclose(tf);
        }
    }
    if (kernel_timing && clock_acc_) {
        unsigned long long sums[2] = {0, 0};
        hip_check(hipMemcpy(sums, clock_acc_, sizeof sums, hipMemcpyDeviceToHost), "read clock sums");
        hip_check(hipMemset(clock_acc_, 0, sizeof sums), "clear clock sums");
        stats.clk_shader_cycles += sums[0];
        stats.clk_ref_ticks += sums[1];
    }
    stats.levels += (uint64_t)levels;
    ++stats.flushes;
    stats.ms_flush_wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - flight_t0_).count();
    timed.clear();
    flight_plan_ = LevelPlan{};
}


void Engine::run_bootstrap_woks(const DeviceKeyImage *key, const Torus32 *lin, int count, Torus32 *u_out, Torus32 *acc_out) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();
    const DevParams &dp = key->dp;
    // temporary "pool": count slots holding lin
    std::vector<int32_t> padded((size_t)count * dp.ct_stride, 0);
    for (int c = 0; c < count; ++c)
        std::memcpy(&padded[(size_t)c * dp.ct_stride], lin + (size_t)c * (dp.n + 1), (size_t)(dp.n + 1) * 4);
    int32_t *dpool = static_cast<int32_t *>(scratch(6, padded.size() * 4));
    hip_check(hipMemcpyAsync(dpool, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, stream_), "upload lin");
    std::vector<RotDesc> rots(count);
    for (int c = 0; c < count; ++c) rots[c] = RotDesc{c, c, 1, 0, 0, c};
    RotDesc *drots = static_cast<RotDesc *>(scratch(0, rots.size() * sizeof(RotDesc)));
    hip_check(hipMemcpyAsync(drots, rots.data(), rots.size() * sizeof(RotDesc), hipMemcpyHostToDevice, stream_), "upload rots");
    int32_t *u_buf = static_cast<int32_t *>(scratch(5, (size_t)count * dp.u_stride * 4));
    int32_t *dacc = acc_out ? static_cast<int32_t *>(scratch(7, (size_t)count * 2 * dp.N * 4)) : nullptr;
    launch_br(key, dpool, drots, count, u_buf, dacc);
    hip_check(hipGetLastError(), "blind_rotate launch");
    std::vector<int32_t> ubuf((size_t)count * dp.u_stride);
    hip_check(hipMemcpyAsync(ubuf.data(), u_buf, ubuf.size() * 4, hipMemcpyDeviceToHost, stream_), "download u");
    if (acc_out) hip_check(hipMemcpyAsync(acc_out, dacc, (size_t)count * 2 * dp.N * 4, hipMemcpyDeviceToHost, stream_), "download acc");
    sync_stream("bootstrap_woks");
    for (int c = 0; c < count; ++c)
        std::memcpy(u_out + (size_t)c * (dp.k * dp.N + 1), &ubuf[(size_t)c * dp.u_stride], (size_t)(dp.k * dp.N + 1) * 4);
    stats.blind_rotates += (uint64_t)count;
}

void Engine::run_keyswitch(const DeviceKeyImage *key, const Torus32 *u, int count, Torus32 *out) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();
    const DevParams &dp = key->dp;
    const int uw = dp.k * dp.N + 1;
    std::vector<int32_t> padded((size_t)count * dp.u_stride, 0);
    for (int c = 0; c < count; ++c) std::memcpy(&padded[(size_t)c * dp.u_stride], u + (size_t)c * uw, (size_t)uw * 4);
    int32_t *u_buf = static_cast<int32_t *>(scratch(5, padded.size() * 4));
    hip_check(hipMemcpyAsync(u_buf, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, stream_), "upload u");
    std::vector<KsDesc> ks(count);
    for (int c = 0; c < count; ++c) ks[c] = KsDesc{c, -1, 0, c};
    KsDesc *dks = static_cast<KsDesc *>(scratch(1, ks.size() * sizeof(KsDesc)));
    hip_check(hipMemcpyAsync(dks, ks.data(), ks.size() * sizeof(KsDesc), hipMemcpyHostToDevice, stream_), "upload ks");
    int32_t *dpool = static_cast<int32_t *>(scratch(6, (size_t)count * dp.ct_stride * 4));
    launch_ks(key, u_buf, dks, count, dpool);
    hip_check(hipGetLastError(), "keyswitch launch");
    std::vector<int32_t> res((size_t)count * dp.ct_stride);
    hip_check(hipMemcpyAsync(res.data(), dpool, res.size() * 4, hipMemcpyDeviceToHost, stream_), "download ks");
    sync_stream("keyswitch");
    for (int c = 0; c < count; ++c) std::memcpy(out + (size_t)c * (dp.n + 1), &res[(size_t)c * dp.ct_stride], (size_t)(dp.n + 1) * 4);
    stats.keyswitches += (uint64_t)count;
}

double Engine::run_wg_times(const DeviceKeyImage *key, int width, unsigned long long *wg_times) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();
    const DevParams &dp = key->dp;
    if (width <= 0 || !wg_times || dp.N != 1024) return -1.0;       // the stamps come from the 4-wave kernel only
    // a private "pool": `width` random input ciphertexts
    std::vector<int32_t> host((size_t)width * dp.ct_stride);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto &w : host) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (int32_t)(x >> 16); }
    int32_t *pool = static_cast<int32_t *>(scratch(40, (size_t)width * dp.ct_stride * 4));
    hip_check(hipMemcpy(pool, host.data(), host.size() * 4, hipMemcpyHostToDevice), "probe pool");
    std::vector<RotDesc> rots(width);
    for (int i = 0; i < width; ++i) rots[i] = RotDesc{i, (i + 1) % width, 1, 1, -dp.mu, i};
    RotDesc *drots = static_cast<RotDesc *>(scratch(41, rots.size() * sizeof(RotDesc)));
    hip_check(hipMemcpy(drots, rots.data(), rots.size() * sizeof(RotDesc), hipMemcpyHostToDevice), "probe rots");
    int32_t *ubuf = static_cast<int32_t *>(scratch(50, (size_t)(width + 1) * dp.u_stride * 4));
    unsigned long long *dtimes = static_cast<unsigned long long *>(scratch(43, (size_t)4 * width * 8));
    hipStream_t st;
    hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "probe stream");
    hipEvent_t e0, e1;
    hip_check(hipEventCreate(&e0), "probe event");
    hip_check(hipEventCreate(&e1), "probe event");
    // launches back to back, the last one stamped and timed: what a level of a circuit sees
    // (TFHE_HIP_PROBE_WARM = "count:width" changes the unstamped launches before it, default one of the same width)
    int warm_count = 1, warm_width = width;
    if (const char *env = std::getenv("TFHE_HIP_PROBE_WARM")) std::sscanf(env, "%d:%d", &warm_count, &warm_width);
    warm_width = std::max(1, std::min(warm_width, width));
    for (int w = 0; w < warm_count; ++w) launch_br(key, pool, drots, warm_width, ubuf, nullptr, st);
    // cleared first so that a launch which wrote no stamps is noticed instead of read as timings
    hip_check(hipMemsetAsync(dtimes, 0, (size_t)4 * width * 8, st), "clear stamps");
    wg_times_dbg_ = dtimes;
    hip_check(hipEventRecord(e0, st), "probe event record");
    launch_br(key, pool, drots, width, ubuf, nullptr, st);
    hip_check(hipEventRecord(e1, st), "probe event record");
    hip_check(hipStreamSynchronize(st), "probe stamps");
    wg_times_dbg_ = nullptr;
    float ems = 0.f;
    hip_check(hipEventElapsedTime(&ems, e0, e1), "probe event time");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    hip_check(hipMemcpy(wg_times, dtimes, (size_t)4 * width * 8, hipMemcpyDeviceToHost), "probe stamps copy");
    hip_check(hipStreamDestroy(st), "probe stream destroy");
    for (int i = 0; i < width; ++i)
        if (wg_times[4 * i] == 0 || wg_times[4 * i + 1] == 0) return -1.0;      // a workgroup left no stamp
    return (double)ems;
}

void Engine::run_negacyclic(const DeviceKeyImage *key, const int32_t *ip, const Torus32 *tp, Torus32 *res, int count) {
    ENGINE_DEVICE_SCOPE();
    wait_flight();
    DevParams dp = key->dp;
    dp.br_variant = br_variant == 2 ? 2 : 0;            // 2: through the split transforms
    const size_t words = (size_t)count * dp.N;
    uint32_t scale[2];
    (void)make_twiddles(dp.N, scale);
    int32_t *dtp = static_cast<int32_t *>(scratch(6, words * 4));
    int32_t *dip = static_cast<int32_t *>(scratch(7, words * 4));
    uint32_t *dimg = static_cast<uint32_t *>(scratch(8, words * 2 * 4));
    int32_t *dres = static_cast<int32_t *>(scratch(9, words * 4));
    hip_check(hipMemcpyAsync(dtp, tp, words * 4, hipMemcpyHostToDevice, stream_), "upload tp");
    hip_check(hipMemcpyAsync(dip, ip, words * 4, hipMemcpyHostToDevice, stream_), "upload ip");
    launch_bk_transform(stream_, dp, dtp, dimg, key->tw, count, 1, scale);
    launch_negacyclic(stream_, dp, key->tw, dip, dimg, dres, count);
    hip_check(hipGetLastError(), "negacyclic launch");
    hip_check(hipMemcpyAsync(res, dres, words * 4, hipMemcpyDeviceToHost, stream_), "download res");
    sync_stream("negacyclic");
}

}  // namespace tfhe_hip
