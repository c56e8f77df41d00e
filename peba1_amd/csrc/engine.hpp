// engine.hpp -- device side of libtfhe-hip: the resident evaluation key, the
// ciphertext slot pool and batched level execution on one HIP stream.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "host_keys.hpp"
#include "br_forms.hpp"
#include "kernels.hpp"
#include "../../include/tfhe_hip.h"

// Device image of one cloud key: NTT image of BK, compact KSK, twiddles.
struct DeviceKeyImage {
    tfhe_hip::DevParams dp;
    tfhe_hip::DevKey key;
    // form_ok[form][tables]: the kernel form's magnitude bounds hold for this key's (l, Bgbit) (br_forms.hpp)
    bool form_ok[tfhe_hip::BR_FORM_COUNT][3] = {};
    uint32_t *bk_img = nullptr;
    int32_t *ksk = nullptr;
    uint32_t *tw = nullptr;
    size_t bk_img_bytes = 0, ksk_bytes = 0, tw_bytes = 0;        // what the three hold (recoverable_alloc's accounting)
};

namespace tfhe_hip {

void set_error(const std::string &msg);
const std::string &last_error_ref();
[[noreturn]] void fatal(const std::string &msg);
void hip_check(hipError_t e, const char *what);
const char *unsupported_reason(const Params &p);   // why the kernels cannot run this parameter set exactly, or null

// A condition the caller can recover from (slot pool exhausted, a sample this library did not
// allocate, a sample used with a key of another LWE dimension, a malformed file): thrown inside
// the library, caught at every extern "C" entry, reported through tfhe_hip_last_error() with
// the call left without effect.  fatal() (abort, like upstream) is kept for HIP runtime
// failures and for "no GPU", after which nothing can work.
struct ApiError { std::string msg; };
// test hook: device allocations of the slot pool and the flush scratch beyond `bytes` in total fail like hipErrorOutOfMemory
// (recoverably: engine.cpp recoverable_alloc); 0 = no cap
void set_alloc_cap(long long bytes);
[[noreturn]] void api_fail(const std::string &msg);

// Pool of device-resident ciphertext slots (ct_stride words each).  Slots are
// immutable once written (the recorder renames every destination), reference
// counted, and recycled through a free list.
// slot ids must fit the 29-bit fields of the recorder's pending-gate keys (shim.cpp gate_key)
constexpr size_t MAX_POOL_SLOTS = (size_t)1 << 29;

class SlotPool {
public:
    SlotPool(int ct_words, int ct_stride, size_t capacity);
    ~SlotPool();
    int32_t alloc();                 // refcount 1, level 0
    void retain(int32_t s) { ++ref_[s]; }
    int32_t refs(int32_t s) const { return ref_[s]; }
    void release(int32_t s);
    int32_t *data() { return data_; }
    int ct_stride() const { return stride_; }
    int ct_words() const { return words_; }
    size_t capacity() const { return max_cap_; }              // what the pool may grow to
    size_t allocated() const { return cap_; }                 // slots backed by device memory now
    size_t in_use() const { return cap_ - free_.size(); }
    std::vector<int32_t> level;      // level of the recorded operation that will write the slot (0: an input)
    std::vector<uint8_t> pending;    // 1 = written by a recorded operation that has not run yet
    int32_t const_slot[2] = {-1, -1};   // shared read-only trivial samples (0, -1/8) and (0, +1/8)
private:
    void grow(size_t new_cap);
    int words_, stride_;
    size_t cap_ = 0, max_cap_;
    int32_t *data_ = nullptr;
    std::vector<int32_t> ref_;
    std::vector<int32_t> free_;
};

struct LevelPlan {
    // descriptors of the whole flush, ordered by level
    std::vector<RotDesc> rots;
    std::vector<KsDesc> kss;
    std::vector<NotDesc> nots;
    int levels = 0;
    std::vector<int32_t> rot_off, ks_off;   // size levels + 1; gates of level L (1-based): index L - 1
    std::vector<int32_t> not_off;           // size levels + 2; NOTs riding on level L (0 = inputs): index L
    int32_t max_rots = 0;                   // widest level (sizes the extract buffer)
};

class Engine {
public:
    static Engine &get();
    void ensure_init();
    int device() const { return device_; }
    int cu_count() const { return cu_count_; }
    void set_device(int d);
    hipStream_t stream() const { return stream_; }

    DeviceKeyImage *upload_key(const TfheHipCloudKey &ck);
    void free_key(DeviceKeyImage *img);
    SlotPool *pool_for(const Params &p);
    SlotPool *find_pool(const Params &p) const;   // the pool of this ciphertext shape if one exists; never initialises the device

    void write_slot(SlotPool *pool, int32_t slot, const Torus32 *a, Torus32 b);       // host -> device
    void read_slot(SlotPool *pool, int32_t slot, Torus32 *a, Torus32 *b);             // device -> host
    // wait = false (device words only): the transfer is enqueued on the engine's stream and the call returns -- whatever
    // is enqueued on that stream afterwards (a collective, the next flush) is ordered behind it without a host wait
    void write_slots_packed(SlotPool *pool, const int32_t *slots, int count, const Torus32 *words, bool words_on_device, bool wait = true);
    void read_slots_packed(SlotPool *pool, const int32_t *slots, int count, Torus32 *words, bool words_on_device, bool wait = true);

    // run a levelised plan.  wait = true: synchronises the stream before returning; false: returns with the launches
    // enqueued ("in flight") -- see engine.cpp
    void execute(const DeviceKeyImage *key, SlotPool *pool, LevelPlan &&plan, bool wait = true);
    void wait_flight();                 // completes an asynchronous execute(): waits, then reads the timing events
    // host waits (engine.cpp "host waits"): bounded by sync_deadline_ms when that is set
    void sync_stream(const char *what); // everything enqueued on the engine's stream has completed
    void sync_io();                     // the stream-ordered transfers nobody waited for have completed
    void wait_all() { wait_flight(); sync_io(); }
    // a caller's event, bounded like the waits above (deadline and label passed in: read by the caller under the recorder lock)
    void wait_event(hipEvent_t ev, const char *what, long long deadline_ms, const std::string &label);
    bool pci_bus_id(char *out, int len);                // "0000:c1:00.0" of the engine's device
    // > 0: no host wait on the engine's stream lasts longer -- on expiry the process prints what it waited for and ends
    // with TFHE_HIP_EXIT_DEADLINE (tuning "sync_deadline_ms", env TFHE_HIP_SYNC_DEADLINE_MS); 0 = wait for ever
    long long sync_deadline_ms = 0;
    std::string diag_label;             // who waits, for the deadline message (tfhe_hip_set_diag_label: "rank 3 of 8")
    bool in_flight() const { return in_flight_; }
    // raw test paths
    void run_bootstrap_woks(const DeviceKeyImage *key, const Torus32 *lin, int count, Torus32 *u_out, Torus32 *acc_out);
    void run_keyswitch(const DeviceKeyImage *key, const Torus32 *u, int count, Torus32 *out);
    void run_negacyclic(const DeviceKeyImage *key, const int32_t *ip, const Torus32 *tp, Torus32 *res, int count);

    TfheHipStats stats{};
    bool kernel_timing = false;
    // Tunings (tfhe_hip_set_tuning; ten names in all: these six, the recorder's reuse_gates / eliminate_dead /
    // balance_levels, and sync_deadline_ms above).  HISTORY.md lists the forms and knobs removed in round 6.
    //
    // Key switches of a narrow launch are split (power of two <= ks_max_splits) until about ks_target_blocks workgroups
    // exist: beyond filling the chip, more splits mean the blocks in flight share a KSK sub-table small enough for an
    // XCD's L2 (measured optimum of the per-gate kernel: 32 splits).  The tiled launches take any count of coefficient
    // ranges up to ks_max_splits whose grid fills whole rounds of resident workgroups (launch_ks); a cap of 48 measured
    // 56.1 against 61.0 ms per match at 32 (env TFHE_HIP_KS_BLOCKS, TFHE_HIP_KS_MAX_SPLITS; not tunings)
    int ks_target_blocks = 32768;
    int ks_max_splits = 48;
    // among the range counts that fill the workgroup slots equally well: 1 = the largest (more, shorter ranges), 0 = the smallest
    // (less partial-sum traffic: one match 55.4 against 56.1 ms; env TFHE_HIP_KS_SPLIT_TIES)
    int ks_split_ties = 0;
    // gates per workgroup of the tiled key switch (16, 24 or 32; 0 = per-gate kernel only); tuning "ks_tile"
    int ks_tile = 16;
    // 1 (default) = the tiled key switch keeps a thread's column of the staged rows in PINNED registers, picked through the
    // VGPR index mode (kernels.hip keyswitch_index_kernel): 56 ms per match; 0 = the LDS-strip form (keyswitch_strip_kernel,
    // tiles of 16): 105 ms -- plain HIP source, the form to fall back on.  Tuning "ks_index", env TFHE_HIP_KS_INDEX
    int ks_index = 1;
    // which form of the blind-rotate kernel runs wide launches (kernels.hip): -1 = the fastest measured for the ring size
    // (N = 1024: 4-wave; N = 2048: split), 0 = 4-wave (N = 1024), 2 = split (8 waves, half transforms), 4 = 2-wave
    // (N = 1024).  A form whose lazy-arithmetic bounds do not admit the key's gadget is replaced by one that does
    // (launch_br).  Tuning "br_variant", env TFHE_HIP_BR_VARIANT
    int br_variant = -1;
    // launches of at most min(this, CU count) rotations (at most one workgroup per CU) use the 8-wave form
    // of the kernel, N = 1024 only; 0 = never (env TFHE_HIP_BR8_MAX, tuning "br8_max_rotations")
    int br8_max_rotations = 1 << 30;
    // A 4-wave launch whose last round would leave at most one workgroup per CU (count = q * 2 * CUs + r, q >= 1,
    // 0 < r <= CUs) hands those r rotations to the 8-wave form as a second launch: 2.9 ms instead of the 3.75 ms a
    // lone 4-wave workgroup per CU takes (env TFHE_HIP_BR_TAIL8, tuning "br_tail8"; 0 = one launch)
    int br_tail8 = 1;
    // 1 = the first radix-4 step of the forward transforms looks digit products up in LDS (gadget digits
    // of at most 7 bits; split form: stage 0, and the first radix-4 step too where digits have at most 6
    // bits); 2 = split form: stage 0 only; 0 = multiplies (env TFHE_HIP_BR_TABLE, tuning "br_digit_table")
    int br_digit_table = 1;
    // stream == nullptr: the engine's stream
    void launch_ks(const DeviceKeyImage *key, const int32_t *u_buf, const KsDesc *descs, int count, int32_t *pool,
                   hipStream_t stream = nullptr);
    // returns true when the launch used the 8-wave form
    bool launch_br(const DeviceKeyImage *key, const int32_t *pool, const RotDesc *rots, int count, int32_t *u_buf,
                   int32_t *acc_dbg, hipStream_t stream = nullptr);
    // diagnostic (tools/wg_times.py): ONE 4-wave blind-rotate launch of `width` random gates whose workgroups stamp s_memtime
    // and s_memrealtime at start and end into wg_times[4 * width]; returns that launch's event time in ms (< 0: no stamps)
    double run_wg_times(const DeviceKeyImage *key, int width, unsigned long long *wg_times);

private:
    Engine() = default;
    void *scratch(size_t idx, size_t bytes);   // grow-only device scratch buffers
    // em / tail: a level whose last round went to the 8-wave form as a second launch (br_tail8) -- the event between
    // the two launches and the rotations of the second, so that each kernel's time and count stay its own
    struct Timed { hipEvent_t e0, e1, e2; bool wide8; int nrot; hipEvent_t em = nullptr; int tail = 0; };
    void note_async_io();
    hipEvent_t io_event_ = nullptr;                     // behind the last stream-ordered transfer that returned without a wait
    bool io_pending_ = false;
    bool in_execute_ = false;                           // launch_br called for a level of execute(): the tail event has a reader
    hipEvent_t tail_event_ = nullptr;                   // set by launch_br when it split a level (kernel_timing only)
    int tail_count_ = 0;
    std::vector<Timed> flight_timed_;
    hipEvent_t flight_base_ = nullptr;
    LevelPlan flight_plan_;
    int flight_levels_ = 0;
    bool in_flight_ = false;
    std::chrono::steady_clock::time_point flight_t0_;
    int32_t *stage_slots(const int32_t *slots, int count);
    int32_t *slot_ring_ = nullptr;
    size_t slot_ring_pos_ = 0;
    int device_ = 0;
    int cu_count_ = 256;
    std::atomic<bool> inited_{false};
    hipStream_t stream_ = nullptr;
    unsigned long long *clock_acc_ = nullptr;           // kernel timing: shader-cycle / reference-tick sums (kernels.hpp)
    unsigned long long *wg_times_dbg_ = nullptr;        // set by the workgroup-time probe only
    size_t timing_used_ = 0;
    hipEvent_t next_timing_event();
    std::vector<hipEvent_t> timing_events_;             // kernel_timing: up to 3 per level + 1 base
    hipEvent_t ev_[3] = {nullptr, nullptr, nullptr};
    std::vector<SlotPool *> pools_;
    std::vector<void *> scratch_ptr_;
    std::vector<size_t> scratch_size_;
};

}  // namespace tfhe_hip
