// shim.cpp -- the C ABI of libtfhe-hip: the tfhe boots* surface that
// /root/reference/src/Math.cpp and src/main.cpp bind plus the
// tfhe_hip_* extensions.  Host logic only: allocation, the SSA recorder,
// levelisation and hand-off to the engine.  No gate arithmetic happens here.
//
// Recorder model.  Every ciphertext value lives in an immutable device slot.
// A boots* call allocates a fresh destination slot, records an operation that
// reads its operands' CURRENT slots, and re-points the destination handle to
// the new slot (SSA renaming).  That makes the reference's patterns safe under
// deferral: result aliasing an input (Math.cpp:272), a temporary overwritten
// four times (Math.cpp:34-42), temporaries freed right after use
// (Math.cpp:47-49).  The level of an operation is 1 + the maximum level of its
// operand slots; a flush executes level 1, 2, ... as batched kernel launches.
// bootsCOPY and bootsCONSTANT only re-point handles (no data moves).
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "br_forms.hpp"
#include "engine.hpp"
#include "scheduler.hpp"
#include "../../include/tfhe/tfhe.h"

using namespace tfhe_hip;

namespace tfhe_hip {
const std::string &last_error_ref();
bool io_forget_owned_cloud(const void *ks);     // io.cpp: keysets created by the file loaders
bool io_forget_owned_secret(const void *ks);
}

namespace {

constexpr uint32_t ARRAY_MAGIC = 0x7F4E11A5u;
constexpr int32_t SLOT_HOST = -1;   // value lives in the host mirror
constexpr int32_t SLOT_ZERO = -2;   // fresh sample: trivial encryption of bit 0, (0, -1/8)

// Hidden header in front of every LweSample array handed to the caller.
struct alignas(16) ArrayHeader {
    uint32_t magic;
    int32_t count;
    int32_t n;
    int32_t unused;
    SlotPool *pool;     // set when the first element moves to the device
};

// Every sample's host mirror is preceded by a back-pointer to its array's header and a marker, so
// the pool a sample lives in can be found from any element pointer (&array[i]).  The marker is
// checked BEFORE the back-pointer is followed.  Why it matters: the reference's Function_g writes
// bitsize+1 results into a bitsize-sample array (Math.cpp:399-401).  That "sample" one
// past the end overlays the first mirror prefix of the block, so its `a` field reads as the
// back-pointer (the block's own start): the words in front of THAT are the allocator's chunk header
// -- readable, and not the marker -- and the call is refused instead of following garbage.
constexpr int MIRROR_PREFIX_WORDS = 4;   // one pointer + marker (8 bytes)
constexpr uint64_t MIRROR_MARKER = 0x7F4E5A3D1E5AA11Cull;

ArrayHeader *header_of(const LweSample *sample) {
    uint64_t marker;
    ArrayHeader *h;
    if (!sample || !sample->a) api_fail("LweSample was not allocated by new_gate_bootstrapping_ciphertext_array");
    std::memcpy(&marker, sample->a - 2, sizeof marker);
    if (marker != MIRROR_MARKER) api_fail("LweSample was not allocated by new_gate_bootstrapping_ciphertext_array");
    std::memcpy(&h, sample->a - MIRROR_PREFIX_WORDS, sizeof h);
    if (!h || h->magic != ARRAY_MAGIC) api_fail("LweSample was not allocated by new_gate_bootstrapping_ciphertext_array");
    return h;
}

// bind the sample's array to `pool` on first device use; one array never spans two pools
void bind_pool(const LweSample *sample, SlotPool *pool) {
    ArrayHeader *h = header_of(sample);
    if (!h->pool) h->pool = pool;
    else if (h->pool != pool) api_fail("ciphertext used with a key of different LWE dimension than the one it was first used with");
}

// prelude constants of the two-input gates: (c0 in eighths, sa, sb), tfhe boot-gates.cpp
struct GateLin { int32_t c8, sa, sb; };
const GateLin GATE_LIN[10] = {
    {1, -1, -1}, {1, 1, 1}, {-1, 1, 1}, {-1, -1, -1}, {2, 2, 2}, {-2, -2, -2},
    {-1, -1, 1}, {-1, 1, -1}, {1, -1, 1}, {1, 1, -1},
};

// truth tables of the two-input gates, index [gate code][2 a + b] (enum TfheHipGate; GATE_LIN's signs say the same)
const uint8_t GATE_TT[10][4] = {
    {1, 1, 1, 0}, {0, 1, 1, 1}, {0, 0, 0, 1}, {1, 0, 0, 0}, {0, 1, 1, 0}, {1, 0, 0, 1},
    {0, 1, 0, 0}, {0, 0, 1, 0}, {1, 1, 0, 1}, {1, 0, 1, 1},
};

struct Recorder {
    std::recursive_mutex mtx;
    // Default: deferred.  The reference (and any caller that stays inside the tfhe C API) never
    // reads a field of LweSample -- results are only ever observed through bootsSymDecrypt or an
    // export, and both run the pending gates first -- so recording is transparent to it and is
    // what lets an UNMODIFIED caller run at batch throughput (one gate per launch is 3.4 ms:
    // 290 gates/s).  TFHE_HIP_DEFERRED=0 / tfhe_hip_set_deferred(0) restores strict per-call
    // completion with the host mirror (a, b) refreshed on return, as upstream's own struct has it.
    bool deferred = true;
    const TFheGateBootstrappingCloudKeySet *key = nullptr;   // key of the pending operations
    SlotPool *pool = nullptr;
    std::vector<PendingOp> ops;
    int32_t max_level = 0;
    // bootsSymEncrypt and the default keygen draw from two ChaCha20 streams keyed by the OS (host_keys.hpp: one for
    // the noise / the secrets, one for the public masks); after tfhe_hip_set_encrypt_seed both roles fall to ONE
    // seeded xoshiro generator (enc_secret; the specified draw order of the fixtures).
    Rng enc_secret = Rng::secure(), enc_mask = Rng::secure();
    bool enc_seeded = false;
    bool balance_levels = true;   // slack-aware level filling (scheduler.hpp)
    std::unordered_map<int32_t, int32_t> not_origin;   // pending NOT output slot -> its operand slot
    // Pending gates by (kind, operand slots): a gate recorded again with the same operands before
    // the flush is the same function of the same ciphertexts, so its result slot is shared
    // instead of evaluated twice (the reference's circuits do this 12,545 times per match,
    // mostly AND / XOR against the shared constant samples).  Results are unchanged.
    bool reuse_gates = true;
    // Dead-gate elimination at flush: an operation whose destination slot is held by nothing but the operation's own
    // pending reference -- every handle that pointed at it has been re-pointed or freed, and no live operation reads it
    // -- can never be observed, so it is dropped (and with it, transitively, what only it read).  The reference's
    // ripple adders compute a carry out of their last bit and drop it (Math.cpp:60-64 into a freed temporary): 5 of the
    // 7 gates of that bit, ~55 gates per slot of the match.
    bool eliminate_dead = true;
    // Constant folding at record time (round 6; OPT-IN: tuning "fold_constants", env TFHE_HIP_FOLD_CONSTANTS).  A trivial
    // sample -- bootsCONSTANT, a fresh sample, a copy of either -- is a PUBLIC constant, and a gate with such an operand
    // needs no bootstrap to be evaluated: its result is a constant, the other operand, or its negation (linear); a MUX with
    // a constant data operand is a two-input gate.  The reference's circuits are full of them (zero-padded partial products,
    // adders fed with constant zeros: 62 % of the 215,544 gates of a 128-slot Function_f).  Decrypted results are the
    // same; the CIPHERTEXTS are not what TFHE produces (it bootstraps every gate whatever its operands), which is why this is
    // off by default -- the drop-in's contract is TFHE's words.  The oracle's provider folds by the same rule
    // (oracle/boots_oracle.c orc_boots_set_fold), so folded circuits have oracle digests of their own.
    bool fold_constants = false;
    // operations of a flush that was launched without waiting (tfhe_hip_flush_async): their pending references are
    // released when the flush is known to be complete (finish_flight_locked)
    std::vector<PendingOp> flight_ops;
    SlotPool *flight_pool = nullptr;
    std::unordered_map<uint64_t, int32_t> pending_gate;            // two-input gates and NOT: key -> result slot
    std::map<std::array<int32_t, 3>, int32_t> pending_mux;         // MUX: (a, b, c) -> result slot
};
// key of a pending gate; symmetric two-input gates (sa == sb in GATE_LIN) with ordered operands.
// Injective: kind < 64 and slot ids < MAX_POOL_SLOTS = 2^29 (enforced where a pool is created).
static_assert(MAX_POOL_SLOTS <= ((size_t)1 << 29), "slot ids must fit the 29-bit fields of gate_key");
uint64_t gate_key(int kind, int32_t a, int32_t b) {
    return ((uint64_t)(uint32_t)kind << 58) ^ ((uint64_t)(uint32_t)a << 29) ^ (uint64_t)(uint32_t)b;
}
Recorder &rec() {
    static Recorder r;
    static bool init = [] {
        if (const char *e = std::getenv("TFHE_HIP_DEFERRED")) r.deferred = std::atoi(e) != 0;
        if (const char *e = std::getenv("TFHE_HIP_FOLD_CONSTANTS")) r.fold_constants = std::atoi(e) != 0;   // opt-in (Recorder)
        return true;
    }();
    (void)init;
    return r;
}

// Every extern "C" entry runs its body through one of these: an ApiError becomes
// tfhe_hip_last_error() and the call has no effect (void entries) or returns -1.
template <typename F>
void guarded(F &&body) {
    try { body(); } catch (const ApiError &e) { set_error(e.msg); }
}
template <typename F>
int guarded_rc(F &&body) {
    try { return body(); } catch (const ApiError &e) { set_error(e.msg); return -1; }
}

SlotPool *pool_of_key(const TFheGateBootstrappingCloudKeySet *bk) {
    if (!bk || !bk->bk) api_fail("null cloud key");
    // keysets made host-only or loaded from a file get their device image at first use
    // (aborts with a clear message when there is no GPU: gates are never evaluated on the CPU)
    if (!bk->bk->dev) bk->bk->dev = Engine::get().upload_key(*bk->bk);
    return Engine::get().pool_for(bk->bk->p);
}

int flush_locked(bool wait = true);
void finish_flight_locked();

// a fresh slot; when the pool is dry, the pending operations (which pin their operands and
// results) are run first.  Throws ApiError if that frees nothing.
int32_t alloc_slot(SlotPool *pool);

// make sure `s` has a device slot holding its current value; returns the slot
int32_t ensure_slot(const LweSample *cs, SlotPool *pool) {
    auto *s = const_cast<LweSample *>(cs);
    bind_pool(cs, pool);
    if (s->slot >= 0) return s->slot;
    if (s->slot == SLOT_ZERO) {
        pool->retain(pool->const_slot[0]);
        s->slot = pool->const_slot[0];
        return s->slot;
    }
    const int32_t slot = alloc_slot(pool);
    Engine::get().write_slot(pool, slot, s->a, s->b);
    s->slot = slot;
    return slot;
}

void drop_slot(LweSample *s) {
    if (s->slot >= 0) header_of(s)->pool->release(s->slot);
    s->slot = SLOT_HOST;
}

// re-point a handle to `slot` (already retained for it), releasing what it held
void repoint(LweSample *s, SlotPool *pool, int32_t slot) {
    bind_pool(s, pool);
    if (s->slot >= 0) pool->release(s->slot);
    s->slot = slot;
}

void begin_op(const TFheGateBootstrappingCloudKeySet *bk) {
    Recorder &r = rec();
    if (r.key && r.key != bk && !r.ops.empty()) flush_locked();
    r.key = bk;
    r.pool = pool_of_key(bk);
    // pending operations pin their slots until they run: flush before the pool runs dry, so
    // arbitrarily long recordings need only bounded device memory
    if (!r.ops.empty() && r.pool->capacity() - r.pool->in_use() < 4096) flush_locked();
}

void sync_sample_locked(const LweSample *cs) {
    Recorder &r = rec();
    auto *s = const_cast<LweSample *>(cs);
    if (s->slot < 0) return;                       // host mirror is authoritative (fresh: trivial 0)
    SlotPool *pool = header_of(s)->pool;
    // pending = written by a recorded operation that has not run (a gate, or a NOT riding on
    // level 0 of an already materialised operand)
    if (pool->pending[s->slot] && !r.ops.empty()) flush_locked();
    Engine::get().read_slot(pool, s->slot, s->a, &s->b);
}

void finish_op(LweSample *result) {
    Recorder &r = rec();
    if (!r.deferred) {
        flush_locked();
        sync_sample_locked(result);   // immediate mode: host mirror valid on return, as upstream
    }
}

void record_gate2_locked(int code, LweSample *result, const LweSample *ca, const LweSample *cb,
                         const TFheGateBootstrappingCloudKeySet *bk);
void record_gate2(int code, LweSample *result, const LweSample *ca, const LweSample *cb,
                  const TFheGateBootstrappingCloudKeySet *bk) {
    guarded([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        record_gate2_locked(code, result, ca, cb, bk);
    });
}
// 0 / 1 if `slot` is one of the pool's shared trivial samples (a public constant), else -1
int const_bit(const SlotPool *pool, int32_t slot) { return slot == pool->const_slot[0] ? 0 : slot == pool->const_slot[1] ? 1 : -1; }
void point_at(LweSample *result, SlotPool *pool, int32_t slot) {        // result becomes (a handle of) `slot`: a COPY
    pool->retain(slot);
    repoint(result, pool, slot);
    finish_op(result);
}
void not_locked(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk);

void record_gate2_locked(int code, LweSample *result, const LweSample *ca, const LweSample *cb,
                         const TFheGateBootstrappingCloudKeySet *bk) {
    Recorder &r = rec();
    begin_op(bk);
    SlotPool *pool = r.pool;
    bind_pool(result, pool);                  // refuse a foreign / mismatched result before anything changes
    const int32_t sa = ensure_slot(ca, pool), sb = ensure_slot(cb, pool);
    if (r.fold_constants) {
        const int ka = const_bit(pool, sa), kb = const_bit(pool, sb);
        if (ka >= 0 || kb >= 0) {
            // the gate as a function of its non-constant operand x: f(0), f(1)
            const uint8_t *tt = GATE_TT[code];
            const int f0 = ka >= 0 ? tt[2 * ka + (kb >= 0 ? kb : 0)] : tt[kb];
            const int f1 = ka >= 0 ? tt[2 * ka + (kb >= 0 ? kb : 1)] : tt[2 + kb];
            ++Engine::get().stats.folded_gates;
            if (f0 == f1) return point_at(result, pool, pool->const_slot[f0]);          // a constant
            if (f0 == 0) return point_at(result, pool, ka >= 0 ? sb : sa);                // x itself
            return not_locked(result, ka >= 0 ? cb : ca, bk);                             // NOT x: linear, no bootstrap
        }
    }
    uint64_t key = 0;
    if (r.reuse_gates) {
        const bool symmetric = GATE_LIN[code].sa == GATE_LIN[code].sb;   // t = c0 + s (A + B)
        key = symmetric && sb < sa ? gate_key(code, sb, sa) : gate_key(code, sa, sb);
        auto it = r.pending_gate.find(key);
        if (it != r.pending_gate.end()) {
            pool->retain(it->second);
            repoint(result, pool, it->second);
            ++Engine::get().stats.reused_gates;
            finish_op(result);
            return;
        }
    }
    const int32_t dst = alloc_slot(pool);     // may flush: levels are read after it
    if (r.reuse_gates) r.pending_gate.emplace(key, dst);
    const int32_t level = 1 + std::max(pool->level[sa], pool->level[sb]);
    pool->level[dst] = level;
    pool->pending[dst] = 1;
    pool->retain(sa); pool->retain(sb); pool->retain(dst);   // pending references
    r.ops.push_back(PendingOp{(uint8_t)code, dst, sa, sb, -1, level});
    r.max_level = std::max(r.max_level, level);
    repoint(result, pool, dst);
    finish_op(result);
}

}  // namespace

// flush: levelise, build descriptors, execute, release pending references
namespace {
int32_t alloc_slot(SlotPool *pool) {
    Recorder &r = rec();
    if (pool->in_use() == pool->capacity()) {
        finish_flight_locked();                          // a completed asynchronous flush still pins its slots
        if (pool->in_use() == pool->capacity() && !r.ops.empty() && r.pool == pool) flush_locked();
    }
    return pool->alloc();
}

// the asynchronous flush (if any) is complete after this: wait for it, release what it pinned
void finish_flight_locked() {
    Recorder &r = rec();
    Engine::get().wait_flight();
    if (r.flight_ops.empty()) return;
    SlotPool *pool = r.flight_pool;
    for (const PendingOp &op : r.flight_ops) {
        pool->release(op.dst);
        pool->release(op.a);
        if (op.b >= 0) pool->release(op.b);
        if (op.c >= 0) pool->release(op.c);
    }
    r.flight_ops.clear();
    r.flight_pool = nullptr;
}

int flush_locked(bool wait) {
    Recorder &r = rec();
    if (r.ops.empty()) { if (wait) finish_flight_locked(); return 0; }
    SlotPool *pool = r.pool;
    if (r.eliminate_dead) {
        // reverse recording order = reverse topological order: dropping a consumer first lets its producers die too
        size_t dead = 0;
        std::vector<uint8_t> is_dead(r.ops.size(), 0);
        for (size_t i = r.ops.size(); i-- > 0;) {
            const PendingOp &op = r.ops[i];
            if (pool->refs(op.dst) != 1) continue;          // a handle or a live operation still holds the result
            pool->level[op.dst] = 0;
            pool->pending[op.dst] = 0;
            pool->release(op.dst);
            pool->release(op.a);
            if (op.b >= 0) pool->release(op.b);
            if (op.c >= 0) pool->release(op.c);
            is_dead[i] = 1;
            ++dead;
        }
        if (dead) {
            size_t w = 0;
            int32_t depth = 0;
            for (size_t i = 0; i < r.ops.size(); ++i)
                if (!is_dead[i]) { depth = std::max(depth, r.ops[i].level); r.ops[w++] = r.ops[i]; }
            r.ops.resize(w);
            r.max_level = depth;
            Engine::get().stats.dead_gates += dead;
            if (r.ops.empty()) {
                r.not_origin.clear(); r.pending_gate.clear(); r.pending_mux.clear(); r.max_level = 0;
                if (wait) finish_flight_locked();
                return 0;
            }
        }
    }
    // level of every op: ASAP, or slack-aware balanced (same depth, fuller narrow levels)
    std::vector<int32_t> lvl, alap;
    const int levels = schedule_levels(r.ops, r.max_level, r.balance_levels, Engine::get().cu_count(), lvl, &alap);
    if (const char *trace = std::getenv("TFHE_HIP_TRACE_DAG")) {      // diagnostic: per op "kind asap alap level dst a b c" (slots)
        if (FILE *f = std::fopen(trace, "w")) {
            for (size_t i = 0; i < r.ops.size(); ++i)
                std::fprintf(f, "%d %d %d %d %d %d %d %d\n", (int)r.ops[i].kind, r.ops[i].level, alap[i], lvl[i],
                             r.ops[i].dst, r.ops[i].a, r.ops[i].b, r.ops[i].c);
            std::fclose(f);
        }
    }
    LevelPlan plan;
    plan.levels = levels;
    // counting sort by level: gates of level L (1-based) in group L - 1, NOTs riding on level L (0 = inputs) in group L
    const size_t ngroups = (size_t)levels, nnotgroups = (size_t)levels + 1;
    std::vector<int32_t> nrot(ngroups + 1, 0), nks(ngroups + 1, 0), nnot(nnotgroups + 1, 0);
    for (size_t i = 0; i < r.ops.size(); ++i) {
        const PendingOp &op = r.ops[i];
        if (op.kind == OP_NOT) { ++nnot[(size_t)lvl[i]]; continue; }
        const size_t g = (size_t)(lvl[i] - 1);
        nrot[g] += op.kind == OP_MUX ? 2 : 1;
        ++nks[g];
    }
    plan.rot_off.assign(ngroups + 1, 0);
    plan.ks_off.assign(ngroups + 1, 0);
    plan.not_off.assign(nnotgroups + 1, 0);
    plan.max_rots = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        plan.rot_off[g + 1] = plan.rot_off[g] + nrot[g];
        plan.ks_off[g + 1] = plan.ks_off[g] + nks[g];
        plan.max_rots = std::max(plan.max_rots, nrot[g]);
    }
    for (size_t g = 0; g < nnotgroups; ++g) plan.not_off[g + 1] = plan.not_off[g] + nnot[g];
    plan.rots.resize(plan.rot_off[ngroups]);
    plan.kss.resize(plan.ks_off[ngroups]);
    plan.nots.resize(plan.not_off[nnotgroups]);
    std::vector<int32_t> rpos(plan.rot_off.begin(), plan.rot_off.end() - 1);   // cursor per group
    std::vector<int32_t> kpos(plan.ks_off.begin(), plan.ks_off.end() - 1);
    std::vector<int32_t> npos(plan.not_off.begin(), plan.not_off.end() - 1);
    const int32_t mu = 1 << 29;
    for (size_t i = 0; i < r.ops.size(); ++i) {
        const PendingOp &op = r.ops[i];
        if (op.kind == OP_NOT) {
            plan.nots[npos[(size_t)lvl[i]]++] = NotDesc{op.a, op.dst};
            continue;
        }
        const size_t g = (size_t)(lvl[i] - 1);
        const int32_t base = plan.rot_off[g];
        if (op.kind == OP_MUX) {
            // tfhe bootsMUX: u1 = BR(-1/8 + a + b), u2 = BR(-1/8 - a + c), KS(u1 + u2 + 1/8)
            const int32_t i0 = rpos[g]++, i1 = rpos[g]++;
            plan.rots[i0] = RotDesc{op.a, op.b, 1, 1, -(mu), i0 - base};
            plan.rots[i1] = RotDesc{op.a, op.c, -1, 1, -(mu), i1 - base};
            plan.kss[kpos[g]++] = KsDesc{i0 - base, i1 - base, mu, op.dst};
        } else {
            const GateLin &gl = GATE_LIN[op.kind];
            const int32_t i0 = rpos[g]++;
            plan.rots[i0] = RotDesc{op.a, op.b, gl.sa, gl.sb, gl.c8 * mu, i0 - base};
            plan.kss[kpos[g]++] = KsDesc{i0 - base, -1, 0, op.dst};
        }
    }
    // (execute() first waits for a flush still in flight: everything above -- elimination, levelling, the plan -- ran
    // while the device was busy with it)
    Engine::get().execute(r.key->bk->dev, pool, std::move(plan), wait);
    {
        // the PREVIOUS asynchronous flush is complete now (execute waited for it): release what it pinned
        if (!r.flight_ops.empty()) {
            for (const PendingOp &op : r.flight_ops) {
                r.flight_pool->release(op.dst);
                r.flight_pool->release(op.a);
                if (op.b >= 0) r.flight_pool->release(op.b);
                if (op.c >= 0) r.flight_pool->release(op.c);
            }
            r.flight_ops.clear();
            r.flight_pool = nullptr;
        }
    }
    for (const PendingOp &op : r.ops) {
        pool->level[op.dst] = 0;          // a later recording reads these slots as inputs: the stream orders it behind
        pool->pending[op.dst] = 0;
        if (wait) {
            pool->release(op.dst);
            pool->release(op.a);
            if (op.b >= 0) pool->release(op.b);
            if (op.c >= 0) pool->release(op.c);
        }
    }
    if (!wait) { r.flight_ops.swap(r.ops); r.flight_pool = pool; }
    r.ops.clear();
    r.not_origin.clear();
    r.pending_gate.clear();
    r.pending_mux.clear();
    r.max_level = 0;
    return levels;
}
}  // namespace

// ===========================================================================
// upstream-compatible C ABI
// ===========================================================================
extern "C" {

int32_t modSwitchFromTorus32(Torus32 phase, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + interv / 2;
    return (int32_t)(phase64 / interv);
}
Torus32 modSwitchToTorus32(int32_t mu, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    return (Torus32)(((uint64_t)(int64_t)mu * interv) >> 32);
}

TFheGateBootstrappingParameterSet *new_default_gate_bootstrapping_parameters(int32_t minimum_lambda) {
    Params p;
    if (!default_params(minimum_lambda, p)) {
        std::fprintf(stderr, "Sorry, for now, the parameters are only implemented for 80bit and 128bit of security!\n");
        set_error("unsupported minimum_lambda");
        return nullptr;
    }
    return &make_param_bundle(p)->set;
}

void delete_gate_bootstrapping_parameters(TFheGateBootstrappingParameterSet *params) {
    delete reinterpret_cast<ParamBundle *>(params);
}

// seed == nullptr: secrets and masks from two ChaCha20 streams keyed by the OS; else the seeded generator (fixtures)
static TFheGateBootstrappingSecretKeySet *make_keyset(const TFheGateBootstrappingParameterSet *params, const uint64_t *seed,
                                                      bool device) {
    if (!params) { set_error("null params"); return nullptr; }
    const Params &p = params_of(params);
    auto *sk = new TfheHipSecretKey();
    auto *ck = new TfheHipCloudKey();
    if (seed) {
        Rng both(*seed);
        generate_keys(p, both, both, *sk, *ck);
    } else {
        Rng secret = Rng::secure(), mask = Rng::secure();
        generate_keys(p, secret, mask, *sk, *ck);
    }
    if (device) {
        try {
            ck->dev = Engine::get().upload_key(*ck);
        } catch (const ApiError &e) {       // a parameter set the kernels cannot run exactly: no keyset, and why
            set_error(e.msg);
            delete sk;
            delete ck;
            return nullptr;
        }
    }
    auto *ks = new TFheGateBootstrappingSecretKeySet();
    ks->params = params;
    ks->lwe_key = sk;
    ks->tgsw_key = sk;
    ks->cloud.params = params;
    ks->cloud.bk = ck;
    ks->cloud.bkFFT = ck;
    return ks;
}

TFheGateBootstrappingSecretKeySet *new_random_gate_bootstrapping_secret_keyset(const TFheGateBootstrappingParameterSet *params) {
    // upstream draws from its global generator; here every keyset comes from ChaCha20 streams keyed with 256 bits
    // of OS entropy each (reproducible keys: tfhe_hip_new_secret_keyset_seeded)
    return make_keyset(params, nullptr, true);
}

void delete_gate_bootstrapping_secret_keyset(TFheGateBootstrappingSecretKeySet *keyset) {
    if (!keyset) return;
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    if (r.key == &keyset->cloud) { flush_locked(); r.key = nullptr; }
    if (keyset->cloud.bk) {
        Engine::get().free_key(keyset->cloud.bk->dev);
        delete keyset->cloud.bk;
    }
    delete keyset->lwe_key;
    if (io_forget_owned_secret(keyset)) delete_gate_bootstrapping_parameters(const_cast<TFheGateBootstrappingParameterSet *>(keyset->params));
    delete keyset;
}

void delete_gate_bootstrapping_cloud_keyset(TFheGateBootstrappingCloudKeySet *keyset) {
    // A cloud keyset embedded in a secret keyset (&key->cloud, main.cpp:23) goes with its owner;
    // only one created by new_tfheGateBootstrappingCloudKeySet_fromFile is released here.
    if (!keyset || !io_forget_owned_cloud(keyset)) return;
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    if (r.key == keyset) { flush_locked(); r.key = nullptr; }
    if (keyset->bk) {
        Engine::get().free_key(keyset->bk->dev);
        delete keyset->bk;
    }
    delete_gate_bootstrapping_parameters(const_cast<TFheGateBootstrappingParameterSet *>(keyset->params));
    delete keyset;
}

LweSample *new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet *params) {
    if (nbelems < 0 || !params) { set_error("bad arguments to new_gate_bootstrapping_ciphertext_array"); return nullptr; }
    const int32_t n = params->in_out_params->n;
    const size_t mirror = (size_t)(n + MIRROR_PREFIX_WORDS + 1) & ~(size_t)1;   // words per sample, pointer-aligned
    const size_t bytes = sizeof(ArrayHeader) + (size_t)nbelems * sizeof(LweSample) + (size_t)nbelems * mirror * sizeof(Torus32);
    char *mem = static_cast<char *>(std::calloc(1, bytes));
    if (!mem) fatal("out of host memory");
    auto *h = reinterpret_cast<ArrayHeader *>(mem);
    h->magic = ARRAY_MAGIC; h->count = nbelems; h->n = n; h->pool = nullptr;
    auto *samples = reinterpret_cast<LweSample *>(mem + sizeof(ArrayHeader));
    auto *words = reinterpret_cast<Torus32 *>(mem + sizeof(ArrayHeader) + (size_t)nbelems * sizeof(LweSample));
    for (int32_t i = 0; i < nbelems; ++i) {
        samples[i].a = words + (size_t)i * mirror + MIRROR_PREFIX_WORDS;
        std::memcpy(samples[i].a - MIRROR_PREFIX_WORDS, &h, sizeof h);
        std::memcpy(samples[i].a - 2, &MIRROR_MARKER, sizeof MIRROR_MARKER);
        samples[i].b = -(1 << 29);      // fresh = trivial encryption of 0, like bootsCONSTANT(.., 0)
        samples[i].slot = SLOT_ZERO;
        samples[i].current_variance = 0.0;
    }
    return samples;
}

void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample *samples) {
    if (!samples) return;
    auto *h = reinterpret_cast<ArrayHeader *>(reinterpret_cast<char *>(samples) - sizeof(ArrayHeader));
    if (h->magic != ARRAY_MAGIC) {      // not ours (or already freed): report, leak rather than abort
        set_error("delete_gate_bootstrapping_ciphertext_array: not an array base pointer");
        return;
    }
    if (h->count != nbelems) set_error("delete_gate_bootstrapping_ciphertext_array: count differs from allocation");
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    for (int32_t i = 0; i < h->count; ++i)
        if (samples[i].slot >= 0) h->pool->release(samples[i].slot);
    h->magic = 0;
    std::free(h);
}

LweSample *new_gate_bootstrapping_ciphertext(const TFheGateBootstrappingParameterSet *params) {
    return new_gate_bootstrapping_ciphertext_array(1, params);
}
void delete_gate_bootstrapping_ciphertext(LweSample *sample) { delete_gate_bootstrapping_ciphertext_array(1, sample); }

void bootsSymEncrypt(LweSample *result, int32_t message, const TFheGateBootstrappingSecretKeySet *key) {
    guarded([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        drop_slot(result);
        encrypt_bit(*key->lwe_key, r.enc_secret, r.enc_seeded ? r.enc_secret : r.enc_mask, message, result->a, &result->b);
    });
}

int32_t bootsSymDecrypt(const LweSample *sample, const TFheGateBootstrappingSecretKeySet *key) {
    int32_t bit = 0;
    guarded([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        sync_sample_locked(sample);
        bit = phase_of(*key->lwe_key, sample->a, sample->b) > 0 ? 1 : 0;
    });
    return bit;
}

void bootsCONSTANT(LweSample *result, int32_t value, const TFheGateBootstrappingCloudKeySet *bk) {
    guarded([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        begin_op(bk);
        bind_pool(result, r.pool);          // refuse before anything changes
        const int32_t s = r.pool->const_slot[value ? 1 : 0];
        r.pool->retain(s);
        repoint(result, r.pool, s);
        if (!r.deferred) {   // keep the host mirror exact without a device round trip
            std::memset(result->a, 0, (size_t)bk->params->in_out_params->n * sizeof(Torus32));
            result->b = value ? (1 << 29) : -(1 << 29);
        }
    });
}

void bootsCOPY(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {
    guarded([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        begin_op(bk);
        if (result == ca) return;
        bind_pool(result, r.pool);
        const int32_t s = ensure_slot(ca, r.pool);
        r.pool->retain(s);
        repoint(result, r.pool, s);
        if (!r.deferred) sync_sample_locked(result);
    });
}

void bootsNOT(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {
    guarded([&] {
        std::lock_guard<std::recursive_mutex> g(rec().mtx);
        not_locked(result, ca, bk);
    });
}
}  // extern "C"
namespace {
void not_locked(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {
    Recorder &r = rec();
    begin_op(bk);
    SlotPool *pool = r.pool;
    bind_pool(result, pool);
    const int32_t sa = ensure_slot(ca, pool);
    if (r.fold_constants && const_bit(pool, sa) >= 0)             // NOT of a public constant is the other constant
        return point_at(result, pool, pool->const_slot[1 - const_bit(pool, sa)]);
    {   // NOT of a still-pending NOT: -(-x) = x exactly, so alias the original operand; two NOTs
        // of one level would otherwise sit in the same launch and race
        auto it = r.not_origin.find(sa);
        if (it != r.not_origin.end()) {
            pool->retain(it->second);
            repoint(result, pool, it->second);
            finish_op(result);
            return;
        }
    }
    if (r.reuse_gates) {
        auto it = r.pending_gate.find(gate_key(OP_NOT, sa, 0));
        if (it != r.pending_gate.end()) {
            pool->retain(it->second);
            repoint(result, pool, it->second);
            ++Engine::get().stats.reused_gates;
            finish_op(result);
            return;
        }
    }
    const int32_t dst = alloc_slot(pool);      // may flush: the level is read after it
    if (r.reuse_gates) r.pending_gate.emplace(gate_key(OP_NOT, sa, 0), dst);
    const int32_t level = pool->level[sa];     // linear: rides on its operand's level
    pool->level[dst] = level;
    pool->pending[dst] = 1;                    // pending even at level 0 (NOT of a materialised sample)
    pool->retain(sa); pool->retain(dst);
    r.ops.push_back(PendingOp{OP_NOT, dst, sa, -1, -1, level});
    r.not_origin.emplace(dst, sa);
    r.max_level = std::max(r.max_level, level);
    repoint(result, pool, dst);
    finish_op(result);
}
}  // namespace
extern "C" {

void bootsNAND(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_NAND, r_, a, b, bk); }
void bootsOR(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_OR, r_, a, b, bk); }
void bootsAND(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_AND, r_, a, b, bk); }
void bootsNOR(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_NOR, r_, a, b, bk); }
void bootsXOR(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_XOR, r_, a, b, bk); }
void bootsXNOR(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_XNOR, r_, a, b, bk); }
void bootsANDNY(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_ANDNY, r_, a, b, bk); }
void bootsANDYN(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_ANDYN, r_, a, b, bk); }
void bootsORNY(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_ORNY, r_, a, b, bk); }
void bootsORYN(LweSample *r_, const LweSample *a, const LweSample *b, const TFheGateBootstrappingCloudKeySet *bk) { record_gate2(TFHE_HIP_ORYN, r_, a, b, bk); }

static void mux_locked(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c,
                       const TFheGateBootstrappingCloudKeySet *bk);
void bootsMUX(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c,
              const TFheGateBootstrappingCloudKeySet *bk) {
    guarded([&] {
        std::lock_guard<std::recursive_mutex> g(rec().mtx);
        mux_locked(result, a, b, c, bk);
    });
}
static void mux_locked(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c,
                       const TFheGateBootstrappingCloudKeySet *bk) {
    Recorder &r = rec();
    begin_op(bk);
    SlotPool *pool = r.pool;
    bind_pool(result, pool);
    const int32_t sa = ensure_slot(a, pool), sb = ensure_slot(b, pool), sc = ensure_slot(c, pool);
    if (r.fold_constants) {
        const int ka = const_bit(pool, sa), kb = const_bit(pool, sb), kc = const_bit(pool, sc);
        if (ka >= 0 || kb >= 0 || kc >= 0 || sb == sc) {
            ++Engine::get().stats.folded_gates;
            if (ka >= 0) return point_at(result, pool, ka ? sb : sc);                     // a constant selector picks an operand
            if (sb == sc) return point_at(result, pool, sb);                                // both data operands the same sample
            if (kb >= 0 && kc >= 0) {                                                       // (kb != kc here)  MUX(a, 1, 0) = a, MUX(a, 0, 1) = NOT a
                if (kb == 1) return point_at(result, pool, sa);
                return not_locked(result, a, bk);
            }
            // one constant data operand: a two-input gate, one blind rotation instead of two
            if (kc >= 0) return record_gate2_locked(kc == 0 ? TFHE_HIP_AND : TFHE_HIP_ORNY, result, a, b, bk);   // a & b | !a | b
            return record_gate2_locked(kb == 0 ? TFHE_HIP_ANDNY : TFHE_HIP_OR, result, a, c, bk);               // !a & c | a | c
        }
    }
    if (r.reuse_gates) {
        auto it = r.pending_mux.find({sa, sb, sc});
        if (it != r.pending_mux.end()) {
            pool->retain(it->second);
            repoint(result, pool, it->second);
            ++Engine::get().stats.reused_gates;
            finish_op(result);
            return;
        }
    }
    const int32_t dst = alloc_slot(pool);
    if (r.reuse_gates) r.pending_mux.emplace(std::array<int32_t, 3>{sa, sb, sc}, dst);
    const int32_t level = 1 + std::max(pool->level[sa], std::max(pool->level[sb], pool->level[sc]));
    pool->level[dst] = level;
    pool->pending[dst] = 1;
    pool->retain(sa); pool->retain(sb); pool->retain(sc); pool->retain(dst);
    r.ops.push_back(PendingOp{OP_MUX, dst, sa, sb, sc, level});
    r.max_level = std::max(r.max_level, level);
    repoint(result, pool, dst);
    finish_op(result);
}

// ===========================================================================
// tfhe_hip_* extensions
// ===========================================================================
const char *tfhe_hip_last_error(void) { return last_error_ref().c_str(); }
void tfhe_hip_clear_error(void) { set_error(""); }

int tfhe_hip_set_device(int device) { Engine::get().set_device(device); return 0; }
int tfhe_hip_get_device(void) { return Engine::get().device(); }

TFheGateBootstrappingParameterSet *tfhe_hip_new_parameters(int32_t n, int32_t N, int32_t k, int32_t l, int32_t Bgbit,
                                                           int32_t ks_t, int32_t ks_basebit, double ks_stdev,
                                                           double bk_stdev, double max_stdev) {
    if (n <= 0 || N <= 0 || (N & (N - 1)) || k < 1 || l < 1 || Bgbit < 1 || l * Bgbit > 32 || ks_t < 1 ||
        ks_basebit < 1 || ks_t * ks_basebit > 31) {
        set_error("tfhe_hip_new_parameters: invalid parameter tuple");
        return nullptr;
    }
    Params p{n, N, k, l, Bgbit, ks_t, ks_basebit, ks_stdev, bk_stdev, max_stdev};
    return &make_param_bundle(p)->set;
}
TFheGateBootstrappingParameterSet *tfhe_hip_new_p2048_parameters(void) { return &make_param_bundle(p2048_params())->set; }

TFheGateBootstrappingSecretKeySet *tfhe_hip_new_secret_keyset_seeded(const TFheGateBootstrappingParameterSet *params, uint64_t seed) {
    return make_keyset(params, &seed, true);
}
TFheGateBootstrappingSecretKeySet *tfhe_hip_new_secret_keyset_seeded_host(const TFheGateBootstrappingParameterSet *params, uint64_t seed) {
    return make_keyset(params, &seed, false);
}
void tfhe_hip_test_chacha20_block(const uint32_t *key8, uint64_t counter, const uint32_t *nonce2, uint32_t *out16) {
    Rng::chacha_block(key8, counter, nonce2, out16);
}
int tfhe_hip_randomness_is_seeded(void) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    return r.enc_seeded ? 1 : 0;
}
void tfhe_hip_set_encrypt_seed(uint64_t seed) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    r.enc_secret.reseed(seed);
    r.enc_seeded = true;
}

const int32_t *tfhe_hip_key_lwe(const TFheGateBootstrappingSecretKeySet *key, int64_t *count) {
    if (count) *count = (int64_t)key->lwe_key->lwe_key.size();
    return key->lwe_key->lwe_key.data();
}
const int32_t *tfhe_hip_key_tlwe(const TFheGateBootstrappingSecretKeySet *key, int64_t *count) {
    if (count) *count = (int64_t)key->lwe_key->tlwe_key.size();
    return key->lwe_key->tlwe_key.data();
}
const Torus32 *tfhe_hip_key_bk(const TFheGateBootstrappingCloudKeySet *cloud, int64_t *count) {
    if (count) *count = (int64_t)cloud->bk->bk.size();
    return cloud->bk->bk.data();
}
const Torus32 *tfhe_hip_key_ksk(const TFheGateBootstrappingCloudKeySet *cloud, int64_t *count) {
    if (count) *count = (int64_t)cloud->bk->ksk.size();
    return cloud->bk->ksk.data();
}

int32_t tfhe_hip_sample_words(const TFheGateBootstrappingParameterSet *params) { return params->in_out_params->n + 1; }

// The pool a group of samples lives in (or will live in): the one their array is already
// bound to, else the pool of this parameter set's ciphertext shape if the engine has one, else --
// only when `create` (a device-side transfer needs one) -- a new pool.  nullptr = no device
// state for this shape yet: the host mirrors are authoritative.
static SlotPool *resolve_pool(const LweSample *samples, const TFheGateBootstrappingParameterSet *params, bool create) {
    if (!params || !params->in_out_params) api_fail("null parameter set");
    const Params &p = params_of(params);
    SlotPool *bound = header_of(samples)->pool;
    if (bound) {
        if (bound->ct_words() != p.ct_words())
            api_fail("samples belong to a parameter set of LWE dimension " + std::to_string(bound->ct_words() - 1) +
                     ", not " + std::to_string(p.n));
        return bound;
    }
    if (SlotPool *pl = Engine::get().find_pool(p)) return pl;
    return create ? Engine::get().pool_for(p) : nullptr;
}

static int export_impl(const LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                       Torus32 *out, bool device_dst, bool wait = true) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    if (count <= 0) return 0;
    const int32_t n = params_of(params).n;
    SlotPool *pool = resolve_pool(samples, params, device_dst);
    bool any_dev = false;
    for (int32_t i = 0; i < count; ++i) any_dev |= samples[i].slot >= 0;
    if (!device_dst && (!any_dev || !pool)) {
        for (int32_t i = 0; i < count; ++i) {
            if (samples[i].slot == SLOT_ZERO) {          // fresh: trivial encryption of 0
                std::memset(out + (size_t)i * (n + 1), 0, (size_t)n * 4);
                out[(size_t)i * (n + 1) + n] = -(1 << 29);
                continue;
            }
            std::memcpy(out + (size_t)i * (n + 1), samples[i].a, (size_t)n * 4);
            out[(size_t)i * (n + 1) + n] = samples[i].b;
        }
        return 0;
    }
    // the stream-ordered form (wait == false) only ENQUEUES the pending gates' launches: the gather below, and whatever
    // the caller puts on the stream after it (a collective), follow them in stream order with no host wait (ADVICE r4:
    // a synchronous flush here made every rank of a sharded match sit out its whole local phase on the host)
    if (!r.ops.empty()) flush_locked(wait);
    std::vector<int32_t> slots(count);
    for (int32_t i = 0; i < count; ++i) slots[i] = ensure_slot(&samples[i], pool);
    Engine::get().read_slots_packed(pool, slots.data(), count, out, device_dst, wait);
    return 0;
}

static int import_impl(LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                       const Torus32 *in, bool device_src, bool wait = true) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    if (count <= 0) return 0;
    const int32_t n = params_of(params).n;
    SlotPool *pool = resolve_pool(samples, params, device_src);
    if (!pool) {                                         // no device state for this shape: host mirrors only
        for (int32_t i = 0; i < count; ++i) {
            std::memcpy(samples[i].a, in + (size_t)i * (n + 1), (size_t)n * 4);
            samples[i].b = in[(size_t)i * (n + 1) + n];
            samples[i].slot = SLOT_HOST;
        }
        return 0;
    }
    for (int32_t i = 0; i < count; ++i) bind_pool(&samples[i], pool);    // refuse before anything changes
    std::vector<int32_t> slots(count);
    for (int32_t i = 0; i < count; ++i) {
        repoint(&samples[i], pool, alloc_slot(pool));
        slots[i] = samples[i].slot;
        if (!device_src) {
            std::memcpy(samples[i].a, in + (size_t)i * (n + 1), (size_t)n * 4);
            samples[i].b = in[(size_t)i * (n + 1) + n];
        }
    }
    Engine::get().write_slots_packed(pool, slots.data(), count, in, device_src, wait);
    return 0;
}

int tfhe_hip_export_samples(const LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                            Torus32 *out_words) {
    return guarded_rc([&] { return export_impl(samples, count, params, out_words, false); });
}
int tfhe_hip_import_samples(LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                            const Torus32 *in_words) {
    return guarded_rc([&] { return import_impl(samples, count, params, in_words, false); });
}
int tfhe_hip_export_samples_device(const LweSample *samples, int32_t count,
                                   const TFheGateBootstrappingParameterSet *params, void *device_words) {
    return guarded_rc([&] { return export_impl(samples, count, params, static_cast<Torus32 *>(device_words), true); });
}
int tfhe_hip_import_samples_device(LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                                   const void *device_words) {
    return guarded_rc([&] { return import_impl(samples, count, params, static_cast<const Torus32 *>(device_words), true); });
}

int tfhe_hip_export_samples_device_async(const LweSample *samples, int32_t count,
                                         const TFheGateBootstrappingParameterSet *params, void *device_words) {
    return guarded_rc([&] { return export_impl(samples, count, params, static_cast<Torus32 *>(device_words), true, false); });
}
int tfhe_hip_import_samples_device_async(LweSample *samples, int32_t count, const TFheGateBootstrappingParameterSet *params,
                                         const void *device_words) {
    return guarded_rc([&] { return import_impl(samples, count, params, static_cast<const Torus32 *>(device_words), true, false); });
}
void *tfhe_hip_stream(void) {
    Engine::get().ensure_init();
    return static_cast<void *>(Engine::get().stream());
}

int tfhe_hip_sync_samples(const LweSample *samples, int32_t count) {
    return guarded_rc([&] {
        Recorder &r = rec();
        std::lock_guard<std::recursive_mutex> g(r.mtx);
        if (!r.ops.empty()) flush_locked();
        for (int32_t i = 0; i < count; ++i) sync_sample_locked(&samples[i]);
        return 0;
    });
}

void tfhe_hip_set_deferred(int on) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    if (!on && !r.ops.empty()) guarded([&] { flush_locked(); });
    r.deferred = on != 0;
}
int tfhe_hip_get_deferred(void) { return rec().deferred ? 1 : 0; }

int tfhe_hip_flush(void) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    return guarded_rc([&] { return flush_locked(); });
}
int tfhe_hip_flush_async(void) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    return guarded_rc([&] { return flush_locked(false); });
}
int tfhe_hip_wait(void) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    // the flush in flight, and the stream-ordered transfers nobody has waited for (an import behind a collective)
    return guarded_rc([&] { finish_flight_locked(); Engine::get().sync_io(); return 0; });
}
int tfhe_hip_stream_sync(void) {
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    return guarded_rc([&] {
        Engine::get().ensure_init();
        finish_flight_locked();
        Engine::get().sync_stream("tfhe_hip_stream_sync");
        return 0;
    });
}
int tfhe_hip_wait_event(void *event, const char *what) {
    if (!event) { set_error("tfhe_hip_wait_event: null event"); return -1; }
    // the deadline and the label are read under the recorder lock (tfhe_hip_set_tuning / tfhe_hip_set_diag_label write them
    // there); the wait itself holds no lock -- it may last until a peer process arrives, and other threads go on recording
    long long deadline_ms;
    std::string label;
    {
        std::lock_guard<std::recursive_mutex> g(rec().mtx);
        Engine::get().ensure_init();
        deadline_ms = Engine::get().sync_deadline_ms;
        label = Engine::get().diag_label;
    }
    return guarded_rc([&] {
        Engine::get().wait_event(static_cast<hipEvent_t>(event), what && *what ? what : "tfhe_hip_wait_event", deadline_ms, label);
        return 0;
    });
}
int tfhe_hip_device_pci_bus_id(char *out, int len) {
    if (!out || len < 16) { set_error("tfhe_hip_device_pci_bus_id: buffer of at least 16 bytes needed"); return -1; }
    return guarded_rc([&] {
        Engine::get().ensure_init();
        return Engine::get().pci_bus_id(out, len) ? 0 : -1;
    });
}
void tfhe_hip_set_diag_label(const char *label) {
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    Engine::get().diag_label = label ? std::string(label).substr(0, 127) : std::string();
}

int tfhe_hip_gate_batch(int gate, LweSample *result, const LweSample *a, const LweSample *b, int32_t count,
                        const TFheGateBootstrappingCloudKeySet *bk) {
    if (gate < 0 || gate > TFHE_HIP_ORYN) { set_error("tfhe_hip_gate_batch: bad gate code"); return -1; }
    Recorder &r = rec();
    std::lock_guard<std::recursive_mutex> g(r.mtx);
    const bool was = r.deferred;
    r.deferred = true;
    const int rc = guarded_rc([&] {
        for (int32_t i = 0; i < count; ++i) record_gate2_locked(gate, &result[i], &a[i], &b[i], bk);
        return 0;
    });
    r.deferred = was;
    if (!was) flush_locked();      // gates recorded before a refused one still run
    return rc;
}

void tfhe_hip_test_set_alloc_cap(int64_t bytes) {
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    set_alloc_cap((long long)bytes);
}

int tfhe_hip_test_form_admissible(int form, int32_t N, int32_t l, int32_t Bgbit, int tables) {
    return br_form_admissible(form, N, l, Bgbit, tables) ? 1 : 0;
}

int tfhe_hip_set_tuning(const char *name, int64_t value) {
    std::lock_guard<std::recursive_mutex> g(rec().mtx);      // the launchers read the tunings under the same lock (flushes)
    if (name && std::strcmp(name, "ks_tile") == 0) {
        if (value != 0 && value != 16 && value != 24 && value != 32) { set_error("ks_tile must be 0, 16, 24 or 32"); return -1; }
        Engine::get().ks_tile = (int)value;
        return 0;
    }
    if (name && std::strcmp(name, "ks_index") == 0) { Engine::get().ks_index = value != 0; return 0; }
    if (name && std::strcmp(name, "br_digit_table") == 0) { Engine::get().br_digit_table = (int)value; return 0; }
    if (name && std::strcmp(name, "br8_max_rotations") == 0) { Engine::get().br8_max_rotations = (int)value; return 0; }
    if (name && std::strcmp(name, "br_tail8") == 0) { Engine::get().br_tail8 = (int)value; return 0; }
    if (name && std::strcmp(name, "br_variant") == 0) { Engine::get().br_variant = (int)value; return 0; }
    if (name && std::strcmp(name, "reuse_gates") == 0) { rec().reuse_gates = value != 0; return 0; }
    if (name && std::strcmp(name, "eliminate_dead") == 0) { rec().eliminate_dead = value != 0; return 0; }
    if (name && std::strcmp(name, "fold_constants") == 0) { rec().fold_constants = value != 0; return 0; }
    if (name && std::strcmp(name, "balance_levels") == 0) { rec().balance_levels = value != 0; return 0; }
    if (name && std::strcmp(name, "sync_deadline_ms") == 0) { Engine::get().sync_deadline_ms = value > 0 ? (long long)value : 0; return 0; }
    set_error(std::string("tfhe_hip_set_tuning: unknown name ") + (name ? name : "(null)"));
    return -1;
}

// the statistics are only ever written under the recorder lock (flushes, reuse counters, test paths)
void tfhe_hip_get_stats(TfheHipStats *out) {
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    Engine::get().wait_flight();          // the times of an asynchronous flush are read when it has completed
    if (out) *out = Engine::get().stats;
}
void tfhe_hip_reset_stats(void) {
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    Engine::get().stats = TfheHipStats{};
}
void tfhe_hip_set_kernel_timing(int on) { Engine::get().kernel_timing = on != 0; }

static int test_build_ops(const int32_t *ops5, int32_t count, std::vector<PendingOp> &ops) {
    // ops5[i] = {kind, dst, a, b, c}; ASAP levels are derived here exactly as the recorder derives them
    ops.resize((size_t)count);
    std::vector<int32_t> slot_level;
    int depth = 0;
    auto level_of = [&](int32_t s) { return s >= 0 && (size_t)s < slot_level.size() ? slot_level[s] : 0; };
    for (int32_t i = 0; i < count; ++i) {
        const int32_t *o = ops5 + 5 * (size_t)i;
        PendingOp op{(uint8_t)o[0], o[1], o[2], o[3], o[4], 0};
        const int32_t in = std::max(level_of(op.a), std::max(level_of(op.b), level_of(op.c)));
        op.level = op.kind == OP_NOT ? in : in + 1;
        if ((size_t)op.dst >= slot_level.size()) slot_level.resize((size_t)op.dst + 1, 0);
        slot_level[op.dst] = op.level;
        depth = std::max(depth, op.level);
        ops[i] = op;
    }
    return depth;
}

int tfhe_hip_test_schedule(const int32_t *ops5, int32_t count, int32_t unit, int32_t balance, int32_t *levels_out) {
    std::vector<PendingOp> ops;
    const int depth = test_build_ops(ops5, count, ops);
    std::vector<int32_t> lvl;
    const int d = schedule_levels(ops, depth, balance != 0, unit, lvl);
    for (int32_t i = 0; i < count; ++i) levels_out[i] = lvl[i];
    return d;
}


int tfhe_hip_test_wg_times(const TFheGateBootstrappingCloudKeySet *bk, int32_t width, uint64_t *times4, double *launch_ms) {
    if (!bk || !bk->bk || !times4 || width <= 0) { set_error("wg_times: bad arguments"); return -1; }
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    pool_of_key(bk);
    const double ms = Engine::get().run_wg_times(bk->bk->dev, width, reinterpret_cast<unsigned long long *>(times4));
    if (launch_ms) *launch_ms = ms;
    if (ms < 0) { set_error("wg_times: the launched kernel form wrote no stamps"); return -1; }
    return 0;
}

int tfhe_hip_kernel_negacyclic(const TFheGateBootstrappingCloudKeySet *bk, const int32_t *ip, const Torus32 *tp,
                               Torus32 *res, int32_t count) {
    if (!bk || !bk->bk) { set_error("negacyclic: null keyset"); return -1; }
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    pool_of_key(bk);
    Engine::get().run_negacyclic(bk->bk->dev, ip, tp, res, count);
    return 0;
}
int tfhe_hip_kernel_bootstrap_woks(const TFheGateBootstrappingCloudKeySet *bk, const Torus32 *lin, int32_t count,
                                   Torus32 *u_out, Torus32 *acc_out) {
    if (!bk || !bk->bk) { set_error("bootstrap_woks: null keyset"); return -1; }
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    pool_of_key(bk);
    Engine::get().run_bootstrap_woks(bk->bk->dev, lin, count, u_out, acc_out);
    return 0;
}
int tfhe_hip_kernel_keyswitch(const TFheGateBootstrappingCloudKeySet *bk, const Torus32 *u, int32_t count, Torus32 *out) {
    if (!bk || !bk->bk) { set_error("keyswitch: null keyset"); return -1; }
    std::lock_guard<std::recursive_mutex> g(rec().mtx);
    pool_of_key(bk);
    Engine::get().run_keyswitch(bk->bk->dev, u, count, out);
    return 0;
}

}  // extern "C"
