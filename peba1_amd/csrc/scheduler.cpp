// scheduler.cpp -- see scheduler.hpp.  Host logic only.
#include "scheduler.hpp"

#include <algorithm>
#include <queue>
#include <unordered_map>

namespace tfhe_hip {

namespace {
struct HeapItem {
    int32_t alap, idx;
    bool operator>(const HeapItem &o) const { return alap != o.alap ? alap > o.alap : idx > o.idx; }
};
}  // namespace


int schedule_levels(const std::vector<PendingOp> &ops, int asap_depth, bool balance, int unit,
                    std::vector<int32_t> &lvl, std::vector<int32_t> *alap_out) {
    const int n = (int)ops.size();
    lvl.resize(n);
    for (int i = 0; i < n; ++i) lvl[i] = ops[i].level;
    if (alap_out) *alap_out = lvl;
    if (!balance || asap_depth <= 2 || n < 4 * unit) return asap_depth;

    // producer of each pending slot (destinations are unique: SSA; slot ids are pool indices, so a flat
    // table replaces the hash map that used to cost a third of a match's scheduling time)
    int32_t max_slot = -1;
    for (int i = 0; i < n; ++i) max_slot = std::max(max_slot, ops[i].dst);
    std::vector<int32_t> producer((size_t)max_slot + 1, -1);
    for (int i = 0; i < n; ++i)
        if (producer[ops[i].dst] < 0) producer[ops[i].dst] = i;       // (first writer, as emplace kept it)

    // predecessor lists (<= 3 each) and successor lists in CSR form
    std::vector<int32_t> pred(3 * (size_t)n, -1), npred(n, 0), succ_off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        const int32_t src[3] = {ops[i].a, ops[i].b, ops[i].c};
        for (int s = 0; s < 3; ++s) {
            if (src[s] < 0) continue;
            const int32_t pr = src[s] <= max_slot ? producer[src[s]] : -1;
            if (pr < 0 || pr >= i) continue;                         // materialised before this flush
            bool dup = false;
            for (int t = 0; t < npred[i]; ++t) dup |= pred[3 * (size_t)i + t] == pr;
            if (dup) continue;
            pred[3 * (size_t)i + npred[i]++] = pr;
            ++succ_off[pr + 1];
        }
    }
    for (int i = 0; i < n; ++i) succ_off[i + 1] += succ_off[i];
    std::vector<int32_t> succ(succ_off[n]), cursor(succ_off.begin(), succ_off.end() - 1);
    for (int i = 0; i < n; ++i)
        for (int t = 0; t < npred[i]; ++t) succ[cursor[pred[3 * (size_t)i + t]]++] = i;

    // ALAP: recording order is topological, so one reverse sweep suffices
    std::vector<int32_t> alap(n, asap_depth);
    for (int i = n - 1; i >= 0; --i) {
        const int32_t need = ops[i].kind == OP_NOT ? alap[i] : alap[i] - 1;   // a NOT runs after its level's gates
        for (int t = 0; t < npred[i]; ++t) {
            int32_t &a = alap[pred[3 * (size_t)i + t]];
            a = std::min(a, need);
        }
    }

    if (alap_out) *alap_out = alap;

    // list scheduling, least slack first
    long long remaining = 0;
    for (const PendingOp &op : ops) remaining += op_rotations(op);
    std::vector<int32_t> left(npred), est(n, 1);
    std::vector<std::vector<int32_t>> avail(asap_depth + 2);
    auto release_successors = [&](int32_t i, int32_t level_done, auto &&self) -> void {
        for (int32_t e = succ_off[i]; e < succ_off[i + 1]; ++e) {
            const int32_t s = succ[e];
            const bool is_not = ops[s].kind == OP_NOT;
            est[s] = std::max(est[s], is_not ? level_done : level_done + 1);
            if (--left[s] != 0) continue;
            if (is_not) {                       // linear: rides along with its operand's level
                lvl[s] = est[s];
                self(s, lvl[s], self);
            } else {
                avail[std::min<int32_t>(est[s], asap_depth + 1)].push_back(s);
            }
        }
    };
    for (int i = 0; i < n; ++i) {
        if (left[i] != 0) continue;
        if (ops[i].kind == OP_NOT) {
            if (npred[i] == 0) { lvl[i] = 0; est[i] = 0; }
        } else {
            avail[1].push_back(i);
        }
    }
    // NOTs of materialised inputs release their successors now (level 0 is "before level 1")
    for (int i = 0; i < n; ++i)
        if (ops[i].kind == OP_NOT && npred[i] == 0) release_successors(i, 0, release_successors);

    std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<HeapItem>> heap;
    std::vector<int32_t> batch;
    // rotations not yet scheduled, by deadline (ALAP level)
    std::vector<long long> due(asap_depth + 2, 0);
    for (int i = 0; i < n; ++i) due[alap[i]] += op_rotations(ops[i]);
    for (int L = 1; L <= asap_depth; ++L) {
        for (int32_t i : avail[L]) heap.push(HeapItem{alap[i], i});
        const long long levels_left = asap_depth - L + 1;
        // Width policy.  A launch costs by occupancy steps (MI355X, P128: up to 1 workgroup per
        // CU 4.2 ms, up to 2 per CU 6.5 ms, then 10.1 and 12.6 ms for 3 and 4 units), so the
        // cheap widths are exactly 1x `unit` for a level that could not be wider anyway and
        // multiples of 2x `unit`; a width just above a multiple of 2x `unit` is the worst buy.
        // Base width: `unit` once the remaining work fits in one unit per remaining level, else
        // 2 units.  The base is doubled only when the deadlines demand it: if for some future
        // level D the rotations due by D exceed what base-wide levels L..D can hold (earliest-
        // deadline-first feasibility), some level must be wider, and a full double-width level
        // now is cheaper than forced overflows later.
        const long long base = remaining <= (long long)unit * levels_left ? unit : 2LL * unit;
        long long cap = base, running = 0, excess = 0;
        const int horizon = std::min(asap_depth, L + 1023);     // bounded look-ahead keeps deep DAGs cheap
        for (int D = L; D <= horizon; ++D) {
            running += due[D];
            excess = std::max(excess, running - base * (long long)(D - L + 1));
        }
        while (excess > cap - base && cap < 16LL * unit) cap *= 2;
        long long width = 0;
        batch.clear();
        auto take = [&](long long limit, bool forced_only) {
            while (!heap.empty()) {
                const HeapItem top = heap.top();
                const int w = op_rotations(ops[top.idx]);
                if (top.alap > L && (forced_only || width + w > limit)) break;
                heap.pop();
                lvl[top.idx] = L;
                width += w;
                due[top.alap] -= w;
                batch.push_back(top.idx);
            }
        };
        take(cap, false);
        // gates at their deadline overflowed the width: round it up to the next multiple of
        // two units and fill that with slack gates (the second half of an occupancy step is cheap)
        if (width > cap) take((width + 2LL * unit - 1) / (2LL * unit) * (2LL * unit), false);
        remaining -= width;
        for (int32_t i : batch) release_successors(i, L, release_successors);
    }
    return asap_depth;
}

}  // namespace tfhe_hip
