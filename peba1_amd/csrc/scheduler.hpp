// scheduler.hpp -- slack-aware levelisation of a recorded gate DAG.
//
// ASAP levelling (level = 1 + max level of the operands) runs the reference's
// Function_f as ~110 fat levels (all slots' squares at once, throughput-bound)
// followed by a ~260-level narrow tail (the slot-by-slot ripple accumulation,
// Math.cpp:351-360, latency-bound at ~80 gates per level).  Most gates of the fat
// part have slack: slot k's square is not needed before the tail reaches slot k.
// schedule_levels() keeps the depth of the DAG (every gate at or before its ALAP
// level) but defers slack gates into later, narrower levels, filling each level up
// to the width the kernels run well at.  Only the order among independent gates
// changes; every gate computes the same integers.
#pragma once
#include <cstdint>
#include <vector>

namespace tfhe_hip {

struct PendingOp {
    uint8_t kind;       // 0..9 two-input gate code, OP_MUX, OP_NOT
    int32_t dst, a, b, c;   // slots; b, c = -1 when absent
    int32_t level;      // ASAP level (NOT: the level of its operand, 0 = already materialised)
};
constexpr uint8_t OP_MUX = 16, OP_NOT = 17;

inline int op_rotations(const PendingOp &op) { return op.kind == OP_NOT ? 0 : (op.kind == OP_MUX ? 2 : 1); }

// Fills lvl[i] with the level at which ops[i] runs (bootstrapped gates: 1..depth,
// NOTs: 0..depth, executed after the gates of that level).  `unit` = rotations one
// full pass of the latency kernel holds (the CU count); levels are filled to 1, 2
// or 4 units depending on how much work is left per remaining level.  With
// balance == false, or for trivial DAGs, lvl = ASAP.  Returns the depth.
// If alap_out is given it receives each op's ALAP level (== lvl when not balancing).
int schedule_levels(const std::vector<PendingOp> &ops, int asap_depth, bool balance, int unit,
                    std::vector<int32_t> &lvl, std::vector<int32_t> *alap_out = nullptr);

#ifdef TFHE_HIP_EXPERIMENTAL
// Execution order for the dataflow executor: a topological order of the DAG in which
// more urgent gates come first (balanced level, then ALAP level, then recording order;
// a NOT directly after the gates of its level).  order[k] = index into ops.
void priority_order(const std::vector<PendingOp> &ops, const std::vector<int32_t> &lvl,
                    const std::vector<int32_t> &alap, std::vector<int32_t> &order);

// Two execution lanes (HIP streams) for a levelised DAG.  Level-synchronous execution leaves
// a bubble at every level boundary (all workgroups of a launch start and end together, then
// the key switch runs alone).  Cutting the gates into two sets that each run their own
// level sequence, ordered against each other only where a gate really needs a result of the
// other set, lets the boundaries of one lane fall inside the launches of the other.
// lane 0 ("urgent"): gates whose slack (ALAP - ASAP level) is at most tight_slack -- the
// critical chains; lane 1: the rest.  A NOT rides in the lane of the gate that produces its
// operand (lane 0 when that is already materialised).  lanes_out[i] in {0, 1}.
void assign_lanes(const std::vector<PendingOp> &ops, const std::vector<int32_t> &alap, int tight_slack,
                  std::vector<uint8_t> &lanes_out);
#endif  // TFHE_HIP_EXPERIMENTAL

}  // namespace tfhe_hip
