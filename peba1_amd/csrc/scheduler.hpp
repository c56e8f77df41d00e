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


}  // namespace tfhe_hip
