#!/bin/bash
# Round 4: does the SQ expose the VALU pipe's busy CYCLES (not instruction counts)?  Lists the counters of the agent and
# collects the VALU-cycle candidates on 4,096 rotations, to set beside the issue-cost model of DESIGN.md section 7.
set -o pipefail
OUT=gpurun_out/r4valu; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/counters_avail.txt 2>&1
grep -i -E "^\s*(Name|Counter).*(VALU|INST_CYCLES|BUSY)" $OUT/counters_avail.txt | sort -u | head -60
for set in "SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  d=$OUT/pass; rm -rf $d
  if timeout -k 10 200 rocprofv3 --pmc $set -d $d -o sq -- python3 tools/gate_throughput.py 4096 > $OUT/pass.log 2>&1; then
    echo "== $set"
    python3 tools/sq_summary.py "$(ls $d/*_results.db $d/*/*_results.db 2>/dev/null | head -1)" blind_rotate
  else
    echo "== $set: refused"; tail -3 $OUT/pass.log
  fi
  rm -rf $d
done
