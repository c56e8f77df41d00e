#!/bin/bash
# Round 4: where the time of the register forms of the key switch goes -- kernel-trace statistics and SQ counters of one match
#   gpurun -- 'bash tools/diag/r4_ksi_counters.sh 2 16'      (TFHE_HIP_KS_BRANCH, TFHE_HIP_KS_TILE)
set -o pipefail
BR=${1:-2}; TILE=${2:-16}
OUT=gpurun_out/r4ksi; mkdir -p $OUT
export TMPDIR=/tmp TFHE_HIP_KS_BRANCH=$BR TFHE_HIP_KS_TILE=$TILE
B="python3 bench.py --extras 0 --no-cpu-baseline --steps 1 --warmup 0"
K=keyswitch_index; [ "$BR" = 1 ] && K=keyswitch_branch; [ "$BR" = 0 ] && K=keyswitch_tile
db() { ls $1/*_results.db $1/*/*_results.db 2>/dev/null | head -1; }
d=$OUT/pass; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o prof -- $B > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
python3 tools/rocpd_to_csv.py stats "$(db $d)" $OUT/kernel_stats_${BR}_${TILE}.csv && head -8 $OUT/kernel_stats_${BR}_${TILE}.csv
: > $OUT/ks_counters_${BR}_${TILE}.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set -d $d -o sq -- $B > $OUT/pass.log 2>&1 || { tail -5 $OUT/pass.log; exit 1; }
  python3 tools/sq_summary.py "$(db $d)" $K >> $OUT/ks_counters_${BR}_${TILE}.txt
done
rm -rf $d
cat $OUT/ks_counters_${BR}_${TILE}.txt
echo KSI-DONE
