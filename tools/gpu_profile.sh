#!/bin/bash
# Round profiles (one gpurun call): kernel-trace statistics of the bench command, HBM traffic from
# separate --pmc FETCH_SIZE / WRITE_SIZE passes (the HBM/rocprofv3 recipe of MI355X_MICROARCH.md),
# SQ counters of the blind-rotate kernel on a 4,096-rotation launch (issue, LDS, stalls).
#   gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh r03'
# Counter passes never carry a trace option (gpurun refuses --pmc with trace domains).
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --extras 0 --no-cpu-baseline"
db() { ls $OUT/$1/*_results.db 2>/dev/null | head -1; }

# 1. kernel statistics of the bench command itself (same command as the JSON line next to it)
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o prof -- $B --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
python3 tools/rocpd_to_csv.py stats "$(db stats)" $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv

# 2. HBM-side traffic: two separate counter passes of the same one-match command
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c -d $OUT/pmc_$c -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_$c.err || { tail -5 $OUT/pmc_$c.err; exit 1; }
  python3 tools/rocpd_to_csv.py counter "$(db pmc_$c)" $OUT/pmc_$c.csv
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_blind_rotate.json > /dev/null && python3 -c "
import json; j=json.load(open('$OUT/pmc_blind_rotate.json')); print('hbm bytes per blind-rotate launch', j['hbm_bytes_per_launch'], 'sha', j['kernels_sha16'])"

# 3. SQ counters of blind_rotate4_kernel, 4,096 independent gates per launch (3 passes: hardware holds 8 SQ counters)
G="python3 tools/gate_throughput.py 4096"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU \
   -d $OUT/sq1 -o sq -- $G > $OUT/sq1.log 2>&1 || { tail -5 $OUT/sq1.log; exit 1; }
python3 tools/sq_summary.py "$(db sq1)" blind_rotate --json $OUT/valu_blind_rotate.json > $OUT/sq_counters_blind_rotate.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES \
   -d $OUT/sq2 -o sq -- $G > $OUT/sq2.log 2>&1 && python3 tools/sq_summary.py "$(db sq2)" blind_rotate >> $OUT/sq_counters_blind_rotate.txt
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
   -d $OUT/sq3 -o sq -- $G > $OUT/sq3.log 2>&1 && python3 tools/sq_summary.py "$(db sq3)" blind_rotate >> $OUT/sq_counters_blind_rotate.txt
cat $OUT/sq_counters_blind_rotate.txt
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq1 $OUT/sq2 $OUT/sq3      # the databases are large; summaries stay
echo PROFILE-DONE
