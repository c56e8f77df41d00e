#!/bin/bash
# Round-4 profiles, one gpurun call: the same depth of rocprofv3 evidence for every parameter set
# on 4,096 independent gates per launch -- kernel-trace statistics, HBM-side traffic from separate --pmc FETCH_SIZE /
# WRITE_SIZE passes (MI355X_MICROARCH.md HBM section), three SQ passes (issue, wave-cycle shares, LDS) -- and the LDS
# conflict attribution of the P128 kernel: the LDS pass with the digit tables on and off.
#   gpurun --timeout 1200 -- 'bash tools/gpu_profile_sets.sh r04 [sets...]'      sets: p128 p80 p2048 attr (default all)
# Counter passes never carry a trace option (gpurun refuses --pmc with trace domains).
set -o pipefail
TAG=${1:-r04}; shift
SETS=${*:-attr p128 p80 p2048}
OUT=gpurun_out/sets_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
db() { ls $1/*/*_results.db $1/*_results.db 2>/dev/null | head -1; }

pass() {    # pass <dir> <log> <rocprof args...> -- <cmd...>
  local d=$1 log=$2; shift 2
  timeout -k 10 300 rocprofv3 "$@" > $log 2>&1 || { echo "FAILED: $d"; tail -5 $log; return 1; }
}

for s in $SETS; do
  case $s in
    attr)
      # LDS conflicts of blind_rotate4_kernel with the digit tables (default) and without them
      for t in 1 0; do
        d=$OUT/attr_table$t; mkdir -p $d
        TFHE_HIP_BR_TABLE=$t pass $d $d.log --pmc $SQ3 -d $d -o sq -- python3 tools/gate_throughput.py 4096 || exit 1
        echo "== br_digit_table=$t" >> $OUT/lds_conflict_attribution.txt
        grep "^G=" $d.log >> $OUT/lds_conflict_attribution.txt
        python3 tools/sq_summary.py "$(db $d)" blind_rotate >> $OUT/lds_conflict_attribution.txt
        TFHE_HIP_BR_TABLE=$t pass $d $d.log2 --pmc $SQ2 -d ${d}_w -o sq -- python3 tools/gate_throughput.py 4096 || exit 1
        python3 tools/sq_summary.py "$(db ${d}_w)" blind_rotate >> $OUT/lds_conflict_attribution.txt
        rm -rf $d ${d}_w
      done
      cat $OUT/lds_conflict_attribution.txt
      ;;
    p128|p80|p2048)
      case $s in p128) FLAG=""; NAME=P128; DIMS="630 1024 6";; p80) FLAG="--p80"; NAME=P80; DIMS="500 1024 4";; p2048) FLAG="--p2048"; NAME=P2048; DIMS="1024 2048 6";; esac
      R=$OUT/$s; mkdir -p $R
      G="python3 tools/gate_throughput.py $FLAG 4096"
      pass $R/stats $R/stats.log --kernel-trace --stats -d $R/stats -o prof -- $G || exit 1
      grep "^G=" $R/stats.log
      python3 tools/rocpd_to_csv.py stats "$(db $R/stats)" $OUT/${s}_kernel_stats.csv
      for c in FETCH_SIZE WRITE_SIZE; do
        pass $R/pmc_$c $R/pmc_$c.log --pmc $c -d $R/pmc_$c -o pmc -- $G || exit 1
      done
      pass $R/sq1 $R/sq1.log --pmc $SQ1 -d $R/sq1 -o sq -- $G || exit 1
      pass $R/sq2 $R/sq2.log --pmc $SQ2 -d $R/sq2 -o sq -- $G || exit 1
      pass $R/sq3 $R/sq3.log --pmc $SQ3 -d $R/sq3 -o sq -- $G || exit 1
      python3 tools/set_profile_summary.py $R $NAME $DIMS 4096 $OUT/set_profile_$NAME.json > /dev/null
      { for p in sq1 sq2 sq3; do python3 tools/sq_summary.py "$(db $R/$p)" blind_rotate; done; } > $OUT/${s}_sq_counters.txt
      python3 -c "
import json; j=json.load(open('$OUT/set_profile_$NAME.json'))
print('$NAME', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k not in ('sq_per_launch', 'source', 'corrections')})"
      rm -rf $R/stats $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/sq1 $R/sq2 $R/sq3      # the databases are large; summaries stay
      ;;
  esac
done
echo SETS-PROFILE-DONE
