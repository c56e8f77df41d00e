#!/bin/bash
# Round 4 A/B on ONE box, variants interleaved: paired digit tables (BR_TAB_PAIRS) and layout H of the inverse
# transposes (BR_INV_LAYOUT_H), each on and off (tools/diag/build_variants.sh builds the variant libraries first):
#   gpurun --timeout 1200 -- 'bash tools/diag/r4_ab.sh r04ab "base pairs h both"'
set -o pipefail
TAG=${1:-r04ab}; VARS=${2:-"base both"}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/diag/_ab
libof() { if [ "$1" = default ]; then echo ""; else echo "$AB/libtfhe-hip-$1.so"; fi; }
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gates.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
  tail -2 $OUT/tests.log
fi
for round in 1 2 3; do
  for v in $VARS; do
    echo "=== round $round variant $v P128" >> $OUT/ab.txt
    PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 200 python tools/gate_throughput.py 1 256 512 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
  done
done
for round in 1 2; do
  for v in $VARS; do
    for s in p80 p2048; do
      echo "=== round $round variant $v $s" >> $OUT/ab.txt
      PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 300 python tools/gate_throughput.py --$s 1 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
    done
  done
done
python - <<PY
import re, collections
best = collections.defaultdict(lambda: collections.defaultdict(list))
key = None
for line in open("$OUT/ab.txt"):
    m = re.match(r"=== round \d+ variant (\S+) (\S+)", line)
    if m: key = (m.group(2), m.group(1)); continue
    m = re.match(r"G=\s*(\d+) .* br\s+([0-9.]+) ms", line)
    if m: best[key][int(m.group(1))].append(float(m.group(2)))
for k in sorted(best):
    print(k, {g: (round(min(v), 3), round(sum(v) / len(v), 3)) for g, v in best[k].items()}, "(min, mean) ms of blind rotate")
PY
# LDS counters of the P128 kernel per variant
SQ3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"
for v in $VARS; do
  export PEBA1_TFHE_HIP_LIB=$(libof $v)
  echo "== variant $v" >> $OUT/lds_counters.txt
  for set in "$SQ3" "$SQ2" "$SQ1"; do
    d=$OUT/sq_$v; rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $set -d $d -o sq -- python3 tools/gate_throughput.py 4096 > $OUT/sq_$v.log 2>&1 || { tail -5 $OUT/sq_$v.log; exit 1; }
    python3 tools/sq_summary.py "$(ls $d/*_results.db $d/*/*_results.db 2>/dev/null | head -1)" blind_rotate >> $OUT/lds_counters.txt
    rm -rf $d
  done
done
grep -E "variant|BANK_CONFLICT|IDX_ACTIVE|INSTS_LDS|WAIT_INST_LDS|INSTS_VALU|WAVE_CYCLES|WAIT_ANY|WAIT_INST_ANY" $OUT/lds_counters.txt
echo AB-DONE
