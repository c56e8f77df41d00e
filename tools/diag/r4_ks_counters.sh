#!/bin/bash
# Round 4: SQ counters of the tiled key switch over one match (what bounds it: DESIGN.md section 5), the dual-issue counter
# of the blind-rotate kernel, and the determinism soak (tools/soak.py: repeated matches, identical ciphertexts).
set -o pipefail
OUT=gpurun_out/r4ksc; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --extras 0 --no-cpu-baseline --steps 1 --warmup 0"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  d=$OUT/pass; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set -d $d -o sq -- $B > $OUT/pass.log 2>&1 || { tail -5 $OUT/pass.log; exit 1; }
  python3 tools/sq_summary.py "$(ls $d/*_results.db $d/*/*_results.db 2>/dev/null | head -1)" keyswitch_tile >> $OUT/ks_counters.txt
  rm -rf $d
done
cat $OUT/ks_counters.txt
d=$OUT/pass; rm -rf $d
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE -d $d -o sq -- python3 tools/gate_throughput.py 4096 > $OUT/pass.log 2>&1 \
  && python3 tools/sq_summary.py "$(ls $d/*_results.db $d/*/*_results.db 2>/dev/null | head -1)" blind_rotate | tee $OUT/valu2.txt
rm -rf $d
timeout -k 10 300 python3 tools/soak.py 6 > $OUT/soak.txt 2>&1 || { tail -5 $OUT/soak.txt; exit 1; }
tail -4 $OUT/soak.txt
echo KSC-DONE
