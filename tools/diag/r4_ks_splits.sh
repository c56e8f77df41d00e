#!/bin/bash
# Coefficient ranges of the (index-form) key switch: whole-match and key-switch time by the cap on the number of ranges
# (TFHE_HIP_KS_MAX_SPLITS; the engine takes the count <= cap whose grid fills whole rounds of resident workgroups)
#   gpurun -- 'bash tools/diag/r4_ks_splits.sh "32 48 64"'
set -o pipefail
VALS=${1:-"32 48 64"}; OUT=gpurun_out/r4kss; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for v in $VALS; do
    TFHE_HIP_KS_MAX_SPLITS=$v timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/b_$v.json 2> $OUT/b_$v.err || { tail -5 $OUT/b_$v.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("$OUT/b_$v.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("max splits $v, round $round: match %.1f ms, blind rotate %.1f, key switch %.1f ms, clock %.3f" % (j["match_ms"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], r["shader_clock_ghz"]))
PY
  done
done
