#!/bin/bash
# Register form of the tiled key switch (TFHE_HIP_KS_BRANCH=1: rows in registers, scalar branches on the digit) against the
# LDS-strip form, whole match, alternating on one box:
#   gpurun -- 'bash tools/diag/r4_ksb.sh "16:0 16:1 32:1"'        (tile:branch)
set -o pipefail
FORMS=${1:-"16:0 16:1 32:1"}
OUT=gpurun_out/r4ksb; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -x -k "keyswitch" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for round in 1 2; do
  for f in $FORMS; do
    IFS=: read t br <<< "$f"
    TFHE_HIP_KS_BRANCH=$br TFHE_HIP_KS_TILE=$t timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/b_${t}_${br}.json 2> $OUT/b_${t}_${br}.err || { tail -5 $OUT/b_${t}_${br}.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("$OUT/b_${t}_${br}.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("tile $t, branch $br, round $round: match %.1f ms, blind rotate %.1f, key switch %.1f ms, clock %.3f" % (j["match_ms"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], r["shader_clock_ghz"]))
PY
  done
done
