#!/bin/bash
# Key switch, last pass: whole-match time and key-switch time by tile size and by the width of a
# thread's column (TFHE_HIP_KS_NARROW: 2 words per thread instead of 4), alternating on one box:
#   gpurun -- 'bash tools/diag/r4_ks.sh "16:0:0 16:1:0 32:1:0 16:0:1"'        (tile:narrow:pipe)
set -o pipefail
FORMS=${1:-"16:0:0 16:0:1"}
OUT=gpurun_out/r4ks; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for f in $FORMS; do
    IFS=: read t nw pp <<< "$f"; pp=${pp:-0}
    TFHE_HIP_KS_PIPE=$pp TFHE_HIP_KS_TILE=$t TFHE_HIP_KS_NARROW=$nw timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/b_${t}_${nw}.json 2> $OUT/b_${t}_${nw}.err || { tail -5 $OUT/b_${t}_${nw}.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("$OUT/b_${t}_${nw}.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("tile $t, %d words per thread, pipe $pp, round $round: match %.1f ms, blind rotate %.1f, key switch %.1f ms, clock %.3f" % (2 if $nw else 4, j["match_ms"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], r["shader_clock_ghz"]))
PY
  done
done
