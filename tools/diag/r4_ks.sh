#!/bin/bash
# Key switch, last pass: whole-match time and key-switch time by tile size on the final sources,
# alternating on one box:  gpurun -- 'bash tools/diag/r4_ks.sh'
set -o pipefail
OUT=gpurun_out/r4ks; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for t in 16 32; do
    TFHE_HIP_KS_TILE=$t timeout -k 10 300 python bench.py --extras 0 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/b_$t.json 2> $OUT/b_$t.err || { tail -5 $OUT/b_$t.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("$OUT/b_$t.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("tile $t round $round: match %.1f ms, blind rotate %.1f, key switch %.1f ms, clock %.3f" % (j["match_ms"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], r["shader_clock_ghz"]))
PY
  done
done
