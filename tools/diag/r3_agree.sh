#!/bin/bash
# does the bench line's per-kernel launch time agree with rocprofv3's for the same command?
set -o pipefail
OUT=gpurun_out/r3agree; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o prof -- python3 bench.py --extras 0 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
python3 tools/rocpd_to_csv.py stats "$(ls $OUT/stats/*_results.db | head -1)" $OUT/kernel_stats.csv && head -4 $OUT/kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r3agree/bench_under_rocprof.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("bench:", r["kernel"], r["launches"], "launches avg %.4f ms;" % r["avg_launch_ms"], r["other_blind_rotate_kernel"]["kernel"], r["other_blind_rotate_kernel"]["launches"], "launches avg %.4f ms" % r["other_blind_rotate_kernel"]["avg_launch_ms"], "frac %.4f" % r["frac"])
PY
rm -rf $OUT/stats
