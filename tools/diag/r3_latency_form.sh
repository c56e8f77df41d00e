#!/bin/bash
# the latency form of the sharded match (both phases depth-optimised): parity tests, then the logical-rank projection
set -o pipefail
OUT=gpurun_out/r3lat; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -k "latency_form or sharded_dag or fast_combine" -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
python bench.py --mode sharded --fast-partial --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/sharded_latency_form.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python bench.py --mode sharded --fast-partial --slots 128 --logical-ranks 8 --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/sharded128_latency_form.json 2>> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python bench.py --mode sharded --steps 1 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/sharded_reference_order.json 2>> $OUT/bench.err || exit 1
python - <<'PY'
import json
for n in ("sharded_latency_form", "sharded128_latency_form", "sharded_reference_order"):
    s = json.loads(open(f"gpurun_out/r3lat/{n}.json").read().strip().split("\n")[-1]); ph = s["logical_rank_phases"]
    print(f"{n}: levels(rank0 flushes) {s['config']['levels_per_step_rank0']} per-rank phase {sum(ph['partial_ms_per_rank'])/len(ph['partial_ms_per_rank']):.1f} ms, combine {ph['combine_ms']:.1f} ms, projected {ph['projected_match_ms_one_gpu_per_rank']:.1f} ms; whole step on one device {s['ms_per_step']:.1f} ms, {s['value']:.0f} gates/s")
PY
