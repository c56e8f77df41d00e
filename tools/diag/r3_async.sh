#!/bin/bash
# the pipelined flush: its GPU tests, then the match and identification bench lines
set -o pipefail
OUT=gpurun_out/r3async; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_gates.py tests/test_gpu_errors.py "tests/test_gpu_sharded.py::test_cfg4_identification_streams_matches_through_bounded_flushes" \
   tests/test_gpu_circuits.py::test_function_f_128_slots_ciphertexts_match_oracle_digest -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
tail -8 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
python bench.py --steps 4 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/bench_async.json 2> $OUT/bench.err || exit 1
python bench.py --mode identify --matches 16 --group 8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_identify_async.json 2>> $OUT/bench.err || exit 1
python - <<'PY'
import json
for n in ("bench_async", "bench_identify_async"):
    j = json.loads(open("gpurun_out/r3async/" + n + ".json").read().strip().split("\n")[-1]); r = j["roofline"]
    print(n, "ms_per_step %.1f value %.0f frac %.3f clock %.2f br %.1f ks %.1f match_ms %.1f" % (j["ms_per_step"], j["value"], r["frac"], r["shader_clock_ghz"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], j["match_ms"]))
PY
