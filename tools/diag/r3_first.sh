#!/bin/bash
# round 3, first GPU call: whole GPU suite, the default bench line, and the key switch with in-place accumulation
set -o pipefail
OUT=gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=25 > $OUT/tests.log 2>&1; rc=$?
echo "pytest rc $rc"; tail -40 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
for a in 0 1; do
  TFHE_HIP_KS_ATOMIC=$a timeout -k 10 300 python bench.py --steps 3 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/bench_ksatomic$a.json 2> $OUT/bench_ksatomic$a.err || exit 1
done
python - <<'PY'
import json
for n in ("bench", "bench_ksatomic0", "bench_ksatomic1"):
    j = json.loads(open(f"gpurun_out/r3a/{n}.json").read().strip().split("\n")[-1])
    r = j["roofline"]
    print(n, "match_ms %.1f value %.0f frac %.3f clock %s br_ms %.1f ks_ms %.1f" % (j["match_ms"], j["value"], r["frac"], r.get("shader_clock_ghz"), r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"]))
    if "independent_gates_4096" in j:
        for k, v in j["independent_gates_4096"].items():
            print("  ", k, "%.0f rot/s frac %.3f clock %s" % (v["rotations_per_s_blind_rotate_only"], v["roofline_frac_algorithmic"], v["shader_clock_ghz"]))
    if j.get("cpu_baseline"):
        print("  cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cpu_model"], j["cpu_baseline"]["cores"])
PY
echo ALL-DONE
