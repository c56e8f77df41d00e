#!/bin/bash
# end of round 4: the whole GPU suite as the driver runs it, smoke(), then the default bench line (the driver's command)
#   gpurun --timeout 1200 -- 'bash tools/diag/r4_final.sh [bench]'
set -o pipefail
OUT=gpurun_out/r4final; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${1:-all}" != "bench" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=15 > $OUT/tests.log 2>&1; rc=$?
  echo "pytest rc $rc"; tail -25 $OUT/tests.log
  [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
t0=$(date +%s)
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
echo "bench.py (no flags) took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4final/bench.json").read().strip().split("\n")[-1]); r = j["roofline"]
print("match_ms %.1f value %.0f frac %.3f clock %s br_ms %.1f ks_ms %.1f" % (j["match_ms"], j["value"], r["frac"], r.get("shader_clock_ghz"), r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"]))
for k, v in j.items():
    if isinstance(v, dict) and "projected_match_ms_one_gpu_per_rank" in v:
        print("  ", k, "per-rank %.1f combine %.1f projected %.1f ms" % (v["per_rank_phase_ms_max"], v["combine_ms"], v["projected_match_ms_one_gpu_per_rank"]))
for k, v in j.get("independent_gates_4096", {}).items():
    print("  ", k, "%.0f rot/s frac %.3f clock %s rocprof %s" % (v["rotations_per_s_blind_rotate_only"], v["roofline_frac_algorithmic"], v["shader_clock_ghz"], "yes" if v.get("rocprof") else "no"))
for k, v in j.get("independent_gates_sweep", {}).items():
    print("   sweep G=%s: br %.2f ms ks %.2f ms -> %.0f rot/s (%s)" % (k, v["ms_blind_rotate"], v["ms_keyswitch"], v["rotations_per_s_blind_rotate_only"], v["kernel"]))
print("  weak_scaling", j.get("weak_scaling"))
if j.get("cpu_baseline"):
    c = j["cpu_baseline"]; print("  cpu", c["value"], c.get("cpu_model"), c["cores"], "all_cores", c.get("all_cores"))
PY
