#!/bin/bash
# End of a round in one gpurun call: the whole GPU suite as the driver runs it, smoke(), the driver's bench command, a
# one-screen digest of the line (incl. roofline.valu_issue and, at N > 1, dist).
#   gpurun --timeout 1200 -- 'bash tools/diag/final.sh [tag]'
set -o pipefail
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=6 -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
tail -10 $OUT/tests.log; echo "pytest rc $rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 150 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().split("\n")[-1])
r = j["roofline"]
print("value %.0f gates/s  %.1f ms per match  scaling %s  clock %.3f GHz" % (j["value"], j["ms_per_step"], j["scaling"], r["shader_clock_ghz"]))
print("roofline.frac %.3f (algorithmic HBM)  valu_issue.frac %.3f (vs chip peak %.3f)  traffic %s  valu_busy %s" %
      (r["frac"], r["valu_issue"]["frac"], r["valu_issue"]["frac_vs_chip_peak"], r["traffic"], (r["valu"] or {}).get("valu_busy_frac")))
print("valu_issue per kernel:", {k: round(v.get("frac") or 0, 3) for k, v in r["valu_issue"]["kernels"].items()})
print("4,096 gates:", {k: (round(v["ms_blind_rotate"], 2), round(v["roofline_frac_algorithmic"], 3)) for k, v in j["independent_gates_4096"].items()})
print("sweep ms:", {k: round(v["ms_blind_rotate"], 3) for k, v in j["independent_gates_sweep"].items()})
c = j["cpu_baseline"]
print("cpu_baseline %.0f gates/s on %d cores (%s; %.1f ms per gate on one thread); exact port %.0f gates/s" %
      (c["value"], c["cores"], c["kind"], c["ms_per_gate_single_thread"], c["exact_port_value"]))
PY
echo FINAL-DONE
