#!/bin/bash
# round 3, first GPU call: whole GPU suite, the default bench line, and the key switch with in-place accumulation
set -o pipefail
OUT=gpurun_out/r3a; mkdir -p $OUT
export TMPDIR=/tmp
SEL=${1:-tests}
timeout -k 10 900 python -m pytest $SEL -m gpu -q -x -p no:cacheprovider --durations=25 > $OUT/tests.log 2>&1; rc=$?
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
# narrow-launch latency of the kernel forms (is the split form a better latency form than the 8-wave one at N = 1024?)
for v in -1 2; do
  echo "=== TFHE_HIP_BR_VARIANT=$v" >> $OUT/latency.txt
  TFHE_HIP_BR_VARIANT=$v timeout -k 10 200 python tools/gate_throughput.py 1 64 256 512 4096 >> $OUT/latency.txt 2>&1 || exit 1
done
cat $OUT/latency.txt
# cross-lane against LDS transposes
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/xl -o xl -- python3 tools/diag/crosslane.py > $OUT/crosslane.log 2>&1 || { tail -20 $OUT/crosslane.log; exit 1; }
python3 tools/rocpd_to_csv.py stats "$(ls $OUT/xl/*_results.db | head -1)" $OUT/crosslane_kernel_stats.csv && grep negacyclic $OUT/crosslane_kernel_stats.csv
tail -8 $OUT/crosslane.log
rm -rf $OUT/xl
echo ALL-DONE
