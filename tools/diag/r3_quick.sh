#!/bin/bash
# quick GPU check of a kernel change: kernel + gate parity tests, the 2-slot digests, throughput table, a short bench
set -o pipefail
TAG=${1:-r3q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gates.py "tests/test_gpu_circuits.py::test_function_f_ciphertexts_match_oracle_digest" \
   tests/test_gpu_sharded.py::test_sharded_dag_ciphertexts_match_oracle_digest -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
echo "pytest rc $rc"; tail -5 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/gate_throughput.py 1 256 512 4096 > $OUT/gate_throughput.txt 2>&1 || exit 1
timeout -k 10 200 python tools/gate_throughput.py --p80 1 4096 >> $OUT/gate_throughput.txt 2>&1 || exit 1
timeout -k 10 200 python tools/gate_throughput.py --p2048 1 4096 >> $OUT/gate_throughput.txt 2>&1 || exit 1
cat $OUT/gate_throughput.txt
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python - <<PY
import json
j=json.loads(open("$OUT/bench.json").read().strip().split("\n")[-1]); r=j["roofline"]
print("match_ms %.1f value %.0f frac %.3f clock %.2f br %.1f ks %.1f avg_launch %.3f" % (j["match_ms"], j["value"], r["frac"], r["shader_clock_ghz"], r["ms_blind_rotate_per_step"], r["ms_keyswitch_per_step"], r["avg_launch_ms"]))
PY
echo ALL-DONE
