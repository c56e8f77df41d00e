#!/bin/bash
# One gpurun call's worth of checks: GPU tests, gate-throughput table, bench line.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round_check.sh <tag> [tests|fast|none] [variants...]'
# variants: environment settings (NAME=value, or a bare value of TFHE_HIP_BR_VARIANT) to run the
# throughput table and the bench with (default "-1": the library's own choice of kernel forms)
set -o pipefail
TAG=${1:-r2x}; WHAT=${2:-tests}; shift; shift
VARIANTS=${@:--1}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rc=0
if [ "$WHAT" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
elif [ "$WHAT" = "fast" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gates.py tests/test_gpu_noise.py \
     "tests/test_gpu_circuits.py::test_function_f_ciphertexts_match_oracle_digest" tests/test_gpu_circuits.py::test_reference_object_code_on_the_gpu_library \
     tests/test_gpu_sharded.py::test_sharded_dag_ciphertexts_match_oracle_digest -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
fi
echo "pytest rc $rc"; [ -f $OUT/tests.log ] && tail -6 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for v in $VARIANTS; do
  case "$v" in *=*) export "$v";; *) export TFHE_HIP_BR_VARIANT=$v;; esac
  echo "=== variant $v" | tee -a $OUT/gate_throughput.txt
  timeout -k 10 200 python tools/gate_throughput.py 1 256 512 768 1024 4096 >> $OUT/gate_throughput.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/gate_throughput.py --p80 1 256 4096 >> $OUT/gate_throughput.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/gate_throughput.py --p2048 1 256 1024 4096 >> $OUT/gate_throughput.txt 2>&1 || exit 1
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/bench_match_v$v.json 2> $OUT/bench_match_v$v.err || exit 1
  python - <<PY
import json
j=json.loads(open("$OUT/bench_match_v$v.json").read().strip().split("\n")[-1])
print("variant $v match_ms", j["match_ms"], "value", j["value"], "frac", j["roofline"]["frac"])
PY
done
cat $OUT/gate_throughput.txt
echo ALL-DONE
