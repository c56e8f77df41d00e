set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider \
  --deselect tests/test_gpu_sharded.py::test_sharded_dag_ciphertexts_match_oracle_digest > gpurun_out/r2a/tests.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -5 gpurun_out/r2a/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r2a/bench_match.json 2> gpurun_out/r2a/bench_match.err || exit 1
timeout -k 10 300 python bench.py --mode sharded --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_sharded.json 2> gpurun_out/r2a/bench_sharded.err || exit 1
timeout -k 10 300 python bench.py --mode identify --matches 8 --group 4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2a/bench_identify.json 2> gpurun_out/r2a/bench_identify.err || exit 1
echo ALL-DONE
