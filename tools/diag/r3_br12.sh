#!/bin/bash
# the 12-wave latency form ("br12"): parity in every combination first, then narrow-launch latency, the match and the
# sharded per-rank phase with it off and on (same box, alternating)
set -o pipefail
OUT=gpurun_out/r3br12; mkdir -p $OUT; rm -f $OUT/latency.txt
timeout -k 10 600 python -m pytest "tests/test_gpu_kernels.py::test_every_selectable_kernel_form_is_bit_exact" tests/test_gpu_kernels.py::test_custom_gadgets_at_the_limit_of_the_kernel_forms \
   tests/test_gpu_circuits.py::test_function_f_128_slots_ciphertexts_match_oracle_digest tests/test_gpu_sharded.py::test_sharded_dag_ciphertexts_match_oracle_digest \
   tests/test_gpu_gates.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for t in 0 1; do
  echo "=== TFHE_HIP_BR12=$t" >> $OUT/latency.txt
  TFHE_HIP_BR12=$t timeout -k 10 200 python tools/gate_throughput.py 1 16 64 128 256 >> $OUT/latency.txt 2>&1 || exit 1
done
cat $OUT/latency.txt
for rep in 1 2; do for t in 0 1; do
  TFHE_HIP_BR12=$t python bench.py --steps 3 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/match_t${t}_$rep.json 2>> $OUT/bench.err || exit 1
  TFHE_HIP_BR12=$t python bench.py --mode sharded --steps 1 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/sharded_t${t}_$rep.json 2>> $OUT/bench.err || exit 1
done; done
python - <<'PY'
import json
for rep in (1, 2):
    for t in (0, 1):
        j = json.loads(open(f"gpurun_out/r3br12/match_t{t}_{rep}.json").read().strip().split("\n")[-1]); r = j["roofline"]
        s = json.loads(open(f"gpurun_out/r3br12/sharded_t{t}_{rep}.json").read().strip().split("\n")[-1]); ph = s["logical_rank_phases"]
        print(f"br12={t} run {rep}: match {j['ms_per_step']:.1f} ms (blind rotate {r['ms_blind_rotate_per_step']:.1f}, shader clock {r['shader_clock_ghz']:.2f} GHz) | sharded 256/8: per-rank phase {sum(ph['partial_ms_per_rank'])/8:.1f} ms, combine {ph['combine_ms']:.1f} ms, projected {ph['projected_match_ms_one_gpu_per_rank']:.1f} ms")
PY
