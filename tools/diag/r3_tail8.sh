#!/bin/bash
# A/B of "br_tail8" (the at most half-filled last round of a wide level on the 8-wave form) on one box:
# parity first, then the match and the sharded per-rank phase with the split off and on, twice each
set -o pipefail
OUT=gpurun_out/r3tail8; mkdir -p $OUT
timeout -k 10 600 python -m pytest "tests/test_gpu_kernels.py::test_every_selectable_kernel_form_is_bit_exact" \
   tests/test_gpu_circuits.py::test_function_f_128_slots_ciphertexts_match_oracle_digest tests/test_gpu_sharded.py::test_sharded_dag_ciphertexts_match_oracle_digest \
   -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for t in 0 1; do
  TFHE_HIP_BR_TAIL8=$t python bench.py --steps 3 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/match_t${t}_$rep.json 2>> $OUT/bench.err || exit 1
  TFHE_HIP_BR_TAIL8=$t python bench.py --mode sharded --steps 1 --warmup 1 --extras 0 --no-cpu-baseline > $OUT/sharded_t${t}_$rep.json 2>> $OUT/bench.err || exit 1
done; done
python - <<'PY'
import json
for rep in (1, 2):
    for t in (0, 1):
        j = json.loads(open(f"gpurun_out/r3tail8/match_t{t}_{rep}.json").read().strip().split("\n")[-1]); r = j["roofline"]
        s = json.loads(open(f"gpurun_out/r3tail8/sharded_t{t}_{rep}.json").read().strip().split("\n")[-1]); ph = s["logical_rank_phases"]
        print(f"tail8={t} rep {rep}: match {j['ms_per_step']:.1f} ms br {r['ms_blind_rotate_per_step']:.1f} clock {r['shader_clock_ghz']:.2f} | sharded partial {sum(ph['partial_ms_per_rank'])/8:.1f} combine {ph['combine_ms']:.1f} projected {ph['projected_match_ms_one_gpu_per_rank']:.1f}")
PY
