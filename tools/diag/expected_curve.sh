#!/bin/bash
# The expected SHAPE of the driver's strong-scaling curve (`bench.py --gpus N`, mode auto = ONE 128-slot match split over N
# ranks), projected from logical ranks timed on ONE device: per N the slowest rank's partial phase + rank 0's combine.
# A projection, not a multi-GPU measurement (README "Multi-GPU").  Also: the N > 1 code path rehearsed at FULL size with
# real processes over gloo on the one GPU (they share it: their times say nothing).
#   gpurun --timeout 900 -- 'bash tools/diag/expected_curve.sh'   ->  gpurun_out/r06_curve/
set -o pipefail
O=gpurun_out/r06_curve; mkdir -p $O
for n in 1 2 4 8; do
  for form in ref fast; do
    extra=""; [ $form = fast ] && extra="--fast-partial"
    timeout -k 10 200 python bench.py --mode sharded --slots 128 --logical-ranks $n --steps 2 --warmup 1 --no-cpu-baseline $extra \
        > $O/logical_${n}_$form.json 2> $O/logical_${n}_$form.err || { tail -3 $O/logical_${n}_$form.err; exit 1; }
  done
done
python3 - <<PY
import json
print("ranks  reference order: slowest partial + combine = projected match ms (speed-up)   latency form (--fast-partial, not the reference's gate order)")
base = {}
for n in (1, 2, 4, 8):
    row = []
    for form in ("ref", "fast"):
        j = json.loads(open("$O/logical_%d_%s.json" % (n, form)).read().strip().split("\n")[-1])
        p = j["logical_rank_phases"]
        t = p["projected_match_ms_one_gpu_per_rank"]
        base.setdefault(form, t)
        row.append("%7.0f + %5.0f = %7.0f ms (%.2fx)" % (max(p["partial_ms_per_rank"]), p["combine_ms"], t, base[form] / t))
    print("%5d  %s      %s" % (n, row[0], row[1]))
PY
timeout -k 10 400 python bench.py --gpus 4 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearsal_4_processes_gloo.json 2> $O/rehearsal_4.err || { tail -5 $O/rehearsal_4.err; exit 1; }
python3 -c "
import json; j=json.loads(open('$O/rehearsal_4_processes_gloo.json').read().strip().split('\n')[-1]); d=j['dist']
print('rehearsal, 4 processes on one GPU over gloo: world', d['world'], 'collectives', d['data_collectives'], 'status', d['status_word_collectives'], 'same issue order', d['same_issue_order_on_every_rank'], 'kernel in roofline', j['roofline']['valu_issue']['kernel'])"
echo CURVE-DONE
