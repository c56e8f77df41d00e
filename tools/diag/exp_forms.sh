#!/bin/bash
# A/B of the experiment forms of the default blind-rotate kernel (TFHE_HIP_BR_EXP, kernels.hip blind_rotate4_body) on ONE
# box, interleaved so that clock drift hits every form alike:  gpurun -- 'bash tools/diag/exp_forms.sh "0 1 2 3 8" r3e'
set -o pipefail
FORMS=${1:-"0 1"}; TAG=${2:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2 3; do
  for x in $FORMS; do
    echo "=== round $round TFHE_HIP_BR_EXP=$x" >> $OUT/forms.txt
    TFHE_HIP_BR_VARIANT=$x timeout -k 10 200 python tools/gate_throughput.py 512 512 4096 4096 2>&1 | grep "G=" >> $OUT/forms.txt || exit 1
  done
done
python - <<PY
import re, collections
best = collections.defaultdict(lambda: collections.defaultdict(list))
x = None
for line in open("$OUT/forms.txt"):
    m = re.match(r"=== round \d+ TFHE_HIP_BR_EXP=(\d+)", line)
    if m: x = int(m.group(1)); continue
    m = re.match(r"G=\s*(\d+) .* br\s+([0-9.]+) ms", line)
    if m: best[x][int(m.group(1))].append(float(m.group(2)))
for x in sorted(best):
    print("exp", x, {g: (round(min(v), 3), round(sum(v) / len(v), 3)) for g, v in best[x].items()}, "(min, mean) ms of blind rotate")
PY
echo ALL-DONE
