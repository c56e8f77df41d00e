#!/bin/bash
# sweep of the priority time slice of co-resident blind-rotate workgroups (TFHE_HIP_BR_FAIR = log2 shader cycles; 0 = off)
OUT=gpurun_out/${1:-fair}; mkdir -p $OUT
for round in 1 2; do
  for f in 18 0 12 14 16 20 22; do
    echo "=== round $round TFHE_HIP_BR_FAIR=$f" >> $OUT/fair.txt
    TFHE_HIP_BR_FAIR=$f timeout -k 10 200 python tools/gate_throughput.py 512 4096 4096 2>&1 | grep "G=" >> $OUT/fair.txt || exit 1
  done
done
python - <<PY
import re, collections
best = collections.defaultdict(lambda: collections.defaultdict(list)); x = None
for line in open("$OUT/fair.txt"):
    m = re.match(r"=== round \d+ TFHE_HIP_BR_FAIR=(\d+)", line)
    if m: x = int(m.group(1)); continue
    m = re.match(r"G=\s*(\d+) .* br\s+([0-9.]+) ms", line)
    if m: best[x][int(m.group(1))].append(float(m.group(2)))
for x in sorted(best):
    print("fair", x, {g: (round(min(v), 3), round(sum(v) / len(v), 3)) for g, v in best[x].items()})
PY
