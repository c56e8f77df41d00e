#!/bin/bash
# Round 4: compiler scheduling strategies and the priority time slice against the final kernel sources, variants
# interleaved on one box; then the raw-word parity soak of the final kernels.
#   tools/diag/build_variants.sh ilp="-mllvm -amdgpu-sched-strategy=max-ilp" ...   (see r4 notes), then
#   gpurun --timeout 1200 -- 'bash tools/diag/r4_sched.sh "default maxilp ..."'
set -o pipefail
VARS=${1:-"default"}; OUT=gpurun_out/r4sched; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/diag/_ab
libof() { if [ "$1" = default ]; then echo ""; else echo "$AB/libtfhe-hip-$1.so"; fi; }
for round in 1 2 3; do
  for v in $VARS; do
    echo "=== round $round variant $v P128" >> $OUT/ab.txt
    PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 200 python tools/gate_throughput.py 1 256 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
  done
done
for v in $VARS; do
  echo "=== round 1 variant $v p2048" >> $OUT/ab.txt
  PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 300 python tools/gate_throughput.py --p2048 1 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
done
for f in 16 17 18 19 20; do
  echo "=== round 1 variant fair$f P128" >> $OUT/ab.txt
  TFHE_HIP_BR_FAIR=$f timeout -k 10 200 python tools/gate_throughput.py 4096 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
done
python - <<PY
import re, collections
best = collections.defaultdict(lambda: collections.defaultdict(list))
key = None
for line in open("$OUT/ab.txt"):
    m = re.match(r"=== round \d+ variant (\S+) (\S+)", line)
    if m: key = (m.group(2), m.group(1)); continue
    m = re.match(r"G=\s*(\d+) .* br\s+([0-9.]+) ms", line)
    if m: best[key][int(m.group(1))].append(float(m.group(2)))
for k in sorted(best):
    print(k, {g: (round(min(v), 3), round(sum(v) / len(v), 3)) for g, v in best[k].items()}, "(min, mean) ms of blind rotate")
PY
echo SCHED-DONE
