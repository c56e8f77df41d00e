#!/bin/bash
# A/B on ONE box, variants interleaved (tools/diag/build_variants.sh builds the variant libraries first):
#   gpurun --timeout 1200 -- 'bash tools/diag/ab.sh <tag> "default pair" [sizes P128] [p80 p2048]'
# PARITY_VARS="a b" limits the parity runs to those variants (priority-only variants compute the same integers by construction).
# Per variant first a short parity run THROUGH that library (a measurement build that computes wrong words is not timed),
# each under its own timeout; then the launch times of independent gates, three rounds, variants alternating.
set -o pipefail
TAG=${1:-ab}; VARS=${2:-"default"}; SIZES=${3:-"1 256 512 4096 4096"}; SETS=${4:-""}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/tools/diag/_ab
# a variant is a library built by build_variants.sh, "default", or NAME=value: the default library under that environment setting
# ... or lib@NAME=value: that library under that setting
libof() { case "$1" in *@*) echo "$AB/libtfhe-hip-${1%%@*}.so";; default|*=*) echo "";; *) echo "$AB/libtfhe-hip-$1.so";; esac; }
envof() { case "$1" in *@*) echo "${1#*@}";; *=*) echo "$1";; *) echo "PEBA1_AB_NOTHING=1";; esac; }
for v in ${PARITY_VARS-$VARS}; do
  env $(envof $v) PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 240 python -m pytest -x -q -m gpu -p no:cacheprovider \
      "tests/test_gpu_kernels.py::test_blind_rotate_matches_oracle" "tests/test_gpu_kernels.py::test_every_selectable_kernel_form_is_bit_exact" \
      "tests/test_gpu_kernels.py::test_random_input_parity_soak_in_every_launch_form" > $OUT/parity_$v.log 2>&1 \
      || { echo "variant $v: parity run failed or timed out"; tail -15 $OUT/parity_$v.log; exit 1; }
  echo "variant $v parity: $(tail -1 $OUT/parity_$v.log)"
done
for round in 1 2 3; do
  for v in $VARS; do
    echo "=== round $round variant $v P128" >> $OUT/ab.txt
    env $(envof $v) PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 200 python tools/gate_throughput.py $SIZES 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
  done
done
for s in $SETS; do
  for round in 1 2; do
    for v in $VARS; do
      echo "=== round $round variant $v $s" >> $OUT/ab.txt
      env $(envof $v) PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 300 python tools/gate_throughput.py --$s 1 4096 4096 2>&1 | grep "G=" >> $OUT/ab.txt || exit 1
    done
  done
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
echo AB-DONE
