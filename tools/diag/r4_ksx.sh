#!/bin/bash
# What bounds the index form of the key switch: key-switch time of 4,096 independent gates (tools/gate_throughput.py) with
# the default library and with timing-only variants (tools/diag/build_variants.sh noload="-DKS_IDX_EXPERIMENT=1"
# nosub="-DKS_IDX_EXPERIMENT=2"), alternating on one box:
#   gpurun -- 'bash tools/diag/r4_ksx.sh "default noload nosub"'
set -o pipefail
VARS=${1:-"default noload nosub"}; OUT=gpurun_out/r4ksx; mkdir -p $OUT
export TMPDIR=/tmp TFHE_HIP_KS_BRANCH=${KSB:-2} TFHE_HIP_KS_TILE=${KST:-16}
AB=$PWD/tools/diag/_ab
libof() { if [ "$1" = default ]; then echo ""; else echo "$AB/libtfhe-hip-$1.so"; fi; }
for round in 1 2; do
  for v in $VARS; do
    echo "=== round $round variant $v" | tee -a $OUT/x.txt
    PEBA1_TFHE_HIP_LIB=$(libof $v) timeout -k 10 200 python tools/gate_throughput.py 512 1024 4096 4096 2>&1 | grep "G=" | tee -a $OUT/x.txt || exit 1
  done
done
