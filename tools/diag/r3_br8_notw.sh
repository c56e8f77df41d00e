#!/bin/bash
# measurement: how much of the 8-wave form's latency is owed to its LDS twiddle images?  Builds the library with
# -DTFHE_HIP_BR8_NOLDSTW into a scratch copy and compares single-launch latencies (TFHE_HIP_BR12=0 both times).
set -eo pipefail
OUT=$PWD/gpurun_out/r3br8tw; mkdir -p $OUT
echo "=== as built" > $OUT/latency.txt
TFHE_HIP_BR12=0 timeout -k 10 200 python tools/gate_throughput.py 1 64 256 >> $OUT/latency.txt 2>&1
W=/tmp/notw; rm -rf $W; mkdir -p $W; cp -r peba1_amd include oracle tools $W/
(cd $W/peba1_amd/csrc && TFHE_HIP_DEFS=-DTFHE_HIP_BR8_NOLDSTW bash build.sh > $OUT/build.log 2>&1)
echo "=== 8-wave form without LDS twiddles" >> $OUT/latency.txt
(cd $W && TFHE_HIP_BR12=0 timeout -k 10 200 python tools/gate_throughput.py 1 64 256 >> $OUT/latency.txt 2>&1)
cat $OUT/latency.txt
