# the reference's unmodified program on the drop-in, every flush traced level by level
#   gpurun -- 'bash tools/diag/refmain_breakdown.sh'  ->  gpurun_out/r06_refmain/{times.txt,breakdown.txt,out.txt}
set -o pipefail
D=gpurun_out/r06_refmain
mkdir -p $D; rm -f $D/times.txt
export TFHE_HIP_TRACE_TIMES=$D/times.txt
S=$(date +%s.%N)
timeout -k 10 500 oracle/_ref/tfhe_protocol_hip > $D/out.txt 2> $D/err.txt; rc=$?
E=$(date +%s.%N)
echo "rc $rc"; tail -2 $D/err.txt
W=$(python3 -c "print($E - $S)")
python3 tools/refmain_breakdown.py $D/times.txt --wall-s $W > $D/breakdown.txt && cat $D/breakdown.txt
grep -E "seconds" $D/out.txt | tail -12
