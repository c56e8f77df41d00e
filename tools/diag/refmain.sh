mkdir -p gpurun_out/refmain
( time timeout -k 10 600 oracle/_ref/tfhe_protocol_hip > gpurun_out/refmain/out.txt 2> gpurun_out/refmain/err.txt ) 2> gpurun_out/refmain/time.txt; echo rc $?
cat gpurun_out/refmain/time.txt; tail -3 gpurun_out/refmain/out.txt; tail -3 gpurun_out/refmain/err.txt
grep -v -E "seconds|Function [fg]( bitwise)?: " gpurun_out/refmain/out.txt | diff - tests/golden/reference_main_output.txt | head -10; echo diffdone
grep -E "seconds" gpurun_out/refmain/out.txt | tail -12
timeout -k 10 600 python -m pytest tests/test_gpu_circuits.py tests/test_gpu_errors.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
