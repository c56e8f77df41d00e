mkdir -p gpurun_out/trace
TFHE_HIP_TRACE_LEVELS=gpurun_out/trace/levels.txt TFHE_HIP_TRACE_DAG=gpurun_out/trace/dag.txt timeout -k 10 300 python bench.py --steps 1 --warmup 0 --extras 0 --no-cpu-baseline > gpurun_out/trace/bench.json 2> gpurun_out/trace/bench.err
ls -la gpurun_out/trace; gzip -f gpurun_out/trace/dag.txt; ls -la gpurun_out/trace
