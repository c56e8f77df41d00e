# BASELINE configs[3] per-GPU load on one GPU: 128 independent 128-slot matches, 8 recorded per flush
mkdir -p gpurun_out/identify128
timeout -k 10 1000 python bench.py --mode identify --matches 128 --group 4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/identify128/bench.json 2> gpurun_out/identify128/bench.err
echo rc $?; tail -c 600 gpurun_out/identify128/bench.json
