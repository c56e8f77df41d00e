# four ranks on one GPU (gloo exchange): the default N>1 mode (one match, slots sharded, gather, combine on rank 0)
# at the full 128 slots, exactly as the driver launches it except for the backend
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus 4 --backend gloo --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/multiproc4.json 2> gpurun_out/multiproc4.err; echo rc $?
grep -E "^\{" gpurun_out/multiproc4.json | cut -c1-1500; tail -3 gpurun_out/multiproc4.err
