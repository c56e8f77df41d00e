# multi-process rehearsal of the N>1 code paths on one GPU (gloo exchange): tests + the three bench modes at tiny size
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider -k "two_processes or rehearsal" 2>&1 | tail -5
for m in match identify; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 2 --backend gloo --mode $m --matches 2 --group 2 --steps 1 --warmup 0 --slots 8 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" | cut -c1-400
done
timeout -k 10 300 python bench.py --mode sharded --force-dist --steps 1 --warmup 0 --slots 8 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" | cut -c1-300
