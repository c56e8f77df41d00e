# the split (8-wave, half-transform) blind-rotate form: parity first, then its rate beside the other forms
set -e
mkdir -p gpurun_out/split
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "split_transform or every_selectable or p2048" > gpurun_out/split/tests.log 2>&1 || { tail -30 gpurun_out/split/tests.log; exit 1; }
tail -3 gpurun_out/split/tests.log
for t in 1 2 0; do
  echo "== N=2048 split, br_digit_table $t"
  TFHE_HIP_BR_TABLE=$t timeout -k 10 120 python tools/gate_throughput.py --p2048 1 256 512 4096 | grep "G=\|params"
done
