# the split (8-wave, half-transform) blind-rotate form: parity first, then its rate beside the other forms
set -e
mkdir -p gpurun_out/split
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu > gpurun_out/split/tests.log 2>&1 || { tail -30 gpurun_out/split/tests.log; exit 1; }
tail -3 gpurun_out/split/tests.log
for v in -1 0 2; do
  echo "== br_variant $v"
  TFHE_HIP_BR_VARIANT=$v timeout -k 10 120 python tools/gate_throughput.py --p2048 1 256 512 1024 4096 | grep "G=\|params"
done
TFHE_HIP_BR_VARIANT=2 timeout -k 10 120 python tools/gate_throughput.py 1 512 4096 | grep "G=\|params"
