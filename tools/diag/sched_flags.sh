# does the compiler's instruction scheduling strategy matter for the blind-rotate kernel?
set -e
for f in "-O3" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -misched-cluster=0" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"; do
  HIP_EXTRA_FLAGS="$f" bash peba1_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== flags: '$f'"
  python tools/gate_throughput.py 1 512 4096 | grep "G="
  python tools/gate_throughput.py 4096 | grep "G="
done
