#!/bin/bash
# Diagnostic build: the library with -DTFHE_HIP_STAMPS (s_memtime phase stamps in the latency
# kernel) and a small harness.  Never used by tests or bench; outputs go to /tmp.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
S=$ROOT/peba1_amd/csrc
O=${1:-$ROOT/tools/diag/_stamps}
mkdir -p $O
ROCM=${ROCM_PATH:-/opt/rocm}
F="-O3 -std=c++17 -fPIC -ffp-contract=off -DTFHE_HIP_STAMPS"
KF="-mllvm -amdgpu-sched-strategy=max-ilp"
$ROCM/bin/hipcc $F $KF --offload-arch=gfx950 -c $S/kernels.hip -o $O/kernels.o
for f in host_keys engine shim scheduler io; do $ROCM/lib/llvm/bin/clang++ $F -D__HIP_PLATFORM_AMD__ -I$ROCM/include -c $S/$f.cpp -o $O/$f.o; done
$ROCM/bin/hipcc -shared -o $O/libtfhe-hip-stamps.so $O/kernels.o $O/host_keys.o $O/engine.o $O/shim.o $O/scheduler.o $O/io.o
$ROCM/lib/llvm/bin/clang++ -O2 -std=c++17 -I$ROOT/include $ROOT/tools/diag/stamp_main.cpp -o $O/stamp_main -L$O -ltfhe-hip-stamps -Wl,-rpath,$O
echo built $O/stamp_main
