#!/bin/bash
# Builds libtfhe-hip variants that differ in compile-time switches of the kernels, for A/B runs on one box:
#   tools/diag/build_variants.sh tg2="-DBR_TAB_GROUPS=2" stamps="-DTFHE_HIP_STAMPS" ...
# -> tools/diag/_ab/libtfhe-hip-<name>.so (git-ignored; travels to the GPU box).  Select with PEBA1_TFHE_HIP_LIB.
# Host objects come from the default build (run peba1_amd/csrc/build.sh first).
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
S=$ROOT/peba1_amd/csrc; O=$ROOT/tools/diag/_ab; mkdir -p $O
ROCM=${ROCM_PATH:-/opt/rocm}
F="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  # the build's scheduling strategy (build.sh) unless the variant names its own, or NOSTRAT for the compiler's default
  case "$defs" in *amdgpu-sched-strategy*) strat="";; *NOSTRAT*) strat=""; defs=${defs//NOSTRAT/};; *) strat="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
  $ROCM/bin/hipcc $F $strat $defs --offload-arch=gfx950 -c $S/kernels.hip -o $O/kernels-$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  $ROCM/bin/hipcc -shared -Wl,-Bsymbolic-functions -o $O/libtfhe-hip-$name.so $O/kernels-$name.o $S/host_keys.o $S/engine.o $S/shim.o $S/scheduler.o $S/io.o
  echo "built $O/libtfhe-hip-$name.so"
done
