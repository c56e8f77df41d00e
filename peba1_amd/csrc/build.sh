#!/bin/bash
# Builds libtfhe-hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libtfhe-hip.so
ROCM=${ROCM_PATH:-/opt/rocm}
HIPCC=${HIPCC:-$ROCM/bin/hipcc}
CXX=${HOSTCXX:-$ROCM/lib/llvm/bin/clang++}
# TFHE_HIP_DEFS: extra -D switches for diagnostic builds (tools/diag/build_variants.sh, build_stamps.sh)
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result ${TFHE_HIP_DEFS-}"
HOSTFLAGS="$FLAGS -D__HIP_PLATFORM_AMD__ -I$ROCM/include"
# max-ilp scheduling: the blind-rotate kernel is bound by multiplier-class issue and dependency stalls;
# measured 1 % faster than the default strategy, max-memory-clause and no clustering 3-4 % slower
# (tools/diag/sched_flags.sh)
KFLAGS="$FLAGS ${HIP_EXTRA_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp} --offload-arch=gfx950"
$HIPCC $KFLAGS -c kernels.hip -o kernels.o &
# EMIT_KERNEL_ASM=<file>: also the assembly listing of exactly this compile (tools/isa_mix.py counts the instructions of a
# blind-rotate step from it; __graft_entry__.build() asks for it when profiles/isa_mix.json is stale)
if [ -n "${EMIT_KERNEL_ASM-}" ]; then
    $HIPCC $KFLAGS --cuda-device-only -S kernels.hip -o "$EMIT_KERNEL_ASM" 2>/dev/null &
    EXTRA_WAIT=1
fi
$CXX $HOSTFLAGS -c host_keys.cpp -o host_keys.o &
$CXX $HOSTFLAGS -c engine.cpp -o engine.o &
$CXX $HOSTFLAGS -c shim.cpp -o shim.o &
$CXX $HOSTFLAGS -c scheduler.cpp -o scheduler.o &
$CXX $HOSTFLAGS -c io.cpp -o io.o &
wait -n; wait -n; wait -n; wait -n; wait -n; wait -n
if [ -n "${EXTRA_WAIT-}" ]; then wait -n; fi
# -Bsymbolic-functions: calls between the library's own exported functions bind inside the library, so
# another provider of the tfhe API loaded RTLD_GLOBAL in the same process (a CPU tfhe, the tests'
# plaintext mock) cannot interpose on them
$HIPCC -shared -Wl,-Bsymbolic-functions -o $OUT kernels.o host_keys.o engine.o shim.o scheduler.o io.o
echo "built $(realpath $OUT)"
# circuits: calls only the public tfhe API; symbols resolve at load time against
# whichever provider is loaded first (libtfhe-hip.so, or the tests' plain mock)
$CXX $FLAGS -I../../include -shared -o ../libpeba1-circuits.so circuits.cpp circuits_fast.cpp
echo "built $(realpath ../libpeba1-circuits.so)"
# multi-GPU forms of the match for a C-ABI host (include/peba1_dist.h): host logic over the public gate API and the
# circuits; RCCL is opened with dlopen on first use, the gate provider resolves at load time like the circuits'
$CXX $HOSTFLAGS -I../../include -shared -o ../libpeba1-dist.so dist.cpp -L.. -lpeba1-circuits -Wl,-rpath,'$ORIGIN' \
    -L$ROCM/lib -lamdhip64 -ldl
echo "built $(realpath ../libpeba1-dist.so)"
