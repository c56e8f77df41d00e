#!/bin/bash
# Host code under AddressSanitizer + UBSan: copies the tracked tree to a scratch directory, builds the host
# objects of the three libraries with the sanitizers (the device object is reused as built), and runs the CPU
# test suite against them.  Sanitizers are CPU-only on this pool; nothing here touches a GPU.
#   bash tools/diag/asan_cpu.sh [scratch-dir]         (round 3: 45 passed, no report)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-/tmp/asan_repo}
ROCM=${ROCM_PATH:-/opt/rocm}; CXX=$ROCM/lib/llvm/bin/clang++
rm -rf "$W"; mkdir -p "$W"; (cd "$ROOT" && git archive HEAD) | tar -x -C "$W"
[ -f "$ROOT/peba1_amd/csrc/kernels.o" ] || bash "$ROOT/peba1_amd/csrc/build.sh"
cp "$ROOT/peba1_amd/csrc/kernels.o" "$W/peba1_amd/csrc/"
cd "$W/peba1_amd/csrc"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -shared-libasan -g"
FLAGS="-O1 -std=c++17 -fPIC -ffp-contract=off $SAN"; HF="$FLAGS -D__HIP_PLATFORM_AMD__ -I$ROCM/include"
for f in host_keys engine shim scheduler io; do $CXX $HF -c $f.cpp -o $f.o & done; wait
$ROCM/bin/hipcc -shared $SAN -Wl,-Bsymbolic-functions -o ../libtfhe-hip.so kernels.o host_keys.o engine.o shim.o scheduler.o io.o
$CXX $FLAGS -I../../include -shared -o ../libpeba1-circuits.so circuits.cpp circuits_fast.cpp
$CXX $HF -I../../include -shared -o ../libpeba1-dist.so dist.cpp -L.. -lpeba1-circuits -Wl,-rpath,'$ORIGIN' -L$ROCM/lib -lamdhip64 -ldl
(cd "$W/oracle" && make -s)
cd "$W"
RT=$($CXX -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider > "$W/asan_run.log" 2>&1 || true
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$W/asan_run.log" || true)"
tail -2 "$W/asan_run.log"
