#!/bin/bash
# Regenerates tests/golden/reference_main_output.txt: everything the REFERENCE's own program
# (/root/reference/src/main.cpp + Math.cpp + Client.cpp, compiled where they lie into a temp dir,
# never copied) prints when it runs over the plaintext-bit provider tests/mock/plain_tfhe.cpp with
# time() fixed (tests/refcompat/fixed_time.c), minus the lines that report durations.
# The GPU test runs the same sources linked against libtfhe-hip.so and must print the same lines.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
g++ -O1 -std=gnu++11 -fPIC -shared -I$ROOT/include $ROOT/tests/mock/plain_tfhe.cpp -o $T/libplain_tfhe.so
gcc -c $ROOT/tests/refcompat/fixed_time.c -o $T/fixed_time.o
g++ -O1 -std=gnu++11 -w -I$ROOT/include -I/root/reference/include /root/reference/src/main.cpp /root/reference/src/Math.cpp \
    /root/reference/src/Client.cpp $T/fixed_time.o -o $T/main_mock -L$T -lplain_tfhe -Wl,-rpath,$T
$T/main_mock | grep -v -E "seconds|Function [fg]( bitwise)?: " > ${1:-$ROOT/tests/golden/reference_main_output.txt}
rm -rf $T
echo "wrote ${1:-tests/golden/reference_main_output.txt}"
