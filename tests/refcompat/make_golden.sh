#!/bin/bash
# Regenerates tests/golden/circuit_known_answers.json by running the REFERENCE's own
# circuit code (/root/reference/src/Math.cpp, compiled where it lies into a temp dir,
# never copied) over the plaintext-bit provider tests/mock/plain_tfhe.cpp, with the
# scenarios of tests/refcompat/driver.cpp (inputs of SURVEY.md 8c).
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
g++ -O1 -std=gnu++11 -fPIC -shared -I$ROOT/include $ROOT/tests/mock/plain_tfhe.cpp -o $T/libplain_tfhe.so
g++ -O1 -std=gnu++11 -w -DUSE_REFERENCE -I$ROOT/include -I/root/reference/include $ROOT/tests/refcompat/driver.cpp \
    /root/reference/src/Math.cpp -o $T/driver_ref -L$T -lplain_tfhe -Wl,-rpath,$T
$T/driver_ref > $ROOT/tests/golden/circuit_known_answers.json
rm -rf $T
echo "wrote tests/golden/circuit_known_answers.json"
