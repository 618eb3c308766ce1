#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY. Builds the plain-C restatement oracle/td_oracle.c -> oracle/_build/libtdoracle.so
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$here/_build"
gcc -std=c11 -O2 -fPIC -shared -Wall -Wextra -I"$here" "$here/td_oracle.c" -o "$here/_build/libtdoracle.so"
echo "built $here/_build/libtdoracle.so"
