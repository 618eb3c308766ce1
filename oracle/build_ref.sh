#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libtdref.so = the UNMODIFIED reference core
# (/root/reference/src/tiktoken/tiktoken.cpp, compiled where it lies) + oracle/ref_driver.cpp,
# linked against the system PCRE2 runtime (libpcre2-8.so.0; no dev header/symlink in this image,
# hence oracle/shim/pcre2.h and the full-path link).  Outputs ONLY into oracle/_ref/ (git-ignored,
# but it travels to the GPU box with gpurun snapshots).  The reference's own build system is not run.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${TD_REFERENCE_ROOT:-/root/reference}"
out="$here/_ref"
if [ ! -f "$ref/src/tiktoken/tiktoken.cpp" ]; then
    echo "build_ref: $ref/src/tiktoken/tiktoken.cpp not present; keeping prebuilt $out (if any)" >&2
    exit 0
fi
pcre="$(ls /usr/lib/x86_64-linux-gnu/libpcre2-8.so.0 2>/dev/null || true)"
if [ -z "$pcre" ]; then echo "build_ref: libpcre2-8.so.0 not found" >&2; exit 1; fi
mkdir -p "$out"
g++ -std=c++17 -O2 -fPIC -w -pthread -shared \
    -I"$here/shim" -I"$ref/src/tiktoken" \
    "$ref/src/tiktoken/tiktoken.cpp" "$here/ref_driver.cpp" \
    "$pcre" -o "$out/libtdref.so"
echo "built $out/libtdref.so"
