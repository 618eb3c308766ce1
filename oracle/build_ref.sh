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
# PCRE2's interpreter behind the reference's split loop (oracle/pcre2_interp.c): the oracle of the random-pattern tests
gcc -O1 -fPIC -shared -I"$here/shim" "$here/pcre2_interp.c" "$pcre" -o "$out/libpcre2interp.so"
echo "built $out/libpcre2interp.so"
# The reference's own Python extension (src/py_binding.cpp, unmodified, compiled where it lies) for the cpu_baseline leg that
# follows the reference's benchmark METHOD (Tokenizer.encode_batch: Python threads over the pybind CoreBPE.encode,
# tests/throughput_test.py:413-422).  extern/pybind11 is an empty submodule in the checkout: the installed pybind11 headers
# are used.  Lives in a directory of its own: its module name is the same as the product's extension.
pyinc="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])' 2>/dev/null || true)"
pbinc="$(python3 -c 'import pybind11; print(pybind11.get_include())' 2>/dev/null || true)"
ext="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))' 2>/dev/null || true)"
if [ -n "$pyinc" ] && [ -n "$pbinc" ] && [ -f "$ref/src/py_binding.cpp" ]; then
    mkdir -p "$out/refmod"
    g++ -std=c++17 -O2 -fPIC -w -pthread -shared \
        -I"$here/shim" -I"$ref/src/tiktoken" -I"$ref/src" -I"$pyinc" -I"$pbinc" \
        "$ref/src/py_binding.cpp" "$ref/src/tiktoken/tiktoken.cpp" \
        "$pcre" -o "$out/refmod/_tokendagger_core$ext"
    echo "built $out/refmod/_tokendagger_core$ext"
fi
