#!/bin/bash
# AddressSanitizer + UBSan over the host-side code that reads what a user hands in: vocabulary files (td_vocab.cpp), split
# patterns (td_regex.cpp + the matcher of td_regex.h), vocabularies (td_tables.cpp: build_tables).  CPU only, g++.
# Round 3: 30 000 mutated files, 170 000 mutated patterns, 1 600 random vocabularies: no report.
# usage: tools/sanitize/run.sh [seed]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/tokendagger_amd/csrc; S=$R/tools/sanitize; seed=${1:-1}
F="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$C -I$R/include"
g++ $F $S/vocab_asan.cpp $C/td_vocab.cpp -o /tmp/td_vocab_asan
g++ $F $S/rx_asan.cpp $C/td_regex.cpp -o /tmp/td_rx_asan
g++ $F $S/tab_asan.cpp $C/td_tables.cpp $C/td_regex.cpp -o /tmp/td_tab_asan
python $S/vocab_files.py $seed 4000 /tmp/td_vocab_asan
python $S/patterns.py $seed 20000 /tmp/td_rx_asan
/tmp/td_tab_asan $seed 300
# The code the DEVICE shares with the host (td_common.h: scanners, exact-key probes, merge rounds; td_regex.h: the matcher) runs
# in the CPU twin; under the sanitizers (round 3: the twin tests, the generic-pattern tests and 30 000 fuzz documents, no report):
#   g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined tests/twin/td_twin.cpp $C/td_tables.cpp $C/td_regex.cpp \
#       -o tests/twin/_build/libtdtwin.so
#   LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_twin_vs_reference.py llama4 1 200
#   (then delete tests/twin/_build/libtdtwin.so: tests/helpers.py rebuilds the plain one)
