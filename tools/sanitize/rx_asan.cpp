// ASan/UBSan harness: rx_compile on every line of a file (patterns, possibly garbage); accepted patterns are also matched
#include "td_regex.h"
#include <stdio.h>
#include <string.h>
#include <fstream>
#include <iostream>
using namespace td;
struct Subj { const uint8_t* p; uint32_t byte(int64_t i) const { return p[i]; } };
int main(int argc, char** argv) {
    std::ifstream f(argv[1], std::ios::binary);
    std::string line; int ok = 0, bad = 0;
    static RxProgram P;
    const RxTables T = rx_host_tables();
    const char* subjects[] = {"hello world 123", "aaa  bbb\n\nccc", "\xc3\xa9\xe4\xb8\xad x_y'z", "", "a", "   ", "<|x|>the an a0x1F"};
    while (std::getline(f, line)) {
        std::string err;
        if (!rx_compile(line, P, err)) { ++bad; continue; }
        ++ok;
        for (const char* s : subjects) {
            const int64_t n = (int64_t)strlen(s);
            Subj sub{(const uint8_t*)s};
            int64_t pos = 0; int guard = 0;
            while (pos < n && guard++ < 100) { int64_t ms, me; rx_next_piece(P, T, sub, pos, n, ms, me); if (me <= pos) break; pos = me; }
        }
    }
    printf("%d accepted %d rejected\n", ok, bad);
    return 0;
}
