// ASan/UBSan harness: build_tables on random vocabularies (duplicates, empty tokens, gaps, huge ranks, missing bytes ...)
#include "td_tables.h"
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <string>
#include <vector>
using namespace td;
int main(int argc, char** argv) {
    std::mt19937 rng((unsigned)atoi(argv[1]));
    const int iters = atoi(argv[2]);
    const char* pats[] = {"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+", "\\w+|\\s+", "[a-z]+|.", "bad(pattern"};
    int ok = 0, bad = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> bytes, sbytes; std::vector<int64_t> offs{0}, soffs{0}; std::vector<int32_t> ranks, sranks;
        const int nb = rng() % 4 == 0 ? (int)(rng() % 256) : 256;  // (some byte tokens missing)
        int32_t next = 0;
        for (int b = 0; b < nb; ++b) { bytes.push_back((uint8_t)b); offs.push_back((int64_t)bytes.size()); ranks.push_back(next++); }
        const int nm = (int)(rng() % 300);
        const bool corrupt = rng() % 4 == 0;
        for (int k = 0; k < nm; ++k) {
            const int len = corrupt ? (int)(rng() % 12) : 2 + (int)(rng() % 10);  // (0 = empty token)
            for (int j = 0; j < len; ++j) bytes.push_back((uint8_t)("abc \n\xc3\xa9xyz01"[rng() % 13]));
            offs.push_back((int64_t)bytes.size());
            const unsigned r = corrupt ? rng() % 20 : 19;
            ranks.push_back(r == 0 ? (int32_t)(rng() % 50) : r == 1 ? (int32_t)0x7FFFFFF0 : r == 2 ? -5 : r == 3 ? (int32_t)(1 << 21) : next++);
        }
        const int ns = (int)(rng() % 6);
        for (int k = 0; k < ns; ++k) {
            const std::string s = std::string("<|") + std::to_string(rng() % 4) + (rng() % 3 ? "|>" : "");
            sbytes.insert(sbytes.end(), s.begin(), s.end()); soffs.push_back((int64_t)sbytes.size());
            sranks.push_back(rng() % 5 == 0 ? 3 : 100000 + k);
        }
        HostTables H; std::string err;
        const int rc = build_tables(pats[rng() % 4], (int64_t)ranks.size(), bytes.data(), offs.data(), ranks.data(), ns, sbytes.data(), soffs.data(), sranks.data(), H, err);
        if (rc == 0) ++ok; else ++bad;
    }
    printf("%d built %d rejected\n", ok, bad);
    return 0;
}
