// ASan/UBSan harness: td_vocab.cpp's loaders on the files given on the command line (kind path)...
#include "td_vocab.h"
#include <stdio.h>
#include <string.h>
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        td::VocabData v;
        bool r = false;
        switch (argv[i][0]) {
            case 't': r = td::load_tiktoken_model(argv[i + 1], v); break;
            case 'h': r = td::load_hf_added_tokens(argv[i + 1], v, true); break;
            case 'k': r = td::load_tekken_json(argv[i + 1], v); break;
            case 'j': r = td::load_wrapper_json(argv[i + 1], "", v); break;
            case 's': r = td::load_wrapper_json("", argv[i + 1], v); break;
        }
        if (r) ++ok; else ++bad;
    }
    printf("%d loaded %d rejected\n", ok, bad);
    return 0;
}
