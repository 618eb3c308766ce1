/* TEST INFRASTRUCTURE ONLY: PCRE2's interpreter against its JIT on single patterns (the reference always JIT-compiles,
 * tiktoken.cpp:63).  Used at the desk to tell which differences a random-pattern fuzz found between td_regex and the compiled
 * reference are PCRE2's semantics (both agree) and which are the JIT's own (tests/test_generic_pattern.py).
 *   gcc -O1 -I oracle/shim oracle/pcre2_probe.c -o /tmp/pcre2_probe /usr/lib/x86_64-linux-gnu/libpcre2-8.so.0 && /tmp/pcre2_probe
 * PCRE2 10.39 here:  a+(?:q)?+a on xaaa1: no match with both (auto-possessification, compile stage: replicated in td_regex.cpp);
 * (?:ab|a)x*b on -ab and \p{P}??\p{Zs}+<space> on "K 9  A": the interpreter matches, the JIT does not (left alone). */
#define PCRE2_CODE_UNIT_WIDTH 8
#include "pcre2.h"
#ifndef PCRE2_NO_JIT
#define PCRE2_NO_JIT 0x00002000u
#endif
#include <stdio.h>
#include <string.h>
static void run(const char* pat, const char* subj, int jit) {
    int ec; PCRE2_SIZE eo;
    pcre2_code* re = pcre2_compile((PCRE2_SPTR8)pat, PCRE2_ZERO_TERMINATED, PCRE2_UTF | PCRE2_UCP, &ec, &eo, NULL);
    if (!re) { printf("compile error\n"); return; }
    if (jit) pcre2_jit_compile(re, PCRE2_JIT_COMPLETE);
    pcre2_match_data* md = pcre2_match_data_create_from_pattern(re, NULL);
    int rc = pcre2_match(re, (PCRE2_SPTR8)subj, strlen(subj), 0, PCRE2_NOTEMPTY | (jit ? 0 : PCRE2_NO_JIT), md, NULL);
    if (rc < 0) printf("  %-4s %-28s on %-10s : no match (%d)\n", jit ? "JIT" : "intp", pat, subj, rc);
    else { PCRE2_SIZE* ov = pcre2_get_ovector_pointer(md); printf("  %-4s %-28s on %-10s : [%zu,%zu)\n", jit ? "JIT" : "intp", pat, subj, ov[0], ov[1]); }
    pcre2_match_data_free(md); pcre2_code_free(re);
}
int main() {
    const char* cases[][2] = {{"a+(?:q)?+a", "xaaa1"}, {"a?(?:q)?+a", "xa1"}, {"(?:ab|a)x*b", "-ab"}, {"(?:ab|a)x*b", "ab"}, {"\\p{P}??\\p{Zs}+ ", "K 9  A"}, {"\\p{P}??\\p{Zs}+ ", "9  A"}};
    for (unsigned i = 0; i < sizeof cases / sizeof cases[0]; ++i) { run(cases[i][0], cases[i][1], 0); run(cases[i][0], cases[i][1], 1); }
    return 0;
}
