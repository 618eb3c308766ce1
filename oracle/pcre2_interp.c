/* TEST INFRASTRUCTURE ONLY: the reference's split loop (tiktoken.cpp:86-122: pcre2_match with PCRE2_NOTEMPTY from the end of the
 * last match, the rest of the subject as the last piece when nothing matches any more) on PCRE2's INTERPRETER.  The reference
 * JIT-compiles its pattern (tiktoken.cpp:63), and the JIT of the PCRE2 this image links (10.39) has bugs of its own on patterns
 * no tokenizer uses (oracle/pcre2_probe.c); random-pattern tests of td_regex compare with this instead.  Built by
 * oracle/build_ref.sh into oracle/_ref/libpcre2interp.so; bound by oracle/ref.py::interp_split. */
#define PCRE2_CODE_UNIT_WIDTH 8
#include "pcre2.h"
#include <stdint.h>
#include <string.h>
#ifndef PCRE2_NO_JIT
#define PCRE2_NO_JIT 0x00002000u
#endif
/* the reference's split loop (tiktoken.cpp:86-122) on PCRE2's INTERPRETER: -> n pieces as (start,end) pairs; -1 compile error */
int64_t interp_split(const char* pat, const uint8_t* subj, int64_t n, int64_t* out, int64_t cap) {
    int ec; PCRE2_SIZE eo;
    pcre2_code* re = pcre2_compile((PCRE2_SPTR8)pat, PCRE2_ZERO_TERMINATED, PCRE2_UTF | PCRE2_UCP, &ec, &eo, NULL);
    if (!re) return -1;
    pcre2_match_data* md = pcre2_match_data_create_from_pattern(re, NULL);
    int64_t k = 0, pos = 0;
    while (pos < n) {
        int rc = pcre2_match(re, subj, (PCRE2_SIZE)n, (PCRE2_SIZE)pos, PCRE2_NOTEMPTY | PCRE2_NO_JIT | 0x40000000u /* NO_UTF_CHECK */, md, NULL);
        if (rc < 0) { if (k < cap) { out[2*k] = pos; out[2*k+1] = n; } ++k; break; }
        PCRE2_SIZE* ov = pcre2_get_ovector_pointer(md);
        if (k < cap) { out[2*k] = (int64_t)ov[0]; out[2*k+1] = (int64_t)ov[1]; }
        ++k; pos = (int64_t)ov[1];
    }
    pcre2_match_data_free(md); pcre2_code_free(re);
    return k;
}
