/*
 * TEST INFRASTRUCTURE ONLY (oracle/): plain-C CPU restatement of the reference hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; nothing under
 * tokendagger_amd/ links, loads or calls it.  It is deliberately simple (scalar, byte-keyed hash
 * map, O(n^2) merge) and shares no code with the product.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against (a) the compiled reference
 * itself (oracle/_ref/libtdref.so, built from /root/reference/src/tiktoken/tiktoken.cpp) on fuzzed
 * and fixture inputs and (b) the committed golden vectors under tests/golden/ that were generated
 * by that compiled reference (tools/make_golden.py).
 *
 * What is restated, and from where:
 *   tdo_split            CoreBPE::split_text            tiktoken.cpp:70-128   (PCRE2 match loop)
 *   next_piece_llama4    the Llama-4 split pattern      src/main.cpp:114, as matched by PCRE2 10.39
 *                        with PCRE2_UTF|PCRE2_UCP (tiktoken.cpp:51-58) and PCRE2_NOTEMPTY (:91)
 *   tdo_merge            get_rank / bpe_merge           tiktoken.cpp:282-368
 *   tdo_encode           CoreBPE::encode(text, {})      tiktoken.cpp:169-234  (whole-piece fast path
 *                        :209-215, byte_pair_encode :371-378)
 *   tdo_encode(ordinary) CoreBPE::encode_ordinary      tiktoken.cpp:156-167  (no fast path)
 *   tdo_decode           CoreBPE::decode_bytes          tiktoken.cpp:236-255
 *
 * The regex engine (PCRE2, a system library absent from /root/reference; version in this image
 * 10.39, Unicode 14.0.0) is replaced by a deterministic scanner that reproduces what its ordered,
 * backtracking alternation yields for this one pattern; per-code-point classes come from
 * oracle/generated/unicode_classes.inc, which tools/gen_unicode_classes.py probed from that PCRE2.
 *
 * Deliberate deviation (SURVEY 8b "Errors"): a single-byte piece whose byte is not in the vocab
 * makes the reference return a garbage id (non-throwing emhash8 at(), tiktoken.cpp:373); here it is
 * an error, like the multi-byte case (tiktoken.cpp:364).
 */
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "generated/unicode_classes.inc"

enum { C_OTHER = 0, C_APOS, C_SLASH, C_SP, C_WS, C_CRLF, C_UP, C_LW, C_LB, C_MK, C_NUM };

static __thread char g_err[256];
const char* tdo_last_error(void) { return g_err; }

/* ---------------------------------------------------------------- UTF-8 + classes ---------- */

static int class_of_cp(uint32_t cp) {
    if (cp > 0x10FFFF) return C_OTHER;
    return td_ucls_stage2[(uint32_t)td_ucls_stage1[cp >> 8] * 256u + (cp & 255u)];
}

/* Character starting at byte i (i < n).  Malformed input never matches a property: a lead byte is
 * followed by at most (declared length - 1) continuation bytes; if fewer are present the truncated
 * group is one OTHER character; stray continuation / invalid lead bytes are 1-byte OTHER characters.
 * (The reference passes PCRE2_NO_UTF_CHECK, tiktoken.cpp:91, i.e. assumes valid UTF-8.) */
static int char_at(const uint8_t* s, int64_t i, int64_t n, int* len, uint32_t* cp_out) {
    uint8_t b = s[i];
    uint32_t cp;
    int need;
    if (b < 0x80) { *len = 1; if (cp_out) *cp_out = b; return class_of_cp(b); }
    if (b >= 0xC2 && b <= 0xDF) { need = 1; cp = b & 0x1F; }
    else if (b >= 0xE0 && b <= 0xEF) { need = 2; cp = b & 0x0F; }
    else if (b >= 0xF0 && b <= 0xF4) { need = 3; cp = b & 0x07; }
    else { *len = 1; if (cp_out) *cp_out = 0xFFFFFFFFu; return C_OTHER; }
    int got = 0;
    while (got < need && i + 1 + got < n && (s[i + 1 + got] & 0xC0) == 0x80) {
        cp = (cp << 6) | (s[i + 1 + got] & 0x3F);
        ++got;
    }
    *len = 1 + got;
    if (got < need || (cp >= 0xD800 && cp <= 0xDFFF)) { if (cp_out) *cp_out = 0xFFFFFFFFu; return C_OTHER; }
    if (cp_out) *cp_out = cp;
    return class_of_cp(cp);
}

static int is_U(int c) { return c == C_UP || c == C_LB || c == C_MK; }   /* [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}] */
static int is_W(int c) { return c == C_LW || c == C_LB || c == C_MK; }   /* [\p{Ll}\p{Lm}\p{Lo}\p{M}] */
static int is_L(int c) { return c == C_UP || c == C_LW || c == C_LB; }   /* \p{L} */
static int is_S(int c) { return c == C_SP || c == C_WS || c == C_CRLF; } /* \s */
static int is_P(int c) { return c != C_CRLF && !is_L(c) && c != C_NUM; } /* [^\r\n\p{L}\p{N}] */
static int is_X(int c) { return !is_S(c) && !is_L(c) && c != C_NUM; }    /* [^\s\p{L}\p{N}] */

/* (?i:'s|'t|'re|'ve|'m|'ll|'d)? at byte e; caseless under UTF+UCP: s also matches U+017F. */
static int64_t contraction(const uint8_t* s, int64_t e, int64_t n) {
    if (e >= n || s[e] != '\'') return e;
    int64_t p = e + 1;
    if (p >= n) return e;
    uint8_t a = s[p], al = (uint8_t)(a | 0x20);
    if (a < 0x80) {
        if (al == 's' || al == 't' || al == 'm' || al == 'd') return p + 1;
        if (p + 1 < n && s[p + 1] < 0x80) {
            uint8_t bl = (uint8_t)(s[p + 1] | 0x20);
            if ((al == 'r' && bl == 'e') || (al == 'v' && bl == 'e') || (al == 'l' && bl == 'l')) return p + 2;
        }
        return e;
    }
    if (a == 0xC5 && p + 1 < n && s[p + 1] == 0xBF) return p + 2; /* 'ſ (U+017F) ~ 's */
    return e;
}

/* End of the piece that starts at pos (pos < n), subject end n. */
/* variant 0: the Llama-4 / o200k pattern (src/main.cpp:114).  variant 1: the Mistral tekken pattern (tekken.json
 * config.pattern, read by tests/throughput_test.py:118): the same alternatives without the contraction suffix and with
 * \p{N} in place of \p{N}{1,3}. */
static int64_t next_piece_cl100k(const uint8_t* s, int64_t pos, int64_t n);
static int64_t next_piece_cl100k_v(const uint8_t* s, int64_t pos, int64_t n, int eos_first, int nmax);
static int64_t next_piece_gpt2(const uint8_t* s, int64_t pos, int64_t n);
static int64_t next_piece_llama4(const uint8_t* s, int64_t pos, int64_t n, int variant) {
    if (variant == 2) return next_piece_cl100k(s, pos, n);
    if (variant == 3) return next_piece_gpt2(s, pos, n);
    if (variant == 4) return next_piece_cl100k_v(s, pos, n, 1, 3);
    if (variant == 5) return next_piece_cl100k_v(s, pos, n, 0, 1);
    const int contr = variant == 0, nmax = variant == 0 ? 3 : 1;
    int l0;
    int c0 = char_at(s, pos, n, &l0, NULL);

    /* alternatives 1 and 2: optional 1-char prefix, tried taken-first (greedy `?`). */
    for (int alt = 1; alt <= 2; ++alt) {
        for (int with_prefix = 1; with_prefix >= 0; --with_prefix) {
            if (with_prefix && !is_P(c0)) continue;
            int64_t st = with_prefix ? pos + l0 : pos;
            /* maximal U-run from st, remembering the last char of it that is also in W */
            int64_t q = st, lastw = -1, lastw_end = -1;
            int l, c = -1;
            while (q < n) {
                c = char_at(s, q, n, &l, NULL);
                if (!is_U(c)) break;
                if (is_W(c)) { lastw = q; lastw_end = q + l; }
                q += l;
                c = -1;
            }
            if (alt == 1) {
                /* U* W+ : greedy U* backs off to the longest prefix after which a W char follows */
                int64_t e;
                if (q < n && c >= 0 && is_W(c)) {
                    e = q;
                    while (e < n) { c = char_at(s, e, n, &l, NULL); if (!is_W(c)) break; e += l; }
                } else if (lastw >= 0) {
                    e = lastw_end; /* W+ = that one char: everything after it in the run is U-only */
                } else {
                    continue;
                }
                return contr ? contraction(s, e, n) : e;
            } else {
                if (q == st) continue; /* U+ */
                int64_t e = q;
                while (e < n) { c = char_at(s, e, n, &l, NULL); if (!is_W(c)) break; e += l; }
                return contr ? contraction(s, e, n) : e;
            }
        }
    }
    /* alternative 3: \p{N}{1,3} */
    if (c0 == C_NUM) {
        int64_t e = pos + l0;
        for (int k = 1; k < nmax && e < n; ++k) {
            int l, c = char_at(s, e, n, &l, NULL);
            if (c != C_NUM) break;
            e += l;
        }
        return e;
    }
    /* alternative 4:  ?[^\s\p{L}\p{N}]+[\r\n/]* */
    for (int with_space = 1; with_space >= 0; --with_space) {
        if (with_space && c0 != C_SP) continue;
        int64_t st = with_space ? pos + 1 : pos;
        int64_t e = st;
        int l, c;
        while (e < n) { c = char_at(s, e, n, &l, NULL); if (!is_X(c)) break; e += l; }
        if (e == st) continue;
        while (e < n && (s[e] == '\r' || s[e] == '\n' || s[e] == '/')) ++e;
        return e;
    }
    /* alternatives 5-7 on the maximal whitespace run [pos, q) */
    {
        int64_t q = pos, last_crlf_end = -1, last_char = pos;
        int l, c;
        while (q < n) {
            c = char_at(s, q, n, &l, NULL);
            if (!is_S(c)) break;
            if (c == C_CRLF) last_crlf_end = q + l;
            last_char = q;
            q += l;
        }
        if (q > pos) {
            if (last_crlf_end >= 0) return last_crlf_end;      /* \s*[\r\n]+   */
            if (q == n) return q;                               /* \s+(?!\S) at end of subject */
            if (last_char > pos) return last_char;              /* \s+(?!\S) gives one char back */
            return q;                                           /* \s+ */
        }
    }
    /* unreachable for this pattern (every character is covered); keep the reference's
     * no-progress rule (tiktoken.cpp:120-122) as a guard */
    return pos + l0;
}

/* variant 2: cl100k_base / Llama-3 (tiktoken's pat_str; the reference accepts any pattern, wrapper.py:39):
 *   (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
 * \p{L} has no marks here: a combining mark is [^\s\p{L}\p{N}]. */
static int is_L2(int c) { return c == C_UP || c == C_LW || c == C_LB; }
/* variant 4: cl100k_base as current tiktoken releases spell it,
 *   '(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|\s+(?!\S)|\s
 * : the possessive quantifiers change nothing, but `\s++$` now stands IN FRONT of `\s*[\r\n]`, so a whitespace run that
 * reaches the end of the subject is one piece even if it contains CR/LF (eos_first). */
/* variant 5: Qwen2 / Qwen2.5 / Qwen3 (tokenizer.json pre_tokenizer): variant 2 with `\p{N}` (one digit per piece, nmax = 1). */
static int64_t next_piece_cl100k_v(const uint8_t* s, int64_t pos, int64_t n, int eos_first, int nmax);
static int64_t next_piece_cl100k(const uint8_t* s, int64_t pos, int64_t n) { return next_piece_cl100k_v(s, pos, n, 0, 3); }
static int64_t next_piece_cl100k_v(const uint8_t* s, int64_t pos, int64_t n, int eos_first, int nmax) {
    int l0;
    int c0 = char_at(s, pos, n, &l0, NULL);
    /* alternative 1: the contraction on its own */
    if (s[pos] == '\'') {
        int64_t e = contraction(s, pos, n);
        if (e != pos) return e;
    }
    /* alternative 2: optional prefix (tried taken first), then letters */
    for (int with_prefix = 1; with_prefix >= 0; --with_prefix) {
        if (with_prefix && (c0 == C_CRLF || is_L2(c0) || c0 == C_NUM)) continue;
        int64_t st = with_prefix ? pos + l0 : pos, e = st;
        int l, c;
        while (e < n) { c = char_at(s, e, n, &l, NULL); if (!is_L2(c)) break; e += l; }
        if (e > st) return e;
    }
    /* alternative 3: \p{N}{1,3} */
    if (c0 == C_NUM) {
        int64_t e = pos + l0;
        for (int k = 1; k < nmax && e < n; ++k) {
            int l, c = char_at(s, e, n, &l, NULL);
            if (c != C_NUM) break;
            e += l;
        }
        return e;
    }
    /* alternative 4:  ?[^\s\p{L}\p{N}]+[\r\n]* */
    for (int with_space = 1; with_space >= 0; --with_space) {
        if (with_space && c0 != C_SP) continue;
        int64_t st = with_space ? pos + 1 : pos, e = st;
        int l, c;
        while (e < n) { c = char_at(s, e, n, &l, NULL); if (is_S(c) || is_L2(c) || c == C_NUM) break; e += l; }
        if (e == st) continue;
        while (e < n && (s[e] == '\r' || s[e] == '\n')) ++e;
        return e;
    }
    /* alternatives 5-7: as in the Llama-4 pattern */
    {
        int64_t q = pos, last_crlf_end = -1, last_char = pos;
        int l, c;
        while (q < n) {
            c = char_at(s, q, n, &l, NULL);
            if (!is_S(c)) break;
            if (c == C_CRLF) last_crlf_end = q + l;
            last_char = q;
            q += l;
        }
        if (q > pos) {
            if (eos_first && q == n) return q;
            if (last_crlf_end >= 0) return last_crlf_end;
            if (q == n) return q;
            if (last_char > pos) return last_char;
            return q;
        }
    }
    return pos + l0;
}

/* variant 3: the GPT-2 pattern (r50k_base / p50k_base):
 *   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
 * case-sensitive contractions, the optional prefix is U+0020 only, digit runs are not cut, no trailer, no CR/LF rule. */
static int64_t next_piece_gpt2(const uint8_t* s, int64_t pos, int64_t n) {
    int l0;
    int c0 = char_at(s, pos, n, &l0, NULL);
    if (s[pos] == '\'' && pos + 1 < n) {
        uint8_t a = s[pos + 1];
        if (a == 's' || a == 't' || a == 'm' || a == 'd') return pos + 2;
        if (pos + 2 < n) {
            uint8_t b = s[pos + 2];
            if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) return pos + 3;
        }
    }
    for (int kind = 0; kind < 3; ++kind) {          /* letters, digits, other */
        for (int with_space = 1; with_space >= 0; --with_space) {
            if (with_space && c0 != C_SP) continue;
            int64_t st = with_space ? pos + 1 : pos, e = st;
            int l, c;
            while (e < n) {
                c = char_at(s, e, n, &l, NULL);
                int ok = kind == 0 ? is_L2(c) : kind == 1 ? (c == C_NUM) : (!is_S(c) && !is_L2(c) && c != C_NUM);
                if (!ok) break;
                e += l;
            }
            if (e > st) return e;
        }
    }
    {
        int64_t q = pos, last_char = pos;
        int l, c;
        while (q < n) {
            c = char_at(s, q, n, &l, NULL);
            if (!is_S(c)) break;
            last_char = q;
            q += l;
        }
        if (q > pos) {
            if (q == n) return q;                  /* \s+(?!\S) at the end of the subject */
            if (last_char > pos) return last_char; /* \s+(?!\S) gives one character back */
            return q;                              /* \s+ */
        }
    }
    return pos + l0;
}

/* ---------------------------------------------------------------- vocab map ---------------- */

typedef struct {
    int64_t n_vocab;
    uint8_t* blob;      /* token bytes, concatenated */
    int64_t* offs;      /* n_vocab + 1 */
    int32_t* ranks;     /* n_vocab */
    uint64_t mask;
    int32_t* slots;     /* index into vocab or -1 */
    int32_t max_rank;
    int64_t* by_rank;   /* rank -> vocab index or -1 (decode) */
    int variant;        /* split pattern: 0 Llama-4 / o200k, 1 tekken */
} tdo_t;

static uint64_t fnv(const uint8_t* p, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

static int32_t lookup(const tdo_t* t, const uint8_t* p, int64_t n) {
    uint64_t i = fnv(p, n) & t->mask;
    for (;;) {
        int32_t v = t->slots[i];
        if (v < 0) return INT_MAX;
        int64_t l = t->offs[v + 1] - t->offs[v];
        if (l == n && memcmp(t->blob + t->offs[v], p, (size_t)n) == 0) return t->ranks[v];
        i = (i + 1) & t->mask;
    }
}

void* tdo_create(int64_t n_vocab, const uint8_t* bytes, const int64_t* offs, const int32_t* ranks) {
    tdo_t* t = (tdo_t*)calloc(1, sizeof(tdo_t));
    t->n_vocab = n_vocab;
    t->blob = (uint8_t*)malloc((size_t)offs[n_vocab] + 1);
    memcpy(t->blob, bytes, (size_t)offs[n_vocab]);
    t->offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_vocab + 1));
    memcpy(t->offs, offs, sizeof(int64_t) * (size_t)(n_vocab + 1));
    t->ranks = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_vocab + 1));
    memcpy(t->ranks, ranks, sizeof(int32_t) * (size_t)n_vocab);
    uint64_t cap = 16;
    while (cap < (uint64_t)n_vocab * 2 + 2) cap <<= 1;
    t->mask = cap - 1;
    t->slots = (int32_t*)malloc(sizeof(int32_t) * cap);
    for (uint64_t i = 0; i < cap; ++i) t->slots[i] = -1;
    t->max_rank = -1;
    for (int64_t v = 0; v < n_vocab; ++v) {
        uint64_t i = fnv(t->blob + offs[v], offs[v + 1] - offs[v]) & t->mask;
        while (t->slots[i] >= 0) i = (i + 1) & t->mask;
        t->slots[i] = (int32_t)v;
        if (ranks[v] > t->max_rank) t->max_rank = ranks[v];
    }
    t->by_rank = (int64_t*)malloc(sizeof(int64_t) * (size_t)(t->max_rank + 2));
    for (int32_t r = 0; r <= t->max_rank; ++r) t->by_rank[r] = -1;
    for (int64_t v = 0; v < n_vocab; ++v) if (ranks[v] >= 0) t->by_rank[ranks[v]] = v;
    return t;
}

void tdo_destroy(void* h) {
    tdo_t* t = (tdo_t*)h;
    if (!t) return;
    free(t->blob); free(t->offs); free(t->ranks); free(t->slots); free(t->by_rank); free(t);
}

/* ---------------------------------------------------------------- split -------------------- */

/* piece END offsets of text[0,n); returns count (cap >= n is always enough) */
int64_t tdo_split_variant(const uint8_t* text, int64_t n, int64_t* ends, int64_t cap, int variant);
int64_t tdo_split(const uint8_t* text, int64_t n, int64_t* ends, int64_t cap) { return tdo_split_variant(text, n, ends, cap, 0); }
void tdo_set_variant(void* h, int variant) { ((tdo_t*)h)->variant = variant; }
int64_t tdo_split_variant(const uint8_t* text, int64_t n, int64_t* ends, int64_t cap, int variant) {
    int64_t pos = 0, k = 0;
    while (pos < n) {
        int64_t e = next_piece_llama4(text, pos, n, variant);
        if (k >= cap) { snprintf(g_err, sizeof g_err, "split capacity"); return -2; }
        ends[k++] = e;
        pos = e;
    }
    return k;
}

/* ---------------------------------------------------------------- merge -------------------- */

/* tiktoken.cpp:282-296 */
static int32_t get_rank(const tdo_t* t, const uint8_t* piece, const int64_t* start, int64_t nparts, int64_t idx) {
    if (idx + 3 < nparts) return lookup(t, piece + start[idx], start[idx + 3] - start[idx]);
    return INT_MAX;
}

/* tiktoken.cpp:298-368; appends ids to out[*k..]; returns 0 or -1 */
static int tdo_merge(const tdo_t* t, const uint8_t* piece, int64_t n, int32_t* out, int64_t* k, int64_t cap) {
    int64_t* start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int32_t* rank = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int64_t nparts = n + 1;
    int32_t min_rank = INT_MAX;
    int64_t min_idx = 0;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int32_t r = lookup(t, piece + i, 2);
        if (r < min_rank) { min_rank = r; min_idx = i; }
        start[i] = i; rank[i] = r;
    }
    start[n - 1] = n - 1; rank[n - 1] = INT_MAX;
    start[n] = n; rank[n] = INT_MAX;
    while (min_rank != INT_MAX) {
        int64_t i = min_idx;
        if (i > 0) rank[i - 1] = get_rank(t, piece, start, nparts, i - 1);
        rank[i] = get_rank(t, piece, start, nparts, i);
        memmove(start + i + 1, start + i + 2, sizeof(int64_t) * (size_t)(nparts - i - 2));
        memmove(rank + i + 1, rank + i + 2, sizeof(int32_t) * (size_t)(nparts - i - 2));
        --nparts;
        min_rank = INT_MAX; min_idx = 0;
        for (int64_t j = 0; j + 1 < nparts; ++j)
            if (rank[j] < min_rank) { min_rank = rank[j]; min_idx = j; }
    }
    int rc = 0;
    for (int64_t i = 0; i + 1 < nparts; ++i) {
        int32_t r = lookup(t, piece + start[i], start[i + 1] - start[i]);
        if (r == INT_MAX) {
            snprintf(g_err, sizeof g_err, "No value found for pair: %lld %lld", (long long)start[i], (long long)start[i + 1]);
            rc = -1; break;
        }
        if (*k >= cap) { snprintf(g_err, sizeof g_err, "encode capacity"); rc = -2; break; }
        out[(*k)++] = r;
    }
    free(start); free(rank);
    return rc;
}

/* The same merge (tiktoken.cpp:298-368: lowest rank first, leftmost on ties, ranks of the two pairs that touch the merged
 * part recomputed) in O(n log n): parts as a doubly linked list over byte positions + a binary min-heap of (rank, position,
 * right end of the pair when it was ranked) with lazy invalidation.  The O(n^2) form above IS the reference's loop; this one
 * exists so that the checker can answer for pieces of a megabyte (the compiled reference needs hours there).  It is pinned
 * against the quadratic form and against the compiled reference on pieces up to 20 KB (tests/test_oracle.py). */
typedef struct { int32_t rank; int32_t pos; int32_t mid; int32_t end; } hent_t;  /* pair = [pos, mid) + [mid, end) */
static int hless(const hent_t* a, const hent_t* b) { return a->rank < b->rank || (a->rank == b->rank && a->pos < b->pos); }
static void hpush(hent_t* h, int64_t* n, hent_t e) {
    int64_t i = (*n)++;
    h[i] = e;
    while (i > 0) { int64_t p = (i - 1) / 2; if (!hless(&h[i], &h[p])) break; hent_t t = h[i]; h[i] = h[p]; h[p] = t; i = p; }
}
static hent_t hpop(hent_t* h, int64_t* n) {
    hent_t top = h[0];
    h[0] = h[--(*n)];
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hless(&h[l], &h[m])) m = l;
        if (r < *n && hless(&h[r], &h[m])) m = r;
        if (m == i) break;
        hent_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
    return top;
}
static int tdo_merge_heap(const tdo_t* t, const uint8_t* piece, int64_t n, int32_t* out, int64_t* k, int64_t cap) {
    /* nxt[i]: start of the part after the part that starts at i (n for the last); prv[i]: start of the part before (-1) */
    int32_t* nxt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* prv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    uint8_t* alive = (uint8_t*)malloc((size_t)n + 1);
    hent_t* heap = (hent_t*)malloc(sizeof(hent_t) * (size_t)(3 * n + 8));
    int64_t hn = 0;
    for (int64_t i = 0; i < n; ++i) { nxt[i] = (int32_t)(i + 1); prv[i] = (int32_t)(i - 1); alive[i] = 1; }
    for (int64_t i = 0; i + 1 < n; ++i) {
        int32_t r = lookup(t, piece + i, 2);
        if (r != INT_MAX) { hent_t e = {r, (int32_t)i, (int32_t)(i + 1), (int32_t)(i + 2)}; hpush(heap, &hn, e); }
    }
    while (hn > 0) {
        hent_t e = hpop(heap, &hn);
        /* stale unless both parts are exactly as they were when the pair was ranked */
        if (!alive[e.pos] || nxt[e.pos] != e.mid || e.mid >= n || !alive[e.mid] || nxt[e.mid] != e.end) continue;
        /* merge: the right part is absorbed */
        alive[e.mid] = 0;
        nxt[e.pos] = e.end;
        if (e.end < n) prv[e.end] = e.pos;
        if (e.end < n) {  /* pair (merged part, part after it) */
            int32_t r = lookup(t, piece + e.pos, nxt[e.end] - e.pos);
            if (r != INT_MAX) { hent_t x = {r, e.pos, e.end, nxt[e.end]}; hpush(heap, &hn, x); }
        }
        if (prv[e.pos] >= 0) {  /* pair (part before it, merged part) */
            int32_t q = prv[e.pos];
            int32_t r = lookup(t, piece + q, e.end - q);
            if (r != INT_MAX) { hent_t x = {r, q, e.pos, e.end}; hpush(heap, &hn, x); }
        }
    }
    int rc = 0;
    for (int64_t i = 0; i < n; i = nxt[i]) {
        int32_t r = lookup(t, piece + i, nxt[i] - i);
        if (r == INT_MAX) { snprintf(g_err, sizeof g_err, "No value found for pair: %lld %lld", (long long)i, (long long)nxt[i]); rc = -1; break; }
        if (*k >= cap) { snprintf(g_err, sizeof g_err, "encode capacity"); rc = -2; break; }
        out[(*k)++] = r;
    }
    free(nxt); free(prv); free(alive); free(heap);
    return rc;
}
static int64_t g_heap_threshold = 4096;  /* pieces longer than this use the heap form */
void tdo_set_heap_threshold(int64_t n) { g_heap_threshold = n; }

/* ordinary != 0: encode_ordinary (no whole-piece fast path). Returns #tokens or <0. */
int64_t tdo_encode(void* h, const uint8_t* text, int64_t n, int32_t* out, int64_t cap, int ordinary) {
    const tdo_t* t = (const tdo_t*)h;
    int64_t pos = 0, k = 0;
    while (pos < n) {
        int64_t e = next_piece_llama4(text, pos, n, t->variant);
        const uint8_t* piece = text + pos;
        int64_t len = e - pos;
        int32_t r = (len == 1 || !ordinary) ? lookup(t, piece, len) : INT_MAX;
        if (r != INT_MAX) {
            if (k >= cap) { snprintf(g_err, sizeof g_err, "encode capacity"); return -2; }
            out[k++] = r;
        } else if (len == 1) {
            snprintf(g_err, sizeof g_err, "byte 0x%02x at offset %lld is not in the vocabulary", piece[0], (long long)pos);
            return -1;
        } else {
            int rc = len > g_heap_threshold ? tdo_merge_heap(t, piece, len, out, &k, cap) : tdo_merge(t, piece, len, out, &k, cap);
            if (rc < 0) return rc;
        }
        pos = e;
    }
    return k;
}

/* byte_pair_encode (tiktoken.cpp:371-378) of ONE piece as given: no split, no whole-piece lookup; the reference's quadratic loop.
 * (tests/test_char_seeds.py: the merge of a piece from seeded parts against the merge from its bytes.)  -> ids or < 0 */
int64_t tdo_merge_piece(void* h, const uint8_t* piece, int64_t len, int32_t* out, int64_t cap) {
    const tdo_t* t = (const tdo_t*)h;
    int64_t k = 0;
    if (len <= 0) return 0;
    if (len == 1) {
        int32_t r = lookup(t, piece, 1);
        if (r == INT_MAX) { snprintf(g_err, sizeof g_err, "byte 0x%02x is not in the vocabulary", piece[0]); return -1; }
        if (cap < 1) return -2;
        out[0] = r;
        return 1;
    }
    int rc = tdo_merge(t, piece, len, out, &k, cap);
    return rc < 0 ? rc : k;
}

/* tiktoken.cpp:236-255 (regular tokens only; specials are looked up by the caller's table) */
int64_t tdo_decode(void* h, const int32_t* toks, int64_t n, uint8_t* out, int64_t cap) {
    const tdo_t* t = (const tdo_t*)h;
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r = toks[i];
        int64_t v = (r >= 0 && r <= t->max_rank) ? t->by_rank[r] : -1;
        if (v < 0) { snprintf(g_err, sizeof g_err, "Invalid token for decoding: %d", r); return -1; }
        int64_t l = t->offs[v + 1] - t->offs[v];
        if (k + l > cap) { snprintf(g_err, sizeof g_err, "decode capacity"); return -2; }
        memcpy(out + k, t->blob + t->offs[v], (size_t)l);
        k += l;
    }
    return k;
}

/* class of one code point (for table tests) */
int tdo_class_of_cp(uint32_t cp) { return class_of_cp(cp); }
