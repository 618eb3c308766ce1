// Generic split patterns: a compiled form of the backtracking subset of PCRE2 patterns that tokenizer split
// patterns are written in, and the matcher that runs it — the SAME code on the device (td_generic.hip: one lane per
// document), on the host (td_regex.cpp compiles; td_api.cpp's last-piece helper) and in the CPU twin of the test-suite.
//
// What the reference does with a pattern (tiktoken.cpp:47-128): pcre2_compile(pattern, PCRE2_UTF | PCRE2_UCP), then a
// loop of pcre2_match(subject = the document, start_offset, PCRE2_NOTEMPTY): the leftmost match at or behind start_offset
// is a piece, the bytes it skipped over are NOT tokenized, and when nothing matches any more the rest of the text is one
// last piece.  rx_next_piece() is that loop's body.
//
// Supported syntax (td_regex.cpp rejects everything else with TD_E_PATTERN — there is no CPU regex fallback):
//   top-level alternation  A1|A2|...  of sequences of
//     character classes   [...]  [^...]  \s \S \d \D \w \W \h \H \v \V \N \p{Xx} \P{Xx} (general categories, one- and two-letter; the scripts of
//                         generated/unicode_scripts.inc: td_regex.cpp expands them into ranges; \p{Any})  .
//                         POSIX classes inside brackets as PCRE2_UCP reads them: [:alpha:] [:lower:] [:upper:] [:digit:] [:alnum:]
//                         [:space:] [:word:] [:cntrl:] and their negations [:^name:]
//                         literal and escaped characters, \r \n \t \f \v \xHH \x{H..}, ranges a-z
//     quantifiers         ? * + {m} {m,} {m,n}   their possessive forms ?+ *+ ++ {m,n}+   and their lazy forms ?? *? +? {m,n}?
//     groups of literals  (?:ab|c|[de])  (?i:'s|'t|ll)  optionally followed by ? ?+ ??   (case-insensitive: ASCII + U+017F / U+212A);
//                         with * + {m,n} (and their possessive / lazy forms) when every alternative is ONE character: a repeated class
//     look-ahead          (?!X)  (?=X)   with X one character class
//     look-behind         (?<!X) (?<=X)  with X one character class (looks at what stands in FRONT of the subject, see ^ \b below)
//     $ \Z                end of the subject (or in front of its final newline);  \z  the very end;  ^ \A  its start
//     \b \B               word boundary (\w as PCRE2_UCP defines it) / not one
//                         (^ \A \b \B and look-behinds look at what stands in FRONT of the subject: with them a call that cuts allowed special
//                         tokens out of the text is refused, TD_E_PATTERN — the reference would match the text behind a special
//                         with the special as left context, tiktoken.cpp:86-93)
//     a group followed by ?+ is atomic (the literal chosen, or the skip, is final); \0 followed by a digit (octal), a class
//     escape as the start of a range ([\d-z]) and surrogates in \x{..} are rejected as PCRE2 rejects or reads them differently
// Semantics are PCRE2's: ordered alternation, greedy quantifiers that give back one character at a time, possessive ones
// that do not, a match may not be empty — including the one place where PCRE2's auto-possessification is visible (a quantified
// class directly in front of a possessive optional group, td_regex.cpp).  Invalid UTF-8 (which PCRE2_NO_UTF_CHECK leaves undefined in the reference) is
// read as one character per byte that belongs to no category.
#pragma once
#include <stdint.h>

#include <string>

#include "td_common.h"

namespace td {

constexpr int RX_MAX_ALTS = 32, RX_MAX_NODES = 128, RX_MAX_CLASSES = 48, RX_MAX_ITEMS = 192, RX_MAX_LITS = 96, RX_MAX_LITBYTES = 768;
constexpr int RX_MAX_SEQ = 16;          // nodes of one alternative (the matcher's backtracking state is that deep)
constexpr uint32_t RX_INF = 0xFFFFu;    // no upper bound
constexpr uint32_t RX_F_S = 1u, RX_F_W = 2u, RX_F_D = 4u;  // item flags: \s \w \d (bits 5..7 of the table byte >> 5)

enum RxKind : uint8_t { RX_CLASS = 0, RX_LITSET = 1, RX_NLOOK = 2, RX_PLOOK = 3, RX_EOS = 4, RX_EOS_STRICT = 5, RX_BOS = 6, RX_WORDB = 7, RX_NWORDB = 8,
                        RX_NLOOKB = 9, RX_PLOOKB = 10 };
constexpr uint8_t RX_GREEDY = 0, RX_POSSESSIVE = 1, RX_LAZY = 2;  // RxNode::possessive

struct RxItem {        // one member of a character class
    uint32_t gc_mask;  // general categories (bit = category id of generated/unicode_gc.inc)
    uint32_t lo, hi;   // code point range (lo > hi: none)
    uint8_t flags;     // RX_F_*
    uint8_t negate;    // \S \W \D \P{..}
    uint8_t pad[2];
};
struct RxClass {
    uint16_t first_item, n_items;
    uint8_t negate;
    uint8_t pad[3];
    uint32_t ascii[4];  // membership of the code points 0..127, filled in by rx_compile: ASCII text never walks the items
};
struct RxLit { uint16_t off, len; };
struct RxNode {
    uint8_t kind;        // RxKind
    uint8_t possessive;  // RX_GREEDY / RX_POSSESSIVE (does not give characters back; a group: atomic) / RX_LAZY (as few as possible first)
    uint8_t caseless;    // RX_LITSET
    uint8_t pad;
    uint16_t a, b;       // RX_CLASS / RX_*LOOK: a = class; RX_LITSET: a = first literal, b = number of literals
    uint16_t min, max;   // RX_CLASS: repeat bounds; RX_LITSET: min 0 (optional) or 1, max 1
};
struct RxAlt { uint16_t first_node, n_nodes; };
struct RxProgram {
    uint32_t n_alts, n_nodes, n_classes, n_items, n_lits, n_litbytes;
    RxAlt alts[RX_MAX_ALTS];
    RxNode nodes[RX_MAX_NODES];
    RxClass classes[RX_MAX_CLASSES];
    RxItem items[RX_MAX_ITEMS];
    RxLit lits[RX_MAX_LITS];
    uint8_t litbytes[RX_MAX_LITBYTES];
    // bit a of first_alts[c]: alternative a CAN match at a position whose first byte is the ASCII character c (a superset,
    // filled in by rx_compile from the alternatives' leading nodes).  The matcher tries only those, in order: the others
    // would fail at their first character.  On the device this is what keeps the lanes of a wavefront together: every lane
    // goes through the alternatives loop once or twice with ITS alternative instead of all lanes through all of them.
    uint32_t first_alts[128];
};
static_assert(RX_MAX_ALTS <= 32, "RxProgram::first_alts holds one bit per alternative");
struct RxTables {  // generated/unicode_gc.inc (host arrays, or their copies in HBM)
    const uint16_t* stage1;
    const uint8_t* stage2;
};

// the character that starts at byte i of the subject (i < n): code point (or 0x110000 | byte for a byte that is not the
// start of a well-formed sequence) and its length in bytes
template <class A>
TD_HD uint32_t rx_char_at(const A& s, int64_t i, int64_t n, uint32_t& len) {
    const uint32_t b = s.byte(i);
    len = 1;
    if (b < 0x80u) return b;
    const uint32_t need = utf8_declared_len(b) - 1u;
    if (need == 0u || i + (int64_t)need >= n) return 0x110000u | b;  // not a lead byte, or the sequence leaves the subject
    uint32_t cp = b & (0xFFu >> (need + 2u));
    for (uint32_t k = 1; k <= need; ++k) {
        const uint32_t c = s.byte(i + k);
        if ((c & 0xC0u) != 0x80u) return 0x110000u | b;
        cp = (cp << 6) | (c & 0x3Fu);
    }
    if (cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) return 0x110000u | b;
    len = need + 1u;
    return cp;
}
// start of the character that ends at byte e (exclusive), not in front of `lo`: the inverse step of rx_char_at for what it
// accepted (a well-formed sequence) and one byte otherwise
template <class A>
TD_HD int64_t rx_prev_char(const A& s, int64_t lo, int64_t e, int64_t n) {
    int64_t p = e - 1;
    for (int k = 0; k < 3 && p > lo && (s.byte(p) & 0xC0u) == 0x80u; ++k) --p;
    uint32_t len;
    (void)rx_char_at(s, p, n, len);
    return (p + (int64_t)len == e) ? p : e - 1;
}

TD_HD bool rx_in_class_slow(const RxProgram& P, const RxTables& T, uint32_t cls, uint32_t cp);
TD_HD bool rx_in_class(const RxProgram& P, const RxTables& T, uint32_t cls, uint32_t cp) {
    if (cp < 128u) return (P.classes[cls].ascii[cp >> 5] >> (cp & 31u)) & 1u;
    return rx_in_class_slow(P, T, cls, cp);
}
TD_HD bool rx_in_class_slow(const RxProgram& P, const RxTables& T, uint32_t cls, uint32_t cp) {
    const RxClass c = P.classes[cls];
    uint32_t props = 28u;  // (category Cs: nothing) for bytes outside UTF-8
    if (cp < 0x110000u) props = T.stage2[(uint32_t)T.stage1[cp >> 8] * 256u + (cp & 255u)];
    const uint32_t gc = props & 31u, fl = props >> 5;
    bool in = false;
    for (uint32_t k = 0; k < c.n_items; ++k) {
        const RxItem it = P.items[c.first_item + k];
        bool m = cp < 0x110000u && (((it.gc_mask >> gc) & 1u) || (it.flags & fl) || (cp >= it.lo && cp <= it.hi));
        if (cp >= 0x110000u) m = false;
        in = in || (m != (it.negate != 0));
    }
    return in != (c.negate != 0);
}

// does literal `l` of the program stand at byte p of the subject?  -> bytes it takes there, or -1
template <class A>
TD_HD int rx_lit_at(const RxProgram& P, const RxLit l, bool caseless, const A& s, int64_t p, int64_t n) {
    int64_t q = p;
    for (uint32_t k = 0; k < l.len; ++k) {
        const uint32_t c = P.litbytes[l.off + k];
        if (q >= n) return -1;
        const uint32_t b = s.byte(q);
        if (b == c) { ++q; continue; }
        if (!caseless) return -1;
        const uint32_t lc = c | 0x20u;
        const bool letter = lc >= 'a' && lc <= 'z';
        if (letter && (b | 0x20u) == lc && b < 0x80u) { ++q; continue; }
        if (lc == 's' && b == 0xC5u && q + 1 < n && s.byte(q + 1) == 0xBFu) { q += 2; continue; }                                    // U+017F
        if (lc == 'k' && b == 0xE2u && q + 2 < n && s.byte(q + 1) == 0x84u && s.byte(q + 2) == 0xAAu) { q += 3; continue; }         // U+212A
        return -1;
    }
    return (int)(q - p);
}

// The matcher's backtracking state: where every node of the alternative ends (relative to the match start, 32 bits) and
// how many characters / which literal it took.  Indexed by the node the matcher is at — a run-time index: as arrays in
// registers every access was a chain of 17 compares and selects on the device, so there the state lives in LDS
// (td_generic.hip: RxStateLds, one bank per lane); on the host it is this struct.
struct RxStateLocal {
    int32_t e[RX_MAX_SEQ + 1];  // e[i + 1] = where node i's match ends; e[0] = 0: a node begins where the one in front of it ends
    uint32_t c[RX_MAX_SEQ];     // RX_CLASS: characters taken; RX_LITSET: literal chosen (b = skipped)
    TD_HD int32_t get_e(int i) const { return e[i]; }
    TD_HD void set_e(int i, int32_t v) { e[i] = v; }
    TD_HD uint32_t get_c(int i) const { return c[i]; }
    TD_HD void set_c(int i, uint32_t v) { c[i] = v; }
};

// one alternative, anchored at `start`: end of its (non-empty) match, or -1
template <class A, class St>
TD_HD int64_t rx_match_alt(const RxProgram& P, const RxTables& T, const RxAlt alt, const A& s, int64_t start, int64_t n, St& st) {
    // (offsets from `start` in 32 bits: the backtracking state is what the matcher's registers go to on the device, and
    // 64-bit positions doubled it; a single match is cut off 2 GiB behind its start)
    if (n - start > 0x7FFFFFF0ll) n = start + 0x7FFFFFF0ll;
    st.set_e(0, 0);
    struct Cnt {
        St& st;
        struct Ref { St& st; int i; TD_HD operator uint32_t() const { return st.get_c(i); } TD_HD Ref& operator=(uint32_t x) { st.set_c(i, x); return *this; }
                     TD_HD Ref& operator--() { st.set_c(i, st.get_c(i) - 1u); return *this; } TD_HD Ref& operator++() { st.set_c(i, st.get_c(i) + 1u); return *this; } };
        TD_HD Ref operator[](int i) const { return Ref{st, i}; }
    };
    struct Rel {
        St& st; int off; int64_t base;
        struct Ref { St& st; int i; int64_t base; TD_HD operator int64_t() const { return base + st.get_e(i); } TD_HD Ref& operator=(int64_t x) { st.set_e(i, (int32_t)(x - base)); return *this; } };
        TD_HD Ref operator[](int i) const { return Ref{st, i + off, base}; }
    };
    const Cnt cnt{st};
    const Rel beg{st, 0, start}, end{st, 1, start};  // (beg[i] reads end[i - 1])
    const int nn = (int)alt.n_nodes;
    int i = 0;
    int64_t pos = start;
    bool forward = true;
    for (;;) {
        if (forward) {
            if (i == nn) {
                if (pos > start) return pos;  // (PCRE2_NOTEMPTY: an empty match is a failure to back out of)
                forward = false;
                continue;
            }
            const RxNode nd = P.nodes[alt.first_node + i];
            bool ok = true;
            if (nd.kind == RX_CLASS) {
                uint32_t c = 0;
                int64_t p = pos;
                const uint32_t take_max = nd.possessive == RX_LAZY ? nd.min : nd.max;  // (lazy: the minimum first, one more per way back)
                while ((take_max == RX_INF || c < take_max) && p < n) {  // (RX_INF = no upper bound, not 65535: cnt[] is 32-bit)
                    uint32_t len;
                    const uint32_t cp = rx_char_at(s, p, n, len);
                    if (!rx_in_class(P, T, nd.a, cp)) break;
                    p += len;
                    ++c;
                }
                ok = c >= nd.min;
                cnt[i] = c;
                end[i] = p;
                pos = p;
            } else if (nd.kind == RX_LITSET) {
                uint32_t k = 0;
                int len = -1;
                for (; k < nd.b; ++k) {
                    len = rx_lit_at(P, P.lits[nd.a + k], nd.caseless != 0, s, pos, n);
                    if (len >= 0) break;
                }
                if (nd.min == 0 && nd.possessive == RX_LAZY) cnt[i] = nd.b;  // (lazy "(?:..)??": skipped first, the literals on the way back)
                else if (k < nd.b) { cnt[i] = k; pos += len; }
                else if (nd.min == 0) cnt[i] = nd.b;
                else ok = false;
                end[i] = pos;
            } else if (nd.kind == RX_EOS) {
                ok = pos == n || (pos == n - 1 && s.byte(pos) == '\n');
                end[i] = pos;
            } else if (nd.kind == RX_EOS_STRICT || nd.kind == RX_BOS) {
                ok = nd.kind == RX_BOS ? pos == 0 : pos == n;
                end[i] = pos;
            } else if (nd.kind == RX_WORDB || nd.kind == RX_NWORDB) {
                bool wl = false, wr = false;
                uint32_t len;
                if (pos > 0) {
                    const uint32_t cp = rx_char_at(s, rx_prev_char(s, 0, pos, n), n, len);
                    wl = cp < 0x110000u && ((T.stage2[(uint32_t)T.stage1[cp >> 8] * 256u + (cp & 255u)] >> 5) & RX_F_W);
                }
                if (pos < n) {
                    const uint32_t cp = rx_char_at(s, pos, n, len);
                    wr = cp < 0x110000u && ((T.stage2[(uint32_t)T.stage1[cp >> 8] * 256u + (cp & 255u)] >> 5) & RX_F_W);
                }
                ok = (wl != wr) == (nd.kind == RX_WORDB);
                end[i] = pos;
            } else if (nd.kind == RX_NLOOKB || nd.kind == RX_PLOOKB) {  // look-behind on one character
                bool in = false;
                if (pos > 0) {
                    uint32_t len;
                    in = rx_in_class(P, T, nd.a, rx_char_at(s, rx_prev_char(s, 0, pos, n), n, len));
                }
                ok = (nd.kind == RX_PLOOKB) ? in : !in;
                end[i] = pos;
            } else {  // look-ahead on one character
                bool in = false;
                if (pos < n) {
                    uint32_t len;
                    in = rx_in_class(P, T, nd.a, rx_char_at(s, pos, n, len));
                }
                ok = (nd.kind == RX_PLOOK) ? in : !in;
                end[i] = pos;
            }
            if (ok) ++i; else forward = false;
            continue;
        }
        // back out: the nearest node in front of i that has another way to match
        bool resumed = false;
        while (--i >= 0) {
            const RxNode nd = P.nodes[alt.first_node + i];
            if (nd.kind == RX_CLASS) {
                if (nd.possessive == RX_LAZY) {  // one character more, if there is one and the bound allows it
                    const int64_t p = end[i];
                    if ((nd.max != RX_INF && cnt[i] >= nd.max) || p >= n) continue;
                    uint32_t len;
                    if (!rx_in_class(P, T, nd.a, rx_char_at(s, p, n, len))) continue;
                    end[i] = p + len;
                    ++cnt[i];
                    pos = end[i];
                    resumed = true;
                    break;
                }
                if (nd.possessive || cnt[i] <= nd.min) continue;
                end[i] = rx_prev_char(s, beg[i], end[i], n);
                --cnt[i];
                pos = end[i];
                resumed = true;
                break;
            }
            if (nd.kind == RX_LITSET) {
                const bool lazy = nd.possessive == RX_LAZY && nd.min == 0;
                if (nd.possessive == RX_POSSESSIVE || (!lazy && cnt[i] >= nd.b)) continue;  // (atomic group "(?:..)?+": no second choice; or already skipped)
                uint32_t k = (lazy && cnt[i] == nd.b) ? 0u : cnt[i] + 1;  // (lazy: the skip came first, now the literals in order)
                int len = -1;
                for (; k < nd.b; ++k) {
                    len = rx_lit_at(P, P.lits[nd.a + k], nd.caseless != 0, s, beg[i], n);
                    if (len >= 0) break;
                }
                if (k < nd.b) { cnt[i] = k; pos = beg[i] + len; end[i] = pos; resumed = true; break; }
                if (nd.min == 0 && !lazy) { cnt[i] = nd.b; pos = beg[i]; end[i] = pos; resumed = true; break; }
                continue;
            }
        }
        if (!resumed) return -1;
        ++i;
        forward = true;
    }
}

// the pattern anchored at `start`: end of the match of the first alternative that matches, or -1
template <class A, class St>
TD_HD int64_t rx_match_at(const RxProgram& P, const RxTables& T, const A& s, int64_t start, int64_t n, St& st) {
    const uint32_t b = s.byte(start);  // (start < n)
    uint32_t m = b < 128u ? P.first_alts[b] : (P.n_alts >= 32u ? 0xFFFFFFFFu : (1u << P.n_alts) - 1u);
    for (; m; m &= m - 1u) {  // (ordered alternation: lowest alternative first)
        const int64_t e = rx_match_alt(P, T, P.alts[td_ctz32(m)], s, start, n, st);
        if (e >= 0) return e;
    }
    return -1;
}

// The reference's loop body (tiktoken.cpp:86-122) from byte `pos` of the subject [0, n): the next piece is [ms, me).
// Bytes [pos, ms) are skipped (no tokens).  When nothing matches any more, the rest [pos, n) is the last piece.
template <class A, class St>
TD_HD void rx_next_piece(const RxProgram& P, const RxTables& T, const A& s, int64_t pos, int64_t n, int64_t& ms, int64_t& me, St& st) {
    for (int64_t p = pos; p < n;) {
        const int64_t e = rx_match_at(P, T, s, p, n, st);
        if (e >= 0) { ms = p; me = e; return; }
        uint32_t len;
        (void)rx_char_at(s, p, n, len);
        p += len;
    }
    ms = pos;
    me = n;
}

template <class A>
TD_HD void rx_next_piece(const RxProgram& P, const RxTables& T, const A& s, int64_t pos, int64_t n, int64_t& ms, int64_t& me) {
    RxStateLocal st;
    rx_next_piece(P, T, s, pos, n, ms, me, st);
}

// ---- host side (td_regex.cpp) ----
bool rx_compile(const std::string& pattern, RxProgram& P, std::string& err);  // false: err says what is not supported
RxTables rx_host_tables();
const uint16_t* rx_stage1(size_t* n);
const uint8_t* rx_stage2(size_t* n);

}  // namespace td
