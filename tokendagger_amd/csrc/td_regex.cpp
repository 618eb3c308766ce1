// Host side of the generic split patterns: pat_str -> RxProgram (td_regex.h).  A recursive-descent reader for the subset
// of PCRE2 syntax listed there; anything else is an error message for TD_E_PATTERN (there is no CPU regex fallback and no
// silent approximation: what the reader accepts, the matcher runs with PCRE2's semantics, pinned against PCRE2 itself by
// tests/test_generic_pattern.py).
#include "td_regex.h"

#include <string>
#include <vector>

#include "generated/unicode_gc.inc"
#include "generated/unicode_scripts.inc"
#include "td_tables.h"

namespace td {

RxTables rx_host_tables() { return RxTables{td_ugc_stage1, td_ugc_stage2}; }
const uint16_t* rx_stage1(size_t* n) { *n = sizeof td_ugc_stage1 / sizeof td_ugc_stage1[0]; return td_ugc_stage1; }
const uint8_t* rx_stage2(size_t* n) { *n = sizeof td_ugc_stage2; return td_ugc_stage2; }

namespace {

// general category ids of generated/unicode_gc.inc
const char* const kGc[30] = {"Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps",
                             "Pe", "Pi", "Pf", "Po", "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co"};

struct Reader {
    const std::string& s;
    size_t i = 0;
    RxProgram& P;
    std::string err;
    Reader(const std::string& s_, RxProgram& P_) : s(s_), P(P_) {}

    bool fail(const std::string& what) {
        if (err.empty()) err = what + " (at offset " + std::to_string(i) + " of the split pattern)";
        return false;
    }
    bool eof() const { return i >= s.size(); }
    int peek(size_t k = 0) const { return i + k < s.size() ? (unsigned char)s[i + k] : -1; }

    // ---- program building ----
    bool new_item(const RxItem& it, uint16_t& idx) {
        if (P.n_items >= (uint32_t)RX_MAX_ITEMS) return fail("split pattern too large (class items)");
        idx = (uint16_t)P.n_items;
        P.items[P.n_items++] = it;
        return true;
    }
    bool new_class(uint16_t first_item, uint16_t n_items, bool negate, uint16_t& cls) {
        if (P.n_classes >= (uint32_t)RX_MAX_CLASSES) return fail("split pattern too large (classes)");
        cls = (uint16_t)P.n_classes;
        RxClass c{};
        c.first_item = first_item; c.n_items = n_items; c.negate = negate ? 1 : 0;
        P.classes[P.n_classes++] = c;
        return true;
    }
    static RxItem range_item(uint32_t lo, uint32_t hi) {
        RxItem it{};
        it.lo = lo; it.hi = hi;
        return it;
    }
    static RxItem none_item() { return range_item(1, 0); }
    bool single_class(const RxItem& it, uint16_t& cls) {
        uint16_t idx = 0;
        return new_item(it, idx) && new_class(idx, 1, false, cls);
    }

    // ---- lexical pieces ----
    bool read_codepoint_literal(uint32_t& cp) {  // one (possibly multi-byte) character of the pattern text
        const uint32_t b = (uint32_t)peek();
        if (b < 0x80u) { cp = b; ++i; return true; }
        const uint32_t need = utf8_declared_len(b) - 1u;
        if (need == 0 || i + need >= s.size()) return fail("split pattern is not valid UTF-8");
        cp = b & (0xFFu >> (need + 2u));
        for (uint32_t k = 1; k <= need; ++k) {
            const uint32_t c = (unsigned char)s[i + k];
            if ((c & 0xC0u) != 0x80u) return fail("split pattern is not valid UTF-8");
            cp = (cp << 6) | (c & 0x3Fu);
        }
        i += need + 1;
        return true;
    }
    bool read_hex(uint32_t& v) {  // behind "\x": HH or {H..}
        v = 0;
        auto hexv = [](int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
        if (peek() == '{') {
            ++i;
            int nd = 0;
            while (!eof() && peek() != '}') {
                const int h = hexv(peek());
                if (h < 0 || ++nd > 6) return fail("bad \\x{...} escape");
                v = v * 16 + (uint32_t)h;
                ++i;
            }
            if (eof() || nd == 0) return fail("bad \\x{...} escape");
            ++i;
            if (v >= 0xD800u && v <= 0xDFFFu) return fail("\\x{...} names a surrogate (PCRE2 error 173 in UTF mode)");
            return v <= 0x10FFFFu ? true : fail("\\x{...} beyond U+10FFFF");
        }
        for (int k = 0; k < 2; ++k) {
            const int h = hexv(peek());
            if (h < 0) break;
            v = v * 16 + (uint32_t)h;
            ++i;
        }
        return true;
    }
    // behind "\\p" / "\\P": a general category (-> it) or a script (-> its ranges in `multi`, `multi_neg` when negated)
    // general categories by name ("L", "Lu", "L&" ...) -> bit mask over the category ids; false: no such category
    static bool gc_mask_of(const std::string& name, uint32_t& mask) {
        mask = 0;
        if (name.empty()) return false;
        if (name.size() == 1 || name == "L&") {
            const char c0 = name[0];
            for (int g = 0; g < 30; ++g)
                if (kGc[g][0] == c0 && (name.size() == 1 || g == 1 || g == 2 || g == 3)) mask |= 1u << g;
            if (name == "L&") mask &= (1u << 1) | (1u << 2) | (1u << 3);
        } else {
            for (int g = 0; g < 30; ++g)
                if (name == kGc[g]) mask |= 1u << g;
        }
        return mask != 0;
    }
    bool read_property(bool upper_p, RxItem& it, std::vector<RxItem>* multi, bool* multi_neg) {
        std::string name;
        bool neg = upper_p;
        if (peek() == '{') {
            ++i;
            if (peek() == '^') { neg = !neg; ++i; }
            while (!eof() && peek() != '}') name.push_back((char)s[i++]);
            if (eof()) return fail("unterminated \\p{");
            ++i;
        } else if (!eof()) {
            name.push_back((char)s[i++]);
        }
        uint32_t mask = 0;
        (void)gc_mask_of(name, mask);
        it = none_item();
        if (mask) {
            it.gc_mask = mask;
            it.negate = neg ? 1 : 0;
            return true;
        }
        if (name == "Any") { it = range_item(0, 0x10FFFF); it.negate = neg ? 1 : 0; return true; }
        for (const TdScript& sc : td_scripts) {
            if (name != sc.name) continue;
            if (!multi) return fail("script property \\p{" + name + "} is not supported here");
            for (unsigned k = 0; k < sc.n; ++k) multi->push_back(range_item(sc.ranges[k].lo, sc.ranges[k].hi));
            if (multi_neg) *multi_neg = neg;
            return true;
        }
        return fail("unsupported Unicode property \\p{" + name + "} (general categories and the scripts of generated/unicode_scripts.inc)");
    }
    // behind a backslash: a class escape (-> item, is_class = true) or a literal character (-> cp)
    bool read_escape(bool& is_class, RxItem& it, uint32_t& cp, std::vector<RxItem>* multi = nullptr, bool* multi_neg = nullptr) {
        if (eof()) return fail("pattern ends in a backslash");
        const int c = peek();
        ++i;
        is_class = true;
        it = none_item();
        switch (c) {
            case 's': it.flags = RX_F_S; return true;
            case 'S': it.flags = RX_F_S; it.negate = 1; return true;
            case 'w': it.flags = RX_F_W; return true;
            case 'W': it.flags = RX_F_W; it.negate = 1; return true;
            case 'd': it.flags = RX_F_D; return true;
            case 'D': it.flags = RX_F_D; it.negate = 1; return true;
            case 'p': return read_property(false, it, multi, multi_neg);
            case 'P': return read_property(true, it, multi, multi_neg);
            case 'h': case 'H': case 'N': case 'v': case 'V': {  // horizontal / vertical white space (PCRE2's lists) / anything but a newline, as ranges
                if (!multi || !multi_neg) return fail(std::string("unsupported escape \\") + (char)c + " here");
                static const uint32_t H_RANGES[][2] = {{0x09, 0x09}, {0x20, 0x20}, {0xA0, 0xA0}, {0x1680, 0x1680}, {0x180E, 0x180E}, {0x2000, 0x200A},
                                                       {0x202F, 0x202F}, {0x205F, 0x205F}, {0x3000, 0x3000}};
                if (c == 'N') { multi->push_back(range_item('\n', '\n')); *multi_neg = true; return true; }
                if (c == 'v' || c == 'V') {
                    multi->push_back(range_item(0x0A, 0x0D)); multi->push_back(range_item(0x85, 0x85)); multi->push_back(range_item(0x2028, 0x2029));
                    *multi_neg = c == 'V';
                    return true;
                }
                for (auto& r : H_RANGES) multi->push_back(range_item(r[0], r[1]));
                *multi_neg = c == 'H';
                return true;
            }
            default: break;
        }
        is_class = false;
        switch (c) {
            case 'n': cp = '\n'; return true;
            case 'r': cp = '\r'; return true;
            case 't': cp = '\t'; return true;
            case 'f': cp = '\f'; return true;
            case 'a': cp = 0x07; return true;
            case 'e': cp = 0x1B; return true;
            case '0':  // PCRE2 reads up to two more octal digits (\012 = newline): not supported, and never NUL + literal digits
                if (peek() >= '0' && peek() <= '9') return fail("octal escapes (\\0 followed by a digit) are not supported");
                cp = 0; return true;
            case 'x': return read_hex(cp);
            default: break;
        }
        if (c == 'R' || c == 'X' || c == 'b' || c == 'B' || c == 'A' || c == 'Z' ||
            c == 'z' || c == 'G' || c == 'K' || c == 'Q' || c == 'E' || c == 'k' || c == 'g' || c == 'c' || c == 'o' || c == 'u' || (c >= '1' && c <= '9'))
            return fail(std::string("unsupported escape \\") + (char)c);
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) return fail(std::string("unsupported escape \\") + (char)c);
        if (c >= 0x80) { --i; return read_codepoint_literal(cp); }
        cp = (uint32_t)c;  // escaped punctuation
        return true;
    }

    // "[...]" (the '[' is consumed): the class; when `chars` is given and the class is a plain positive set of ASCII
    // characters, they are listed there too (for the expansion inside literal groups)
    bool read_bracket(uint16_t& cls, std::vector<uint32_t>* chars) {
        bool neg = false, plain = true, has_escape = false;
        if (peek() == '^') { neg = true; ++i; }
        const uint16_t first = (uint16_t)P.n_items;
        uint16_t n = 0, idx;
        bool first_char = true;
        for (;;) {
            if (eof()) return fail("unterminated character class");
            int c = peek();
            if (c == ']' && !first_char) { ++i; break; }
            first_char = false;
            if (c == '[' && peek(1) == ':') {  // POSIX class: what PCRE2_UCP makes of it (pcre2pattern: "[:alpha:] becomes \p{L}" ...)
                const size_t e = s.find(":]", i + 2);
                if (e == std::string::npos) return fail("unterminated POSIX class");
                std::string name = s.substr(i + 2, e - (i + 2));
                bool pneg = false;
                if (!name.empty() && name[0] == '^') { pneg = true; name.erase(0, 1); }
                RxItem pit = none_item();
                auto gc = [&](std::initializer_list<const char*> cats) {
                    uint32_t m = 0;
                    for (const char* cname : cats) { uint32_t one = 0; if (gc_mask_of(cname, one)) m |= one; }
                    return m;
                };
                if (name == "alpha") pit.gc_mask = gc({"L"});
                else if (name == "lower") pit.gc_mask = gc({"Ll"});
                else if (name == "upper") pit.gc_mask = gc({"Lu"});
                else if (name == "digit") pit.flags = RX_F_D;
                else if (name == "alnum") pit.gc_mask = gc({"L", "N"});
                else if (name == "space") pit.flags = RX_F_S;
                else if (name == "word") pit.flags = RX_F_W;
                else if (name == "cntrl") pit.gc_mask = gc({"Cc"});
                else return fail("POSIX class [:" + name + ":] is not supported");
                if (pit.gc_mask == 0 && pit.flags == 0) return fail("POSIX class [:" + name + ":] is not supported");
                pit.negate = pneg ? 1 : 0;
                i = e + 2;
                plain = false;
                has_escape = true;
                if (!new_item(pit, idx)) return false;
                ++n;
                continue;
            }
            uint32_t lo;
            RxItem it;
            bool is_class = false;
            std::vector<RxItem> multi;
            bool multi_neg = false;
            if (c == '\\') {
                ++i;
                if (!read_escape(is_class, it, lo, &multi, &multi_neg)) return false;
            } else if (!read_codepoint_literal(lo)) {
                return false;
            }
            if (is_class) {
                plain = false;
                has_escape = true;
                if (!multi.empty()) {  // a script: its ranges
                    if (multi_neg) return fail("a negated script property inside a character class is not supported");
                    for (const RxItem& m : multi) { if (!new_item(m, idx)) return false; ++n; }
                    if (peek() == '-' && peek(1) != ']' && peek(1) != -1) return fail("a class escape cannot start a range");
                    continue;
                }
                if (!new_item(it, idx)) return false;
                ++n;
                // PCRE2 refuses "[\\d-z]" (error 150: invalid range in character class), so the reference cannot be built from it
                if (peek() == '-' && peek(1) != ']' && peek(1) != -1) return fail("a class escape cannot start a range");
                continue;
            }
            uint32_t hi = lo;
            if (peek() == '-' && peek(1) != ']' && peek(1) != -1) {
                ++i;
                bool hc = false;
                RxItem dummy;
                if (peek() == '\\') {
                    ++i;
                    if (!read_escape(hc, dummy, hi)) return false;
                    if (hc) return fail("a class escape cannot end a range");
                } else if (!read_codepoint_literal(hi)) {
                    return false;
                }
                if (hi < lo) return fail("range out of order in character class");
            }
            if (!new_item(range_item(lo, hi), idx)) return false;
            ++n;
            if (chars) {
                if (hi - lo > 16 || hi >= 0x80u) plain = false;
                else for (uint32_t v = lo; v <= hi; ++v) chars->push_back(v);
            }
        }
        if (n == 0) return fail("empty character class");
        if (chars && (!plain || neg)) chars->clear();
        if (!new_class(first, n, neg, cls)) return false;
        P.classes[cls].pad[0] = has_escape ? 1 : 0;  // (compile-time note: \d \p{..} [:name:] ... inside, see rx_compile)
        return true;
    }

    // quantifier behind a class atom
    bool read_quantifier(RxNode& nd) {
        nd.min = 1; nd.max = 1; nd.possessive = 0;
        const int c = peek();
        if (c == '?') { nd.min = 0; nd.max = 1; ++i; }
        else if (c == '*') { nd.min = 0; nd.max = (uint16_t)RX_INF; ++i; }
        else if (c == '+') { nd.min = 1; nd.max = (uint16_t)RX_INF; ++i; }
        else if (c == '{' && peek(1) >= '0' && peek(1) <= '9') {
            size_t j = i + 1;
            uint32_t a = 0, b = 0;
            bool comma = false, have_b = false;
            while (j < s.size() && s[j] >= '0' && s[j] <= '9') { a = a * 10 + (uint32_t)(s[j] - '0'); if (a > 60000) return fail("repeat count too large"); ++j; }
            if (j < s.size() && s[j] == ',') {
                comma = true; ++j;
                while (j < s.size() && s[j] >= '0' && s[j] <= '9') { b = b * 10 + (uint32_t)(s[j] - '0'); have_b = true; if (b > 60000) return fail("repeat count too large"); ++j; }
            }
            if (j >= s.size() || s[j] != '}') return fail("malformed {m,n} quantifier");
            i = j + 1;
            nd.min = (uint16_t)a;
            nd.max = (uint16_t)(!comma ? a : have_b ? b : RX_INF);
            if (nd.max < nd.min) return fail("{m,n} with n < m");
        } else {
            return true;
        }
        if (peek() == '+') { nd.possessive = RX_POSSESSIVE; ++i; }
        else if (peek() == '?') { nd.possessive = RX_LAZY; ++i; }
        return true;
    }

    bool push_node(const RxNode& nd, uint32_t alt_first) {
        if (P.n_nodes >= (uint32_t)RX_MAX_NODES) return fail("split pattern too large (nodes)");
        if (P.n_nodes - alt_first >= (uint32_t)RX_MAX_SEQ) return fail("an alternative of the split pattern has too many elements");
        P.nodes[P.n_nodes++] = nd;
        return true;
    }
    static void utf8_append(std::string& o, uint32_t cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }

    // "(?:" / "(?i:" group of literal alternatives (the opening is consumed) -> one RX_LITSET node
    bool read_literal_group(bool caseless, uint32_t alt_first) {
        std::vector<std::string> alts(1);  // the alternatives in order; a bracket of plain ASCII characters multiplies the current one
        std::vector<std::string> done;
        bool had_bracket = false;          // (compile-time note for rx_compile's auto-possessification pass)
        for (;;) {
            if (eof()) return fail("unterminated group");
            const int c = peek();
            if (c == ')') { ++i; break; }
            if (c == '|') { ++i; for (auto& a : alts) done.push_back(a); alts.assign(1, std::string()); continue; }
            if (c == '(' || c == '*' || c == '+' || c == '?' || c == '{' || c == '.' || c == '^' || c == '$')
                return fail("only literal alternatives are supported inside a group");
            if (c == '[') {
                ++i;
                std::vector<uint32_t> chars;
                uint16_t cls;
                const uint32_t items0 = P.n_items, classes0 = P.n_classes;
                if (!read_bracket(cls, &chars)) return false;
                P.n_items = items0; P.n_classes = classes0;  // (only its characters are used)
                if (chars.empty()) return fail("inside a group only classes of a few ASCII characters are supported");
                if (chars.size() > 1) had_bracket = true;  // (PCRE2 compiles a bracket of ONE character to that character)
                std::vector<std::string> next;
                for (auto& a : alts)
                    for (uint32_t ch : chars) { next.push_back(a); utf8_append(next.back(), ch); }
                if (next.size() > 64) return fail("group expands to too many literals");
                alts.swap(next);
                continue;
            }
            uint32_t cp;
            if (c == '\\') {
                ++i;
                bool is_class;
                RxItem it;
                if (!read_escape(is_class, it, cp)) return false;
                if (is_class) return fail("only literal alternatives are supported inside a group");
            } else if (!read_codepoint_literal(cp)) {
                return false;
            }
            for (auto& a : alts) utf8_append(a, cp);
        }
        for (auto& a : alts) done.push_back(a);
        {   // a REPEATED group whose alternatives are single characters is a repeated class: (?:a|b|[cd])* == [abcd]*
            const int q = peek();
            if (q == '*' || q == '+' || (q == '{' && peek(1) >= '0' && peek(1) <= '9')) {
                std::vector<uint32_t> cps;
                for (auto& a : done) {
                    if (a.empty()) return fail("empty alternative inside a group");
                    uint32_t cp = (unsigned char)a[0];
                    size_t need = cp < 0x80 ? 0 : cp < 0xE0 ? 1 : cp < 0xF0 ? 2 : 3;
                    if (a.size() != need + 1) return fail("repeated groups are supported for alternatives of single characters only");
                    if (need) cp &= 0xFFu >> (need + 2);
                    for (size_t k = 1; k <= need; ++k) cp = (cp << 6) | ((unsigned char)a[k] & 0x3Fu);
                    cps.push_back(cp);
                    if (caseless) {
                        if (cp >= 0x80) return fail("case-insensitive groups support ASCII literals only");
                        const uint32_t lc = cp | 0x20u;
                        if (lc >= 'a' && lc <= 'z') {
                            cps.push_back(lc); cps.push_back(lc ^ 0x20u);
                            if (lc == 's') cps.push_back(0x17Fu);   // (as the literal matcher: PCRE2_UCP folds these two onto s / k)
                            if (lc == 'k') cps.push_back(0x212Au);
                        }
                    }
                }
                uint16_t first = 0, idx = 0;
                for (size_t k = 0; k < cps.size(); ++k) {
                    if (!new_item(range_item(cps[k], cps[k]), idx)) return false;
                    if (k == 0) first = idx;
                }
                RxNode nd{};
                nd.kind = RX_CLASS;
                if (!new_class(first, (uint16_t)cps.size(), false, nd.a)) return false;
                if (!read_quantifier(nd)) return false;
                nd.pad = 1;  // (compile-time note: a group in the pattern — PCRE2 never auto-possessifies it, see rx_compile)
                return push_node(nd, alt_first);
            }
        }
        RxNode nd{};
        nd.kind = RX_LITSET;
        nd.pad = had_bracket ? 1 : 0;
        nd.caseless = caseless ? 1 : 0;
        nd.a = (uint16_t)P.n_lits;
        nd.b = (uint16_t)done.size();
        nd.min = 1; nd.max = 1;
        for (auto& a : done) {
            if (a.empty()) return fail("empty alternative inside a group");
            if (P.n_lits >= (uint32_t)RX_MAX_LITS || P.n_litbytes + a.size() > (size_t)RX_MAX_LITBYTES) return fail("split pattern too large (literals)");
            if (caseless)
                for (unsigned char ch : a)
                    if (ch >= 0x80) return fail("case-insensitive groups support ASCII literals only");
            RxLit l;
            l.off = (uint16_t)P.n_litbytes; l.len = (uint16_t)a.size();
            for (unsigned char ch : a) P.litbytes[P.n_litbytes++] = ch;
            P.lits[P.n_lits++] = l;
        }
        if (peek() == '?') {
            nd.min = 0; ++i;
            if (peek() == '+') { nd.possessive = RX_POSSESSIVE; ++i; }  // atomic: once a literal (or the skip) is chosen the matcher never comes back for another
            else if (peek() == '?') { nd.possessive = RX_LAZY; ++i; }   // lazy: the skip first
        }
        return push_node(nd, alt_first);
    }

    // one class atom at the cursor -> class id (for look-aheads and sequences)
    bool read_class_atom(uint16_t& cls) {
        const int c = peek();
        if (c == '[') { ++i; return read_bracket(cls, nullptr); }
        if (c == '.') {
            ++i;
            uint16_t idx = 0;
            return new_item(range_item('\n', '\n'), idx) && new_class(idx, 1, true, cls);
        }
        uint32_t cp;
        if (c == '\\') {
            ++i;
            bool is_class, multi_neg = false;
            RxItem it;
            std::vector<RxItem> multi;
            if (!read_escape(is_class, it, cp, &multi, &multi_neg)) return false;
            if (!multi.empty()) {  // a script on its own: a class of its ranges
                const uint16_t first = (uint16_t)P.n_items;
                uint16_t idx = 0;
                for (const RxItem& m : multi)
                    if (!new_item(m, idx)) return false;
                if (!new_class(first, (uint16_t)multi.size(), multi_neg, cls)) return false;
                P.classes[cls].pad[0] = 1;
                return true;
            }
            if (!single_class(is_class ? it : range_item(cp, cp), cls)) return false;
            P.classes[cls].pad[0] = is_class ? 1 : 0;
            return true;
        }
        if (!read_codepoint_literal(cp)) return false;
        return single_class(range_item(cp, cp), cls);
    }

    bool flush_literal(std::string& lit, uint32_t alt_first) {
        if (lit.empty()) return true;
        if (P.n_lits >= (uint32_t)RX_MAX_LITS || P.n_litbytes + lit.size() > (size_t)RX_MAX_LITBYTES) return fail("split pattern too large (literals)");
        RxNode nd{};
        nd.kind = RX_LITSET;
        nd.a = (uint16_t)P.n_lits;
        nd.b = 1;
        nd.min = 1; nd.max = 1;
        RxLit l;
        l.off = (uint16_t)P.n_litbytes; l.len = (uint16_t)lit.size();
        for (unsigned char ch : lit) P.litbytes[P.n_litbytes++] = ch;
        P.lits[P.n_lits++] = l;
        lit.clear();
        return push_node(nd, alt_first);
    }

    bool read_alternative() {
        if (P.n_alts >= (uint32_t)RX_MAX_ALTS) return fail("split pattern has too many alternatives");
        const uint32_t first = P.n_nodes;
        std::string pending;  // literal characters read so far that no node holds yet
        while (!eof() && peek() != '|') {
            const int c = peek();
            if (c == ')') return fail("unbalanced parenthesis");
            {   // zero-width assertions: ^ \A \z \Z \b \B
                int zk = -1;
                size_t adv = 0;
                if (c == '^') { zk = RX_BOS; adv = 1; }
                else if (c == '\\') {
                    const int e1 = peek(1);
                    if (e1 == 'A') zk = RX_BOS; else if (e1 == 'z') zk = RX_EOS_STRICT; else if (e1 == 'Z') zk = RX_EOS;
                    else if (e1 == 'b') zk = RX_WORDB; else if (e1 == 'B') zk = RX_NWORDB;
                    adv = 2;
                }
                if (zk >= 0) {
                    if (!flush_literal(pending, first)) return false;
                    i += adv;
                    const int q = peek();
                    if (q == '?' || q == '*' || q == '+' || (q == '{' && peek(1) >= '0' && peek(1) <= '9')) return fail("a quantifier on an assertion is not supported");
                    RxNode nd{};
                    nd.kind = (uint8_t)zk;
                    if (!push_node(nd, first)) return false;
                    continue;
                }
            }
            if (c == '*' || c == '+' || c == '?' || (c == '{' && peek(1) >= '0' && peek(1) <= '9')) return fail("quantifier without an atom");
            if (c == '$' || c == '(') {
                if (!flush_literal(pending, first)) return false;
            }
            if (c == '$') {
                ++i;
                RxNode nd{};
                nd.kind = RX_EOS;
                if (!push_node(nd, first)) return false;
                continue;
            }
            if (c == '(') {
                if (peek(1) != '?') return fail("capturing groups are not supported");
                if (peek(2) == ':' ) { i += 3; if (!read_literal_group(false, first)) return false; continue; }
                if (peek(2) == 'i' && peek(3) == ':') { i += 4; if (!read_literal_group(true, first)) return false; continue; }
                const bool behind = peek(2) == '<' && (peek(3) == '!' || peek(3) == '=');
                if (peek(2) == '!' || peek(2) == '=' || behind) {
                    const bool negative = peek(behind ? 3 : 2) == '!';
                    i += behind ? 4 : 3;
                    RxNode nd{};
                    nd.kind = behind ? (negative ? RX_NLOOKB : RX_PLOOKB) : (negative ? RX_NLOOK : RX_PLOOK);
                    if (!read_class_atom(nd.a)) return false;
                    if (peek() != ')') return fail("a look-ahead or look-behind may hold one character class only");
                    ++i;
                    if (!push_node(nd, first)) return false;
                    continue;
                }
                return fail("unsupported group syntax");
            }
            // a literal character without a quantifier joins the literal run in front of it (one node for "0x", not a class per
            // character)
            if (c != '[' && c != '.') {
                const size_t i0 = i;
                const uint32_t items0 = P.n_items, classes0 = P.n_classes;
                uint32_t cp = 0;
                bool is_class = false, ok = true;
                if (c == '\\') {
                    ++i;
                    RxItem it;
                        std::vector<RxItem> multi;
                    bool multi_neg = false;
                    ok = read_escape(is_class, it, cp, &multi, &multi_neg);
                } else {
                    ok = read_codepoint_literal(cp);
                }
                if (!ok) return false;
                const int q = peek();
                const bool quantified = q == '?' || q == '*' || q == '+' || (q == '{' && peek(1) >= '0' && peek(1) <= '9');
                if (!is_class && !quantified) { utf8_append(pending, cp); continue; }
                i = i0; P.n_items = items0; P.n_classes = classes0;  // (read again below as a class atom)
            }
            if (!flush_literal(pending, first)) return false;
            RxNode nd{};
            nd.kind = RX_CLASS;
            if (!read_class_atom(nd.a)) return false;
            if (!read_quantifier(nd)) return false;
            if (!push_node(nd, first)) return false;
        }
        if (!flush_literal(pending, first)) return false;
        if (P.n_nodes == first) return fail("empty alternative in the split pattern");
        RxAlt a;
        a.first_node = (uint16_t)first; a.n_nodes = (uint16_t)(P.n_nodes - first);
        P.alts[P.n_alts++] = a;
        return true;
    }
};

}  // namespace

bool rx_compile(const std::string& pattern, RxProgram& P, std::string& err) {
    P = RxProgram();
    Reader R(pattern, P);
    if (pattern.empty()) { err = "empty split pattern"; return false; }
    for (;;) {
        if (!R.read_alternative()) { err = R.err; return false; }
        if (R.eof()) break;
        ++R.i;  // '|'
        if (R.eof()) { err = "split pattern ends in '|'"; return false; }
    }
    // membership of the ASCII code points, once (the matcher tests one bit for them)
    const RxTables T = rx_host_tables();
    for (uint32_t c = 0; c < P.n_classes; ++c) {
        uint32_t m[4] = {0, 0, 0, 0};
        for (uint32_t cp = 0; cp < 128u; ++cp)
            if (rx_in_class_slow(P, T, c, cp)) m[cp >> 5] |= 1u << (cp & 31u);
        for (int k = 0; k < 4; ++k) P.classes[c].ascii[k] = m[k];
    }
    // PCRE2's auto-possessification, where it changes results.  PCRE2 makes a greedy quantifier possessive when the item
    // behind it cannot start with a character the quantified item matches — harmless, except for one case it gets wrong
    // (observed with the PCRE2 the reference links, 10.4x; probed in tests/test_generic_pattern.py): behind a quantified
    // class it looks INTO a possessive optional group "(?:..)?+" and not past it, so "a+(?:q)?+a" never gives an 'a' back
    // and does not match "aaa".  The reference is what PCRE2 does: a class with a greedy quantifier that can give back,
    // directly in front of such a group none of whose literals starts with a member of the class, is possessive here too
    // (PCRE2's interpreter and its JIT agree on this, oracle/pcre2_probe.c; the decision was probed class kind by class kind
    // and group kind by group kind, 868 combinations in tests/test_generic_pattern.py).  A class that stands for a repeated
    // group of the pattern ((?:a|b)+) is left alone: PCRE2 does not auto-possessify groups.
    for (uint32_t a = 0; a < P.n_alts; ++a) {
        for (uint32_t k = 0; k + 1 < P.alts[a].n_nodes; ++k) {
            RxNode& nd = P.nodes[P.alts[a].first_node + k];
            const RxNode& nx = P.nodes[P.alts[a].first_node + k + 1];
            if (nd.kind != RX_CLASS || nd.possessive != RX_GREEDY || nd.min == nd.max || nd.pad) continue;
            if (nx.kind != RX_LITSET || nx.possessive != RX_POSSESSIVE || nx.min != 0) continue;
            // (a group with a bracket among its alternatives: PCRE2 compares only a class written without class escapes,
            // properties and POSIX names with it — probed base by base, tests/test_generic_pattern.py)
            if (nx.pad && P.classes[nd.a].pad[0]) continue;
            bool touches = false;
            for (uint32_t l = 0; l < nx.b && !touches; ++l) {
                const RxLit lit = P.lits[nx.a + l];
                const uint8_t* b = P.litbytes + lit.off;
                uint32_t cp = b[0];
                const uint32_t need = cp < 0x80 ? 0 : cp < 0xE0 ? 1 : cp < 0xF0 ? 2 : 3;
                if (need) cp &= 0xFFu >> (need + 2);
                for (uint32_t q = 1; q <= need && q < lit.len; ++q) cp = (cp << 6) | (b[q] & 0x3Fu);
                touches = rx_in_class_slow(P, T, nd.a, cp);
                if (!touches && nx.caseless && cp < 0x80) {
                    const uint32_t lc = cp | 0x20u;
                    if (lc >= 'a' && lc <= 'z') {
                        touches = rx_in_class_slow(P, T, nd.a, lc) || rx_in_class_slow(P, T, nd.a, lc ^ 0x20u) ||
                                  (lc == 's' && rx_in_class_slow(P, T, nd.a, 0x17Fu)) || (lc == 'k' && rx_in_class_slow(P, T, nd.a, 0x212Au));
                    }
                }
            }
            if (!touches) nd.possessive = RX_POSSESSIVE;
        }
    }
    // which alternatives can start with which ASCII character (a superset).  first(i) of an alternative's node i: a class
    // gives its members (and what follows, if it may be empty), a group of literals their first bytes (both cases when
    // caseless; and what follows, if it is optional), a zero-width assertion only ever restricts what follows, and the end
    // of the alternative stands for "anything".
    for (uint32_t c = 0; c < 128u; ++c) P.first_alts[c] = 0;
    for (uint32_t a = 0; a < P.n_alts; ++a) {
        uint32_t m[4] = {0, 0, 0, 0};
        bool open = true;  // the nodes so far may all match empty
        for (uint32_t k = 0; k < P.alts[a].n_nodes && open; ++k) {
            const RxNode& nd = P.nodes[P.alts[a].first_node + k];
            if (nd.kind == RX_CLASS) {
                for (int q = 0; q < 4; ++q) m[q] |= P.classes[nd.a].ascii[q];
                open = nd.min == 0;
            } else if (nd.kind == RX_LITSET) {
                for (uint32_t l = 0; l < nd.b; ++l) {
                    const uint32_t ch = P.litbytes[P.lits[nd.a + l].off];
                    if (ch >= 128u) continue;
                    m[ch >> 5] |= 1u << (ch & 31u);
                    if (nd.caseless) {
                        const uint32_t lc = ch | 0x20u;
                        if (lc >= 'a' && lc <= 'z') { m[(lc ^ 0x20u) >> 5] |= 1u << ((lc ^ 0x20u) & 31u); m[lc >> 5] |= 1u << (lc & 31u); }
                    }
                }
                open = nd.min == 0;
            }  // (assertions: zero-width, `open` stays)
        }
        if (open) m[0] = m[1] = m[2] = m[3] = 0xFFFFFFFFu;
        for (uint32_t c = 0; c < 128u; ++c)
            if ((m[c >> 5] >> (c & 31u)) & 1u) P.first_alts[c] |= 1u << a;
    }
    return true;
}

}  // namespace td
