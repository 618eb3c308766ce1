// Vocabulary file readers (see td_vocab.h).  Self-contained: a small JSON reader, base64, and the loaders.
#include "td_vocab.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>

namespace td {

// ------------------------------------------------------------------ file + base64 -----------
static bool read_file(const std::string& path, std::string& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path + ": " + strerror(errno);
        return false;
    }
    out.clear();
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, got);
    const bool bad = ferror(f) != 0;
    fclose(f);
    if (bad) err = "read error on " + path;
    return !bad;
}

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    static int8_t T[256];
    static bool init = false;
    if (!init) {
        memset(T, -1, sizeof T);
        const char* abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) T[(uint8_t)abc[i]] = (int8_t)i;
        init = true;
    }
    while (n && s[n - 1] == '=') --n;  // padding
    if (n % 4 == 1) return false;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const int v = T[(uint8_t)s[i]];
        if (v < 0) return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return true;
}

// ------------------------------------------------------------------ JSON --------------------
// A plain recursive-descent reader into a small tree; enough for tokenizer metadata files (objects keep insertion
// order, numbers are kept as text and converted on demand, strings are UTF-8 with \uXXXX escapes resolved).
namespace {
struct JVal;
using JPtr = std::unique_ptr<JVal>;
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    std::string s;  // Str: the text; Num: the literal
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;

    const JVal* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    bool as_int(int64_t& out) const {
        if (kind == Num || kind == Str) {
            if (s.empty()) return false;
            char* end = nullptr;
            errno = 0;
            const long long v = strtoll(s.c_str(), &end, 10);
            if (errno || end == s.c_str()) return false;
            if (*end == '.' || *end == 'e' || *end == 'E') {  // 1.0e3 style: accept when integral
                const double d = strtod(s.c_str(), &end);
                if (*end || d != (double)(long long)d) return false;
                out = (long long)d;
                return true;
            }
            if (*end) return false;
            out = v;
            return true;
        }
        return false;
    }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;

    bool fail(const char* what) {
        if (err.empty()) err = std::string("JSON: ") + what + " at byte " + std::to_string((long long)(p - begin));
        return false;
    }
    const char* begin;
    JParser(const std::string& text) : p(text.data()), end(text.data() + text.size()), begin(text.data()) {}

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    static void put_utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t& v) {
        if (end - p < 4) return fail("short \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        while (p < end) {
            const char c = *p++;
            if (c == '"') return true;
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= end) break;
            const char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        const char* save = p;
                        p += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else p = save;  // lone high surrogate: keep it as is, the next escape is parsed on its own
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool value(JVal& v) {
        if (++depth > 256) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            ++p;
            v.kind = JVal::Obj;
            ws();
            if (p < end && *p == '}') ++p;
            else
                for (;;) {
                    ws();
                    std::string key;
                    if (!string(key)) { ok = false; break; }
                    ws();
                    if (p >= end || *p != ':') { ok = fail("expected ':'"); break; }
                    ++p;
                    JPtr child(new JVal);
                    if (!value(*child)) { ok = false; break; }
                    v.obj.emplace_back(std::move(key), std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
        } else if (c == '[') {
            ++p;
            v.kind = JVal::Arr;
            ws();
            if (p < end && *p == ']') ++p;
            else
                for (;;) {
                    JPtr child(new JVal);
                    if (!value(*child)) { ok = false; break; }
                    v.arr.push_back(std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
        } else if (c == '"') {
            v.kind = JVal::Str;
            ok = string(v.s);
        } else if (c == 't' && end - p >= 4 && !memcmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; }
        else if (c == 'f' && end - p >= 5 && !memcmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; }
        else if (c == 'n' && end - p >= 4 && !memcmp(p, "null", 4)) { v.kind = JVal::Null; p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s0 = p;
            ++p;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) ++p;
            v.kind = JVal::Num;
            v.s.assign(s0, p);
        } else {
            ok = fail("unexpected character");
        }
        --depth;
        return ok;
    }
};

bool parse_json_file(const std::string& path, JVal& root, std::string& err) {
    std::string text;
    if (!read_file(path, text, err)) return false;
    JParser ps(text);
    if (text.size() >= 3 && (uint8_t)text[0] == 0xEF && (uint8_t)text[1] == 0xBB && (uint8_t)text[2] == 0xBF) ps.p += 3;  // BOM
    if (!ps.value(root)) { err = path + ": " + ps.err; return false; }
    ps.ws();
    if (ps.p != ps.end) { err = path + ": JSON: trailing characters"; return false; }
    return true;
}
}  // namespace

// ------------------------------------------------------------------ loaders -----------------
bool load_tiktoken_model(const std::string& path, VocabData& v) {
    std::string text;
    if (!read_file(path, text, v.err)) return false;
    std::vector<uint8_t> tok;
    size_t pos = 0, line_no = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        size_t a = pos, b = eol;
        pos = eol + 1;
        ++line_no;
        while (a < b && (text[a] == ' ' || text[a] == '\t' || text[a] == '\r')) ++a;
        while (b > a && (text[b - 1] == ' ' || text[b - 1] == '\t' || text[b - 1] == '\r')) --b;
        if (a == b) continue;  // blank line
        size_t sp = a;
        while (sp < b && text[sp] != ' ' && text[sp] != '\t') ++sp;
        size_t r0 = sp;
        while (r0 < b && (text[r0] == ' ' || text[r0] == '\t')) ++r0;
        auto bad = [&](const char* what) {
            v.err = path + ":" + std::to_string(line_no) + ": " + what;
            return false;
        };
        if (r0 == b) return bad("expected '<base64> <rank>'");
        tok.clear();
        if (!base64_decode(text.data() + a, sp - a, tok) || tok.empty()) return bad("invalid base64 token");
        char* end = nullptr;
        errno = 0;
        const std::string num(text, r0, b - r0);
        const long long rank = strtoll(num.c_str(), &end, 10);
        if (errno || end == num.c_str() || *end || rank < 0 || rank > 0x7FFFFFFF) return bad("invalid rank");
        v.regular.add(tok.data(), tok.size(), (int32_t)rank);
    }
    return true;
}

bool load_hf_added_tokens(const std::string& path, VocabData& v, bool also_mergeable) {
    JVal root;
    if (!parse_json_file(path, root, v.err)) return false;
    const JVal* dec = root.get("added_tokens_decoder");
    if (!dec) return true;  // a config without added tokens is valid: no special tokens
    if (dec->kind != JVal::Obj) { v.err = path + ": added_tokens_decoder is not an object"; return false; }
    for (const auto& kv : dec->obj) {
        char* end = nullptr;
        errno = 0;
        const long long id = strtoll(kv.first.c_str(), &end, 10);
        if (errno || end == kv.first.c_str() || *end || id < 0 || id > 0x7FFFFFFF) {
            v.err = path + ": added_tokens_decoder key '" + kv.first + "' is not a token id";
            return false;
        }
        const JVal* content = kv.second->get("content");
        if (!content || content->kind != JVal::Str || content->s.empty()) {
            v.err = path + ": added token " + kv.first + " has no content string";
            return false;
        }
        v.special.add((const uint8_t*)content->s.data(), content->s.size(), (int32_t)id);
        if (also_mergeable) v.regular.add((const uint8_t*)content->s.data(), content->s.size(), (int32_t)id);
    }
    return true;
}

bool load_tekken_json(const std::string& path, VocabData& v) {
    JVal root;
    if (!parse_json_file(path, root, v.err)) return false;
    const JVal* cfg = root.get("config");
    const JVal* vocab = root.get("vocab");
    if (!cfg || cfg->kind != JVal::Obj || !vocab || vocab->kind != JVal::Arr) {
        v.err = path + ": not a tekken file (no config / vocab)";
        return false;
    }
    const JVal* pat = cfg->get("pattern");
    int64_t n_total = 0, n_special = 0;
    if (!pat || pat->kind != JVal::Str || !cfg->get("default_vocab_size") || !cfg->get("default_vocab_size")->as_int(n_total) ||
        !cfg->get("default_num_special_tokens") || !cfg->get("default_num_special_tokens")->as_int(n_special) || n_special < 0 ||
        n_total < n_special) {
        v.err = path + ": config.pattern / default_vocab_size / default_num_special_tokens missing or invalid";
        return false;
    }
    const int64_t n = n_total - n_special;
    if ((int64_t)vocab->arr.size() < n) {
        v.err = path + ": vocab has " + std::to_string(vocab->arr.size()) + " entries, config asks for " + std::to_string((long long)n);
        return false;
    }
    v.pattern = pat->s;
    std::vector<uint8_t> tok;
    for (int64_t i = 0; i < n; ++i) {
        const JVal* tb = vocab->arr[(size_t)i]->get("token_bytes");
        tok.clear();
        if (!tb || tb->kind != JVal::Str || !base64_decode(tb->s.data(), tb->s.size(), tok) || tok.empty()) {
            v.err = path + ": vocab[" + std::to_string((long long)i) + "].token_bytes is not base64";
            return false;
        }
        v.regular.add(tok.data(), tok.size(), (int32_t)(i + n_special));
    }
    return true;
}

bool load_wrapper_json(const std::string& vocab_path, const std::string& special_path, VocabData& v) {
    if (!vocab_path.empty()) {
        JVal root;
        if (!parse_json_file(vocab_path, root, v.err)) return false;
        if (root.kind != JVal::Arr) { v.err = vocab_path + ": expected a list of vocabulary items"; return false; }
        std::vector<uint8_t> tok;
        for (size_t i = 0; i < root.arr.size(); ++i) {
            const JVal* rank = root.arr[i]->get("rank");
            const JVal* tb = root.arr[i]->get("token_bytes");
            int64_t r = 0;
            if (!rank || !rank->as_int(r) || r < 0 || r > 0x7FFFFFFF || !tb || tb->kind != JVal::Arr || tb->arr.empty()) {
                v.err = vocab_path + ": item " + std::to_string(i) + " needs 'rank' and a non-empty 'token_bytes' list";
                return false;
            }
            tok.clear();
            for (const auto& e : tb->arr) {
                int64_t b = -1;
                if (!e->as_int(b) || b < 0 || b > 255) { v.err = vocab_path + ": item " + std::to_string(i) + ": token_bytes entries must be 0..255"; return false; }
                tok.push_back((uint8_t)b);
            }
            v.regular.add(tok.data(), tok.size(), (int32_t)r);
        }
    }
    if (!special_path.empty()) {
        JVal root;
        if (!parse_json_file(special_path, root, v.err)) return false;
        if (root.kind != JVal::Obj) { v.err = special_path + ": expected an object {token string: id}"; return false; }
        for (const auto& kv : root.obj) {
            int64_t id = 0;
            if (kv.first.empty() || !kv.second->as_int(id) || id < 0 || id > 0x7FFFFFFF) {
                v.err = special_path + ": special token '" + kv.first + "' needs a non-negative integer id";
                return false;
            }
            v.special.add((const uint8_t*)kv.first.data(), kv.first.size(), (int32_t)id);
        }
    }
    return true;
}

}  // namespace td
