// Host-side table construction.  See td_tables.h.
#include "td_tables.h"
#include <cstdio>
#include <cstdlib>
#include "td_regex.h"

#include <string.h>

#include <algorithm>
#include <map>

#include "generated/unicode_classes.inc"

namespace td {

static const char kO200k[] =
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    "|[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    "|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

// Mistral "tekken" (tekken.json config.pattern): the same seven alternatives without the contraction suffix and
// with single-digit number pieces.
static const char kTekken[] =
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+"
    "|[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*"
    "|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

// cl100k_base / Llama-3: the classic form and tiktoken's later possessive spelling of the same language.
static const char kCl100k[] =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
static const char kCl100kPossessive[] =
    "'(?i:[sdmt]|ll|ve|re)|[^\\r\\n\\p{L}\\p{N}]?+\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]++[\\r\\n]*|\\s*[\\r\\n]|\\s+(?!\\S)|\\s+";

static const char kCl100kCurrent[] =  // cl100k_base as tiktoken ships it today (openai_public.py); NOT the same language: \s++$ comes first
    "'(?i:[sdmt]|ll|ve|re)|[^\\r\\n\\p{L}\\p{N}]?+\\p{L}++|\\p{N}{1,3}+| ?[^\\s\\p{L}\\p{N}]++[\\r\\n]*+|\\s++$|\\s*[\\r\\n]|\\s+(?!\\S)|\\s";

// Qwen2 / Qwen2.5 / Qwen3: cl100k_base's classic form with `\\p{N}` in place of `\\p{N}{1,3}`.
static const char kQwen2[] =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

static const char kGpt2[] =
    "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";

static const char kGpt2Possessive[] =  // tiktoken's later spelling of the same language
    "'(?:[sdmt]|ll|ve|re)| ?\\p{L}++| ?\\p{N}++| ?[^\\s\\p{L}\\p{N}]++|\\s++$|\\s+(?!\\S)|\\s";

const char* o200k_pattern() { return kO200k; }
const char* tekken_pattern() { return kTekken; }
const char* cl100k_pattern() { return kCl100k; }

PatternKind classify_pattern(const std::string& pat) {
    if (pat == kO200k) return PATTERN_O200K;
    if (pat == kTekken) return PATTERN_TEKKEN;
    if (pat == kCl100k || pat == kCl100kPossessive) return PATTERN_CL100K;
    if (pat == kCl100kCurrent) return PATTERN_CL100K_EOS;
    if (pat == kQwen2) return PATTERN_QWEN2;
    if (pat == kGpt2 || pat == kGpt2Possessive) return PATTERN_GPT2;
    return PATTERN_UNSUPPORTED;
}

uint32_t pattern_flags(PatternKind k) {
    if (k == PATTERN_TEKKEN) return PV_NO_CONTRACTION | PV_SINGLE_DIGIT;
    if (k == PATTERN_CL100K) return PV_NO_CONTRACTION | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS;
    if (k == PATTERN_CL100K_EOS) return PV_NO_CONTRACTION | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS | PV_WS_EOS_FIRST;
    if (k == PATTERN_QWEN2) return PV_NO_CONTRACTION | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS | PV_SINGLE_DIGIT;
    if (k == PATTERN_GPT2) return PV_GPT2;
    if (k == PATTERN_GENERIC) return PV_GENERIC;
    return 0u;
}

uint64_t piece_key_host(const uint8_t* p, uint32_t len) {
    if (len <= 8) {
        uint64_t k = 0;
        for (uint32_t i = 0; i < len; ++i) k |= (uint64_t)p[i] << (8 * i);
        return k;
    }
    return hash_bytes([p](uint32_t i) { return (uint32_t)p[i]; }, len);
}

Tables HostTables::view() const {
    Tables T;
    memset(&T, 0, sizeof T);
    T.ascii_cls = ascii_cls.data();
    T.ucls1 = td_ucls_stage1;
    T.ucls2 = ucls2_remap.empty() ? td_ucls_stage2 : ucls2_remap.data();
    T.byte_id = byte_id.data();
    T.byte_pair = byte_pair.data();
    T.byte_pair_id = byte_pair_id.data();
    T.piece_slots = piece_slots.data();
    T.piece12_slots = piece12_slots.data();
    T.piece12_mask = piece12_mask;
    T.pat_flags = pattern_flags(pattern_kind);
    T.pair_slots = pair_slots.data();
    T.tok_off = tok_off.data();
    T.tok_bytes = tok_bytes.data();
    T.piece_mask = piece_mask;
    T.pair_mask = pair_mask;
    T.max_id = max_id;
    T.pseudo_base = pseudo_base;
    T.max_token_len = max_token_len;
    T.cseed = cseed.empty() ? nullptr : cseed.data();
    T.cseed_pm = cseed_pm.empty() ? nullptr : cseed_pm.data();
    T.cseed_nm = cseed_nm.empty() ? nullptr : cseed_nm.data();
    return T;
}

static uint32_t pow2_at_least(uint64_t n) {
    uint32_t c = 16;
    while (c < n) c <<= 1;
    return c;
}

int merge_piece_host(const Tables& T, const uint8_t* piece, uint32_t n, std::vector<int32_t>& out) {
    // ids of the current parts + rank of (part i, part i+1); leftmost minimum merges first
    // (strict '<' scans in tiktoken.cpp:312,338).
    std::vector<int32_t> id(n), rk(n, NO_RANK);
    for (uint32_t i = 0; i < n; ++i) id[i] = T.byte_id[piece[i]];
    for (uint32_t i = 0; i + 1 < n; ++i) rk[i] = T.byte_pair[((uint32_t)piece[i] << 8) | piece[i + 1]];
    for (;;) {
        int32_t best = NO_RANK;
        size_t bi = 0;
        for (size_t i = 0; i + 1 < id.size(); ++i)
            if (rk[i] < best) { best = rk[i]; bi = i; }
        if (best == NO_RANK) break;
        id[bi] = best;
        id.erase(id.begin() + bi + 1);
        rk.erase(rk.begin() + bi + 1);
        rk[bi] = (bi + 1 < id.size()) ? pair_lookup(T, (uint32_t)id[bi], (uint32_t)id[bi + 1]) : NO_RANK;
        if (bi > 0) rk[bi - 1] = pair_lookup(T, (uint32_t)id[bi - 1], (uint32_t)id[bi]);
    }
    for (int32_t v : id) {
        if (v >= T.pseudo_base) return TD_E_UNKNOWN_BYTE;
        out.push_back(v);
    }
    return TD_OK;
}

// Character seeds (td_common.h: "character seeds"): which characters of 2 and 3 bytes may be entered into the merge loop as one
// part, and the bytes that must not stand next to them.  Conditions 1, 3 and 4 of the comment there are settled here per character,
// condition 2 becomes two 256-bit sets per character: pm = the bytes in front of c with which some token ends in a proper prefix of
// c, nm = the bytes behind c with which some token starts with a proper suffix of c.  TD_CHAR_SEEDS=0 in the environment: none.
namespace {
struct Bits256 {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void set(uint32_t b) { w[b >> 5] |= 1u << (b & 31); }
    void join(const Bits256& o) { for (int i = 0; i < 8; ++i) w[i] |= o.w[i]; }
    bool operator<(const Bits256& o) const { return memcmp(w, o.w, sizeof w) < 0; }
};
uint32_t key_of(const uint8_t* p, uint32_t n) {  // up to three bytes and their number
    uint32_t k = n;
    for (uint32_t i = 0; i < n; ++i) k = (k << 8) | p[i];
    return k;
}
}  // namespace
static void build_char_seeds(HostTables& H, int64_t n_vocab, const uint8_t* token_bytes, const int64_t* token_offsets, const int32_t* ranks,
                             const std::vector<std::pair<uint64_t, uint32_t>>& pairs) {
    H.cseed.clear(); H.cseed_pm.clear(); H.cseed_nm.clear(); H.n_char_seeds = 0;
    if (const char* e = getenv("TD_CHAR_SEEDS")) if (atoi(e) == 0) return;
    const Tables T = H.view();
    // candidates: the tokens that are one character of 2 or 3 bytes in its shortest form, whose bytes merge to exactly that token (1.)
    // and whose id is the rank (4.: always so for the regular vocabulary); rho = the highest rank among those merges
    struct Cand { uint32_t cp, k, id, rho; uint8_t b[3]; bool ok; };
    std::vector<Cand> cands;
    std::vector<int32_t> cand_of((size_t)H.max_id + 1, -1);
    for (int64_t v = 0; v < n_vocab; ++v) {
        const uint8_t* p = token_bytes + token_offsets[v];
        const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        if (len != 2 && len != 3) continue;
        uint32_t cp;
        if (len == 2) {
            if (p[0] < 0xC2 || p[0] >= 0xE0 || (p[1] & 0xC0) != 0x80) continue;
            cp = ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu);
        } else {
            if (p[0] < 0xE0 || p[0] >= 0xF0 || (p[1] & 0xC0) != 0x80 || (p[2] & 0xC0) != 0x80) continue;
            cp = ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu);
            if (cp < 0x800) continue;
        }
        if ((uint32_t)ranks[v] >= (1u << 21)) continue;
        // the merge loop on the character's bytes (tiktoken.cpp:322-343): lowest rank, leftmost
        uint32_t part[3], np = len, rho = 0;
        bool fail = false;
        for (uint32_t i = 0; i < len; ++i) part[i] = (uint32_t)H.byte_id[p[i]];
        while (np > 1) {
            int32_t best = NO_RANK; uint32_t bi = 0;
            for (uint32_t i = 0; i + 1 < np; ++i) {
                const int32_t r = pair_lookup(T, part[i], part[i + 1]);
                if (r < best) { best = r; bi = i; }
            }
            if (best == NO_RANK) { fail = true; break; }
            rho = std::max<uint32_t>(rho, (uint32_t)best);
            part[bi] = (uint32_t)best;
            for (uint32_t i = bi + 1; i + 1 < np; ++i) part[i] = part[i + 1];
            --np;
        }
        if (fail || part[0] != (uint32_t)ranks[v]) continue;
        Cand c{cp, len, (uint32_t)ranks[v], rho, {p[0], p[1], len == 3 ? p[2] : (uint8_t)0}, true};
        cand_of[ranks[v]] = (int32_t)cands.size();
        cands.push_back(c);
    }
    if (cands.empty()) return;
    // 3.: a pair with the character as one of its parts must rank above every merge inside the character
    for (const auto& pr : pairs) {
        const uint32_t l = (uint32_t)(pr.first >> ID_BITS), r = (uint32_t)(pr.first & ((1u << ID_BITS) - 1));
        if (l <= (uint32_t)H.max_id && cand_of[l] >= 0 && pr.second <= cands[cand_of[l]].rho) cands[cand_of[l]].ok = false;
        if (r <= (uint32_t)H.max_id && cand_of[r] >= 0 && pr.second <= cands[cand_of[r]].rho) cands[cand_of[r]].ok = false;
    }
    // 2.: every token that ends in a proper prefix of some character (a lead byte and up to two continuation bytes short of the
    // character's length) with at least one byte in front of it, and every token that starts with continuation bytes followed by at
    // least one more byte (proper suffixes of characters: every prefix of the run of continuation bytes counts)
    std::map<uint32_t, Bits256> tails, heads;
    for (int64_t v = 0; v < n_vocab; ++v) {
        const uint8_t* p = token_bytes + token_offsets[v];
        const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        if (len < 2) continue;
        {   // tail: ... x | lead cont*   (shorter than the lead's character)
            uint32_t i = len - 1, nc = 0;
            while (i > 0 && (p[i] & 0xC0) == 0x80 && nc < 3) { --i; ++nc; }
            const uint32_t lead = p[i];
            const uint32_t need = lead >= 0xC2 && lead < 0xE0 ? 2u : lead >= 0xE0 && lead < 0xF0 ? 3u : lead >= 0xF0 && lead < 0xF8 ? 4u : 0u;
            if (need && nc + 1 < need && i > 0 && need <= 3) tails[key_of(p + i, nc + 1)].set(p[i - 1]);
        }
        {   // heads: cont{j} y ...
            uint32_t j = 0;
            while (j < len && j < 3 && (p[j] & 0xC0) == 0x80) ++j;
            for (uint32_t jj = 1; jj <= j && jj < len; ++jj)
                if (jj <= 2) heads[key_of(p, jj)].set(p[jj]);
        }
    }
    std::map<Bits256, uint32_t> pm_rows, nm_rows;
    auto row_of = [](std::map<Bits256, uint32_t>& rows, std::vector<uint32_t>& store, const Bits256& b) -> int {
        auto it = rows.find(b);
        if (it != rows.end()) return (int)it->second;
        if (rows.size() >= 256) return -1;
        const uint32_t r = (uint32_t)rows.size();
        rows.emplace(b, r);
        store.insert(store.end(), b.w, b.w + 8);
        return (int)r;
    };
    H.cseed.assign(65536, 0);
    for (const Cand& c : cands) {
        if (!c.ok) continue;
        Bits256 pm, nm;
        for (uint32_t i = 1; i < c.k; ++i) {
            auto t = tails.find(key_of(c.b, i));
            if (t != tails.end()) pm.join(t->second);
            auto h = heads.find(key_of(c.b + i, c.k - i));
            if (h != heads.end()) nm.join(h->second);
        }
        const int pr = row_of(pm_rows, H.cseed_pm, pm), nr = row_of(nm_rows, H.cseed_nm, nm);
        if (pr < 0 || nr < 0) continue;  // (more distinct sets than an entry can name: the character is entered byte by byte)
        H.cseed[c.cp] = CS_VALID | ((uint64_t)nr << 29) | ((uint64_t)pr << 21) | c.id;
        ++H.n_char_seeds;
    }
    if (getenv("TD_DEBUG_TABLES"))
        fprintf(stderr, "[tokendagger] character seeds: %u of %zu one-character tokens of 2..3 bytes, %zu + %zu sets of neighbour bytes\n",
                H.n_char_seeds, cands.size(), pm_rows.size(), nm_rows.size());
    if (!H.n_char_seeds) { H.cseed.clear(); H.cseed_pm.clear(); H.cseed_nm.clear(); }
}

int build_tables(const char* pattern, int64_t n_vocab, const uint8_t* token_bytes, const int64_t* token_offsets,
                 const int32_t* ranks, int64_t n_special, const uint8_t* special_bytes,
                 const int64_t* special_offsets, const int32_t* special_ranks, HostTables& H, std::string& err) {
    H = HostTables();
    H.pattern = pattern ? pattern : "";
    H.pattern_kind = classify_pattern(H.pattern);
    if (H.pattern_kind == PATTERN_UNSUPPORTED) {
        // not a member of the family with a scanner of its own: the generic engine, if the pattern is within its subset
        static_assert(std::is_trivially_copyable<RxProgram>::value, "the compiled pattern is uploaded as bytes");
        H.rx_program.resize(sizeof(RxProgram));
        std::string why;
        RxProgram* P = reinterpret_cast<RxProgram*>(H.rx_program.data());
        if (rx_compile(H.pattern, *P, why)) {
            H.pattern_kind = PATTERN_GENERIC;
            for (uint32_t k = 0; k < P->n_nodes; ++k)
                if (P->nodes[k].kind == RX_BOS || P->nodes[k].kind == RX_WORDB || P->nodes[k].kind == RX_NWORDB || P->nodes[k].kind == RX_NLOOKB || P->nodes[k].kind == RX_PLOOKB) H.rx_left_context = true;
        } else {
            H.rx_program.clear();
            err = "split pattern is not supported by the device pre-tokenizer: " + why +
                  " (supported: the o200k/Llama-4, Mistral tekken, cl100k_base/Llama-3, Qwen2 and GPT-2 patterns, and patterns within the "
                  "subset td_regex.h lists); there is no CPU regex fallback";
            return TD_E_PATTERN;
        }
    }
    if (n_vocab <= 0) { err = "empty vocabulary"; return TD_E_VOCAB; }

    // Class tables: the probed Unicode table; cl100k reads marks as punctuation and gives '/' no trailer role, which is
    // a remap of two class ids (the scanners never see the difference)
    if (H.pattern_kind == PATTERN_CL100K || H.pattern_kind == PATTERN_CL100K_EOS || H.pattern_kind == PATTERN_QWEN2 || H.pattern_kind == PATTERN_GPT2) {
        H.ucls2_remap.assign(td_ucls_stage2, td_ucls_stage2 + sizeof td_ucls_stage2);
        for (auto& c : H.ucls2_remap)
            if (c == C_MK || c == C_SLASH) c = C_OTHER;
    }
    const uint8_t* stage2 = H.ucls2_remap.empty() ? td_ucls_stage2 : H.ucls2_remap.data();
    H.ascii_cls.resize(128);
    for (uint32_t c = 0; c < 128; ++c) H.ascii_cls[c] = stage2[(uint32_t)td_ucls_stage1[0] * 256u + c];

    int32_t max_rank = -1;
    uint32_t max_len = 0;
    for (int64_t v = 0; v < n_vocab; ++v) {
        const int64_t len = token_offsets[v + 1] - token_offsets[v];
        if (len <= 0) { err = "vocabulary contains an empty token"; return TD_E_VOCAB; }
        if (ranks[v] < 0) { err = "negative rank"; return TD_E_VOCAB; }
        max_rank = std::max(max_rank, ranks[v]);
        max_len = std::max<uint32_t>(max_len, (uint32_t)len);
    }
    int32_t max_id = max_rank;
    for (int64_t s = 0; s < n_special; ++s) max_id = std::max(max_id, special_ranks[s]);
    H.max_rank = max_rank;
    H.pseudo_base = max_id + 1;
    if ((int64_t)H.pseudo_base + 256 >= (1ll << ID_BITS) - 1) {  // (the id 2^21 - 1 is the empty pair slot's)
        err = "token ids must be < 2^21 - 256";
        return TD_E_VOCAB;
    }
    H.max_id = max_id;
    H.max_token_len = max_len;

    // rank -> bytes store (regular tokens first; specials fill ids the regular vocab leaves free)
    std::vector<uint32_t> len_of((size_t)max_id + 1, 0);
    std::vector<int64_t> src_of((size_t)max_id + 1, -1);  // >=0: regular vocab index; <=-2: -(special index)-2
    for (int64_t v = 0; v < n_vocab; ++v) {
        if (len_of[ranks[v]] != 0) { err = "duplicate rank " + std::to_string(ranks[v]); return TD_E_VOCAB; }
        len_of[ranks[v]] = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        src_of[ranks[v]] = v;
    }
    for (int64_t s = 0; s < n_special; ++s) {
        const int32_t id = special_ranks[s];
        if (id < 0) { err = "negative special id"; return TD_E_VOCAB; }
        H.special_strs.emplace_back((const char*)special_bytes + special_offsets[s],
                                    (size_t)(special_offsets[s + 1] - special_offsets[s]));
        H.special_ids.push_back(id);
        if (len_of[id] == 0) {
            len_of[id] = (uint32_t)(special_offsets[s + 1] - special_offsets[s]);
            src_of[id] = -s - 2;
        }
    }
    H.tok_off.assign((size_t)max_id + 2, 0);
    for (int32_t id = 0; id <= max_id; ++id) H.tok_off[id + 1] = H.tok_off[id] + len_of[id];
    H.tok_bytes.assign((size_t)H.tok_off[max_id + 1] + 32, 0);  // (the device compares 16 bytes at a time from an aligned dword: reads up to 23 bytes behind a token)
    for (int32_t id = 0; id <= max_id; ++id) {
        if (src_of[id] >= 0)
            memcpy(&H.tok_bytes[H.tok_off[id]], token_bytes + token_offsets[src_of[id]], len_of[id]);
        else if (src_of[id] <= -2)
            memcpy(&H.tok_bytes[H.tok_off[id]], special_bytes + special_offsets[-src_of[id] - 2], len_of[id]);
    }

    // piece table: bytes -> rank
    const uint32_t pcap = pow2_at_least((uint64_t)n_vocab * 2);
    H.piece_mask = pcap - 1;
    H.piece_slots.assign(pcap, PieceSlot{0, 0, 0});
    H.byte_id.assign(256, 0);
    for (int b = 0; b < 256; ++b) H.byte_id[b] = H.pseudo_base + b;
    H.byte_pair.assign(65536, NO_RANK);
    for (int64_t v = 0; v < n_vocab; ++v) {
        const uint8_t* p = token_bytes + token_offsets[v];
        const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        const uint64_t key = piece_key_host(p, len);
        uint32_t h = hash_piece(key, len) & H.piece_mask;
        for (;;) {
            PieceSlot& s = H.piece_slots[h];
            if (s.len == 0) { s.key = key; s.rank = (uint32_t)ranks[v]; s.len = len; break; }
            if (s.key == key && s.len == len &&
                (len <= 8 || memcmp(&H.tok_bytes[H.tok_off[s.rank]], p, len) == 0)) {
                err = "duplicate token bytes in vocabulary";
                return TD_E_VOCAB;
            }
            h = (h + 1) & H.piece_mask;
        }
        if (len == 1) H.byte_id[p[0]] = ranks[v];
        if (len == 2) H.byte_pair[((uint32_t)p[0] << 8) | p[1]] = ranks[v];
    }

    H.byte_pair_id.resize(65536);
    for (uint32_t q = 0; q < 65536; ++q) H.byte_pair_id[q] = (uint64_t)(uint32_t)H.byte_pair[q] | ((uint64_t)(uint32_t)H.byte_id[q >> 8] << 32);

    // exact-key table for the tokens of 1..12 bytes (the hot probe loop of td_probe_tiles: one 16-byte load per piece, first
    // slot only).  Inserted in RANK order: a key leaves its home slot only when a lower-rank (as a rule: more frequent) key
    // took it, so the pieces that matter are answered by the first slot.
    {
        std::vector<int64_t> order;
        for (int64_t v = 0; v < n_vocab; ++v)
            if (token_offsets[v + 1] - token_offsets[v] <= (int64_t)P12_MAXLEN) order.push_back(v);
        std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return ranks[x] < ranks[y]; });
        const uint32_t cap12 = pow2_at_least((uint64_t)order.size() * 2 + 2);
        H.piece12_mask = cap12 - 1;
        H.piece12_slots.assign(cap12, Piece12Slot{0, 0, 0, 0});
        for (int64_t v : order) {
            const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
            const uint8_t* p = token_bytes + token_offsets[v];
            uint32_t k[3] = {0, 0, 0};
            for (uint32_t i = 0; i < len; ++i) k[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
            uint32_t h = hash_piece12(k[0], k[1], k[2], len) & H.piece12_mask;
            while (H.piece12_slots[h].meta != 0) h = (h + 1) & H.piece12_mask;
            H.piece12_slots[h] = Piece12Slot{k[0], k[1], k[2], p12_meta((uint32_t)ranks[v], len)};
        }
    }

    // pair table: every split of every token whose halves are both "parts" the merge loop can
    // hold: a token, or a single byte (even one that is not a token -> pseudo id), because the
    // reference keys get_rank by the bytes of two adjacent parts (tiktoken.cpp:282-296).
    Tables T = H.view();
    auto id_of = [&](const uint8_t* p, uint32_t len) -> int32_t {
        if (len == 1) return H.byte_id[p[0]];
        return piece_lookup(T, piece_key_host(p, len), len, [p](uint32_t i) { return (uint32_t)p[i]; });
    };
    std::vector<std::pair<uint64_t, uint32_t>> pairs;
    pairs.reserve((size_t)n_vocab * 3);
    for (int64_t v = 0; v < n_vocab; ++v) {
        const uint8_t* p = token_bytes + token_offsets[v];
        const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        for (uint32_t k = 1; k < len; ++k) {
            const int32_t l = id_of(p, k);
            if (l == NO_RANK) continue;
            const int32_t r = id_of(p + k, len - k);
            if (r == NO_RANK) continue;
            pairs.emplace_back(((uint64_t)(uint32_t)l << ID_BITS) | (uint32_t)r, (uint32_t)ranks[v]);
        }
    }
    H.n_pairs = pairs.size();
    // cuckoo insertion (2 hash functions, random-walk eviction); grow the table if a walk does not terminate
    for (uint32_t qcap = pow2_at_least((uint64_t)pairs.size() * 2 + 2);; qcap <<= 1) {
        H.pair_mask = qcap - 1;
        H.pair_slots.assign(qcap, PAIR_EMPTY);
        bool ok = true;
        uint32_t rng = 0x9E3779B9u;
        for (auto& pr : pairs) {
            uint64_t cur = (pr.first << ID_BITS) | pr.second;
            bool placed = false;
            uint32_t h = 0;
            for (int kick = 0; kick < 1000; ++kick) {
                const uint32_t l = (uint32_t)(cur >> (2 * ID_BITS)), r = (uint32_t)((cur >> ID_BITS) & ((1u << ID_BITS) - 1));
                const uint32_t h1 = hash_pair(l, r) & H.pair_mask, h2 = hash_pair2(l, r) & H.pair_mask;
                if (H.pair_slots[h1] == PAIR_EMPTY) { H.pair_slots[h1] = cur; placed = true; break; }
                if (H.pair_slots[h2] == PAIR_EMPTY) { H.pair_slots[h2] = cur; placed = true; break; }
                rng = rng * 1664525u + 1013904223u;
                h = (kick == 0) ? ((rng >> 16) & 1 ? h1 : h2) : (h == h1 ? h2 : h1);  // evict from the other seat
                std::swap(cur, H.pair_slots[h]);
            }
            if (!placed) { ok = false; break; }
        }
        if (ok) break;
        if (qcap >= (1u << 28)) { err = "pair table construction failed"; return TD_E_VOCAB; }
    }
    // PAIR_FINAL (td_common.h): a slot nobody was pushed out of says so
    {
        std::vector<uint8_t> pushed(H.pair_slots.size(), 0);
        for (size_t sl = 0; sl < H.pair_slots.size(); ++sl) {
            const uint64_t e = H.pair_slots[sl];
            if (e == PAIR_EMPTY) continue;
            const uint32_t l = (uint32_t)(e >> (2 * ID_BITS)), r = (uint32_t)((e >> ID_BITS) & ((1u << ID_BITS) - 1));
            const uint32_t h1 = hash_pair(l, r) & H.pair_mask;
            if (h1 != sl) { pushed[h1] = 1; ++H.n_pairs_second_seat; }
        }
        size_t open_slots = 0;
        for (size_t sl = 0; sl < H.pair_slots.size(); ++sl) {
            if (pushed[sl]) { H.pair_slots[sl] &= ~PAIR_FINAL; ++open_slots; }
            else H.pair_slots[sl] |= PAIR_FINAL;
        }
        if (getenv("TD_DEBUG_TABLES"))
            fprintf(stderr, "[tokendagger] pair table: %llu pairs in %zu slots, %llu in their second seat, %zu first seats not final\n",
                    (unsigned long long)H.n_pairs, H.pair_slots.size(), (unsigned long long)H.n_pairs_second_seat, open_slots);
    }

    build_char_seeds(H, n_vocab, token_bytes, token_offsets, ranks, pairs);

    // Is the whole-piece fast path redundant (encode == encode_ordinary on every input)?
    T = H.view();
    H.merge_closed = true;
    std::vector<int32_t> tmp;
    for (int64_t v = 0; v < n_vocab && H.merge_closed; ++v) {
        const uint32_t len = (uint32_t)(token_offsets[v + 1] - token_offsets[v]);
        if (len < 2) continue;
        tmp.clear();
        const int rc = merge_piece_host(T, token_bytes + token_offsets[v], len, tmp);
        if (rc != TD_OK || tmp.size() != 1 || tmp[0] != ranks[v]) H.merge_closed = false;
    }
    return TD_OK;
}

}  // namespace td
