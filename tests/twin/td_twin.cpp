// TEST INFRASTRUCTURE ONLY: CPU "twin" of the device algorithm.
//
// Compiles the SAME header code the kernels use (tokendagger_amd/csrc/td_common.h: classify_at,
// is_sync, scan_piece, split_unresolved_heads, scan_chain, piece/pair table probes) plus the host table builder with the
// host compiler and runs it lane by lane / tile by tile on the CPU, so that the not-gpu test-suite
// can check the product's host logic (tables, scanner, sync-point speculation, tile ownership)
// against the oracle without a GPU.  It is NOT reachable from the product: nothing under
// tokendagger_amd/ links or loads it, and libtokendagger_hip.so has no CPU tokenization path.
#include <string.h>

#include <string>
#include <vector>

#include "../../tokendagger_amd/csrc/td_common.h"
#include "../../tokendagger_amd/csrc/td_regex.h"
#include "../../tokendagger_amd/csrc/td_tables.h"

using namespace td;

namespace {

struct Twin {
    HostTables H;
    std::string err;
};

struct Src {  // byte + document-start source over the whole text
    const uint8_t* text;
    const uint8_t* docs;  // 1 byte per position
    int64_t lo, hi;
    uint32_t byte(int64_t i) const { return text[i]; }
    bool doc(int64_t i) const { return docs[i] != 0; }
};

// accessor over a precomputed class+flag array of the whole text (the twin's "HBM slow path")
struct GAcc {
    using pos_t = int64_t;
    const uint8_t* cls;
    const uint8_t* text;
    int64_t n, lim;
    uint32_t pv;  // PV_* pattern variant
    uint32_t cf(int64_t i) const { return i >= n ? (uint32_t)F_DOC : cls[i]; }
    uint32_t byte(int64_t i) const { return i < n ? text[i] : 0u; }
    int64_t scan(int64_t pos) const { return scan_piece(*this, pos, pv); }
};

struct WAcc {  // one tile window
    using pos_t = int;
    uint8_t* cls;
    const uint8_t* txt;
    int lim;
    int ext_start;
    long long ext_end;
    uint32_t cf(int i) const { return cls[i]; }
    uint32_t byte(int i) const { return txt[i]; }
    void mark(int i) { cls[i] |= F_START; }
    void set_ext(int i, int64_t ge) { ext_start = i; ext_end = ge; }
    int n_far = 0, last_far = -1;
    void note_far(int i) { ++n_far; last_far = i; }
};

void classify_all(const Tables& T, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs,
                  std::vector<uint8_t>& cls) {
    std::vector<uint8_t> doc((size_t)n + 1, 0);
    for (int64_t d = 0; d < n_docs; ++d)
        if (offs[d] < n) doc[(size_t)offs[d]] = 1;
    Src s{text, doc.data(), 0, n};
    cls.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t v = classify_at(T, s, i);
        if (doc[(size_t)i]) v |= F_DOC;
        cls[(size_t)i] = (uint8_t)v;
    }
}

}  // namespace

static int64_t g_fast_total = 0, g_fast_hit = 0;
static int64_t g_mg_len[17] = {0};
static int64_t g_mg_pieces[5] = {0, 0, 0, 0, 0}, g_mg_rounds[5] = {0, 0, 0, 0, 0};  // merged pieces / merge rounds by length class

extern "C" {

void twin_merge_lens(int64_t* out17) { for (int i = 0; i < 17; ++i) { out17[i] = g_mg_len[i]; g_mg_len[i] = 0; } }
void twin_merge_stats(int64_t* out10) {
    for (int c = 0; c < 5; ++c) { out10[c] = g_mg_pieces[c]; out10[5 + c] = g_mg_rounds[c]; g_mg_pieces[c] = g_mg_rounds[c] = 0; }
}

void twin_fast_stats(int64_t* total, int64_t* hit) { *total = g_fast_total; *hit = g_fast_hit; g_fast_total = g_fast_hit = 0; }

void* twin_create(const char* pat, int64_t n_vocab, const uint8_t* bytes, const int64_t* offs, const int32_t* ranks,
                  int64_t n_special, const uint8_t* sbytes, const int64_t* soffs, const int32_t* sranks, int* rc_out) {
    Twin* t = new Twin;
    int rc = build_tables(pat, n_vocab, bytes, offs, ranks, n_special, sbytes, soffs, sranks, t->H, t->err);
    if (rc == TD_OK && t->H.pattern_kind == PATTERN_GENERIC) {  // (the twin models the family's kernels; generic patterns: twin_rx_split)
        rc = TD_E_PATTERN;
        t->err = "the CPU twin models the pattern family only";
    }
    if (rc_out) *rc_out = rc;
    if (rc != TD_OK) {
        static thread_local std::string keep;
        keep = t->err;
        delete t;
        return nullptr;
    }
    return t;
}
void twin_destroy(void* h) { delete (Twin*)h; }

int64_t twin_info(void* h, int what) {
    Twin* t = (Twin*)h;
    switch (what) {
        case 1: return (int64_t)t->H.n_pairs;
        case 2: return t->H.merge_closed;
        case 3: return t->H.max_id;
        case 4: return t->H.piece_mask + 1;
        case 5: return t->H.pair_mask + 1;
        case 6: return t->H.n_char_seeds;
    }
    return -1;
}

// Character seeds (td_common.h): the pieces pieces[offs[i] .. offs[i+1]) merged from the SEEDED parts — cseed_part_at at every byte, as
// td_long_pieces' set-up asks it, then the reference's loop (lowest rank, leftmost; tiktoken.cpp:322-343) over those parts through the
// pair table.  ids_out / id_offs as usual; stats2 = [characters entered whole, parts at the start summed over the pieces].
// Also checks the set-up's own bookkeeping: the part in front of every part start, found the way the kernel finds it.  -> ids or -1.
int64_t twin_seeded_merge(void* h, const uint8_t* pieces, const int64_t* offs, int64_t n_pieces, int32_t* ids_out, int64_t cap,
                          int64_t* id_offs, int64_t* stats2) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    int64_t n_out = 0, seeded = 0, parts0 = 0;
    std::vector<uint32_t> ids, start;
    for (int64_t i = 0; i < n_pieces; ++i) {
        const uint8_t* p = pieces + offs[i];
        const uint32_t len = (uint32_t)(offs[i + 1] - offs[i]);
        auto get = [p](uint32_t k) { return (uint32_t)p[k]; };
        ids.clear(); start.clear();
        uint32_t expect = 0;  // next part start when walking parts left to right
        for (uint32_t q = 0; q < len; ++q) {
            const SeedPart sp = cseed_part_at(T, get, len, q);
            if (sp.kind == 0u) {
                if (q >= expect || sp.back == 0u || start.empty() || start.back() != q - sp.back) return -2;  // the inside of a character that was not seen to start
                continue;
            }
            if (q != expect) return -3;  // a part starts inside another one
            if (q) {  // the kernel's way to the part in front
                const SeedPart pp = cseed_part_at(T, get, len, q - 1u);
                if (start.empty() || q - 1u - pp.back != start.back()) return -4;
            }
            start.push_back(q);
            ids.push_back(sp.kind == 2u ? sp.id : (uint32_t)T.byte_id[p[q]]);
            seeded += sp.kind == 2u;
            expect = q + sp.k;
        }
        if (expect != len) return -5;
        parts0 += (int64_t)ids.size();
        for (;;) {
            int32_t best = NO_RANK; size_t bi = 0;
            for (size_t k = 0; k + 1 < ids.size(); ++k) {
                const int32_t r = pair_lookup(T, ids[k], ids[k + 1]);
                if (r < best) { best = r; bi = k; }
            }
            if (best == NO_RANK) break;
            ids[bi] = (uint32_t)best;
            ids.erase(ids.begin() + (long)bi + 1);
        }
        id_offs[i] = n_out;
        for (uint32_t v : ids) {
            if (n_out >= cap) return -1;
            ids_out[n_out++] = (int32_t)v;
        }
    }
    id_offs[n_pieces] = n_out;
    stats2[0] = seeded; stats2[1] = parts0;
    return n_out;
}

// The pair table as the DEVICE's merge rounds probe it (mg_round_t, td_common.h): the first seat; the second one only where the first
// neither holds the pair nor is marked PAIR_FINAL.  Checked against pair_lookup (both seats) on every pair of the table, on every
// pair with its halves swapped and on n_random random id pairs.  -> mismatches; stats: [pairs, second-seat probes for them,
// other pairs tried, second-seat probes for those, byte_pair_id entries that differ from byte_pair / byte_id]
int64_t twin_pair_probe_check(void* h, int64_t n_random, uint64_t seed, int64_t* stats5) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    int64_t bad = 0, st[5] = {0, 0, 0, 0, 0};
    auto device_order = [&](uint32_t l, uint32_t r, int64_t& second) -> int32_t {
        const uint64_t e1 = T.pair_slots[hash_pair(l, r) & T.pair_mask];
        uint64_t e2 = PAIR_EMPTY;
        if (!(e1 & PAIR_FINAL) && !pair_slot_is(e1, l, r)) { e2 = T.pair_slots[hash_pair2(l, r) & T.pair_mask]; ++second; }
        return pair_match(e1, e2, l, r);
    };
    for (size_t sl = 0; sl <= T.pair_mask; ++sl) {
        const uint64_t e = T.pair_slots[sl];
        if ((e | PAIR_FINAL) == PAIR_EMPTY) continue;
        const uint32_t l = (uint32_t)((e >> (2 * ID_BITS)) & ((1u << ID_BITS) - 1)), r = (uint32_t)((e >> ID_BITS) & ((1u << ID_BITS) - 1));
        ++st[0];
        const int32_t want = pair_lookup(T, l, r);
        if (want == NO_RANK || want != (int32_t)(e & ((1u << ID_BITS) - 1)) || device_order(l, r, st[1]) != want) ++bad;
        ++st[2];
        if (device_order(r, l, st[3]) != pair_lookup(T, r, l)) ++bad;
    }
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (int64_t i = 0; i < n_random; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint32_t l = (uint32_t)(x % (uint64_t)(t->H.max_id + 1)), r = (uint32_t)((x >> 32) % (uint64_t)(t->H.max_id + 1));
        ++st[2];
        if (device_order(l, r, st[3]) != pair_lookup(T, l, r)) ++bad;
    }
    for (uint32_t q = 0; q < 65536; ++q)
        if ((uint32_t)T.byte_pair_id[q] != (uint32_t)T.byte_pair[q] || (uint32_t)(T.byte_pair_id[q] >> 32) != (uint32_t)T.byte_id[q >> 8]) { ++st[4]; ++bad; }
    for (int i = 0; i < 5; ++i) stats5[i] = st[i];
    return bad;
}

// per-byte class + F_CONT + F_DOC exactly as phase 1 of the kernel defines them
void twin_classify(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, uint8_t* out) {
    Twin* t = (Twin*)h;
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    if (n) memcpy(out, cls.data(), (size_t)n);
}

// piece starts by serial scanning from every document start (flags[i] = 1 where a piece starts)
int twin_split_serial(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, uint8_t* flags) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    memset(flags, 0, (size_t)n);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    int64_t p = 0;
    while (p < n) {
        flags[p] = 1;
        int64_t e = scan_piece(g, p, T.pat_flags);
        if (e <= p) return -1;
        p = e;
    }
    return 0;
}

// piece starts found the way td_split_tiles finds them: tile windows, whole-word rules, one scan_chain per open head.
// stats[0] = pieces that left their window (ext), stats[1] = tiles without a sync point in the left halo,
// stats[3] = synchronisation points, stats[4] = those the whole-word rules left open.
int twin_split_tiled(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, uint8_t* flags,
                     int64_t* ext_ends /* [n] or NULL: global end for ext pieces, else 0 */, int64_t* stats) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    memset(flags, 0, (size_t)n);
    if (ext_ends) memset(ext_ends, 0, sizeof(int64_t) * (size_t)n);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    const int64_t n_tiles = (n + KS_TILE - 1) / KS_TILE;
    std::vector<uint8_t> wcls(K_WIN), wtxt(K_WIN);
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
        const int64_t tile_g0 = tile * KS_TILE, wg0 = tile_g0 - K_HL;
        const int tile_hi = K_HL + (int)((n - tile_g0 < KS_TILE) ? (n - tile_g0) : KS_TILE);
        for (int i = 0; i < K_WIN; ++i) {
            const int64_t gi = wg0 + i;
            if (gi < 0) { wcls[i] = C_OTHER; wtxt[i] = 0; }
            else if (gi >= n) { wcls[i] = F_DOC; wtxt[i] = 0; }
            else { wcls[i] = cls[(size_t)gi]; wtxt[i] = text[gi]; }
        }
        // NOTE: the kernel classifies the window from LDS; bytes within 3 of the window edges can
        // differ from the whole-text classification, which is why the scanner is limited to < K_LIM
        // and the back-search to >= 4.
        WAcc w{wcls.data(), wtxt.data(), K_LIM, -1, 0};
        // the kernel's phase 2: (a) whole-word rules per 32-byte stride on 64-bit mask windows, START = the sync points;
        // (b) chains from the unresolved heads, the last sync point before the tile and the tile's last head.  Chains only
        // communicate through F_START marks, which no chain reads for its decisions, so the order does not matter.
        std::vector<int> heads;
        int last_head = -1;
        for (int b0 = K_HL; b0 < tile_hi; b0 += 32) {
            BitWin bw;
            for (int k = 0; k < MK_COUNT; ++k) bw.m[k] = 0;
            for (int i = 0; i < 64; ++i) {
                const int q = b0 + i;
                if (q >= K_WIN) break;  // (zero word behind the window)
                const uint32_t bits = mask_bits_of(w.cf(q - 1), w.cf(q), T.pat_flags);
                for (int k = 0; k < MK_COUNT; ++k) bw.m[k] |= (uint64_t)((bits >> k) & 1u) << i;
            }
            uint32_t sy = (uint32_t)bw.m[MK_SYNC];
            if (tile_hi - b0 < 32) sy &= (1u << (tile_hi - b0)) - 1u;
            uint64_t extra = 0;
            const uint32_t un = (uint32_t)split_unresolved_heads(bw, T.pat_flags, &extra) & sy;
            extra &= (uint64_t)sy << 3;  // (as the kernel: behind this stride's heads, inside the tile)
            const int room = tile_hi - b0;
            extra &= room >= 64 ? ~0ull : room <= 0 ? 0ull : ((1ull << room) - 1ull);
            for (int i = 0; i < 32; ++i) {
                if ((sy >> i) & 1u) { w.mark(b0 + i); last_head = b0 + i; }
                if ((un >> i) & 1u) heads.push_back(b0 + i);
            }
            for (int i = 0; i < 64; ++i)
                if ((extra >> i) & 1ull) w.mark(b0 + i);
            if (stats) { stats[3] += __builtin_popcount(sy); stats[4] += __builtin_popcount(un); }
        }
        if (!is_sync(w.cf(K_HL - 1), w.cf(K_HL), T.pat_flags)) {
            int s = -1;
            for (int i = K_HL - 1; i >= 4; --i)
                if (is_sync(w.cf(i - 1), w.cf(i), T.pat_flags)) { s = i; break; }
            if (s >= 0) heads.push_back(s);
            else {  // flagged tile (inside a piece longer than the halo): what the td_split_far_* kernels do
                if (stats) stats[1]++;
                int64_t gs = 0;
                for (int64_t gi = wg0 + 3; gi > 0; --gi)
                    if (is_sync(g.cf(gi - 1), g.cf(gi), T.pat_flags)) { gs = gi; break; }
                int64_t p = gs;
                while (p < tile_g0) p = g.scan(p);
                // piece by piece up to the first synchronisation point of the tile (the fast kernel's heads take over)
                if (p - wg0 < (int64_t)tile_hi && !is_sync(g.cf(p - 1), g.cf(p), T.pat_flags)) {
                    w.mark((int)(p - wg0));
                    heads.push_back((int)(p - wg0));
                }
            }
        }
        if (last_head >= 0) heads.push_back(last_head);
        for (int hd : heads) scan_chain(w, g, hd, tile_hi, wg0, T.pat_flags);
        for (int i = K_HL; i < tile_hi; ++i)
            if (wcls[i] & F_START) flags[wg0 + i] = 1;
        if (stats && w.n_far) { stats[2] += w.n_far; stats[5] = wg0 + w.last_far; }
        if (w.ext_start >= 0) {
            if (stats) stats[0]++;
            if (ext_ends) ext_ends[wg0 + w.ext_start] = w.ext_end;
        }
        // piece lengths as phase 3 derives them: distance to the next START mark inside the window
        for (int i = K_HL; i < tile_hi; ++i) {
            if (!(wcls[i] & F_START) || i == w.ext_start) continue;
            int j = i + 1;
            while (j < K_WIN && !(wcls[j] & F_START)) ++j;
            if (j >= K_WIN) return -2;  // a non-ext piece must be delimited inside the window
        }
    }
    return 0;
}

// number of positions where is_sync() claims a piece start that the serial scan does not produce
int64_t twin_sync_violations(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs,
                             int64_t* n_sync) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    std::vector<uint8_t> flags((size_t)n + 1, 0);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    for (int64_t p = 0; p < n;) { flags[(size_t)p] = 1; p = scan_piece(g, p, T.pat_flags); }
    int64_t bad = 0, ns = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t vp = i > 0 ? cls[(size_t)i - 1] : 0u;
        if (is_sync(vp, cls[(size_t)i], T.pat_flags)) { ++ns; if (!flags[(size_t)i]) ++bad; }
    }
    if (n_sync) *n_sync = ns;
    return bad;
}

// whole pipeline on the host tables: tiled split -> piece table -> pair-table merge.
// Returns number of tokens, or -(TD_E_*) on error.
int64_t twin_encode(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, int mode,
                    int32_t* out, int64_t cap, int64_t* out_offs) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> flags((size_t)n + 1, 0);
    if (twin_split_tiled(h, text, n, offs, n_docs, flags.data(), nullptr, nullptr) != 0) return -TD_E_INVALID;
    const bool fast = (mode == TD_MODE_ENCODE) || t->H.merge_closed;
    std::vector<int64_t> tok_at((size_t)n + 1, 0);  // tokens emitted before byte i
    int64_t k = 0;
    std::vector<int32_t> tmp;
    for (int64_t p = 0; p < n;) {
        int64_t e = p + 1;
        while (e < n && !flags[(size_t)e]) ++e;
        tok_at[(size_t)p] = k;
        const uint32_t len = (uint32_t)(e - p);
        const uint8_t* pb = text + p;
        tmp.clear();
        if (len == 1) {
            if (T.byte_id[pb[0]] >= T.pseudo_base) return -TD_E_UNKNOWN_BYTE;
            tmp.push_back(T.byte_id[pb[0]]);
        } else {
            int32_t r = NO_RANK;
            if (fast) r = piece_lookup(T, piece_key_host(pb, len), len, [pb](uint32_t i) { return (uint32_t)pb[i]; });
            if (len <= P12_MAXLEN) {
                // what the hot probe of td_probe_tiles does: the FIRST slot of the exact-key table decides a hit (same key) or
                // a miss (empty slot); any other slot sends the piece to the generic lookup.  Both must agree with it.
                uint32_t k[3] = {0, 0, 0};
                for (uint32_t i = 0; i < len; ++i) k[i >> 2] |= (uint32_t)pb[i] << (8 * (i & 3));
                const Piece12Slot& sl = T.piece12_slots[hash_piece12(k[0], k[1], k[2], len) & T.piece12_mask];
                const int32_t want = piece_lookup(T, piece_key_host(pb, len), len, [pb](uint32_t i) { return (uint32_t)pb[i]; });
                if (sl.meta == 0) { if (want != NO_RANK) return -100; }
                else if (sl.k0 == k[0] && sl.k1 == k[1] && sl.k2 == k[2] && (sl.meta >> 24) == (0x80u | len)) {
                    if ((int32_t)(sl.meta & 0x1FFFFFu) != want) return -101;
                }
            }
            if (r != NO_RANK) tmp.push_back(r);
            else if (len <= (uint32_t)K_MAXSHORT) {
                // what one lane of td_merge_tiles runs (mg_put / mg_pad / mg_round, td_common.h), on host arrays; the
                // piece sits at a varying unit so that the slot swizzle is exercised
                alignas(16) static thread_local uint32_t keys[256 * MG_UNIT + 64], ids[256 * MG_UNIT + 64];
                MergeState st;
                st.t = (uint32_t)(p % 250);
                st.len = len;
                st.alive = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
                for (uint32_t j = 0; j < len; ++j) mg_put(T, T.byte_id, keys, ids, st, j, pb[j], j + 1 < len ? pb[j + 1] : 0u);
                mg_pad(keys, st);
                uint32_t rounds = 0;
                if (len <= 32) { while (mg_round_t<uint32_t>(T, keys, ids, st)) ++rounds; }  // (as td_merge_pieces picks the mask width)
                else { while (mg_round_t<uint64_t>(T, keys, ids, st)) ++rounds; }
                { const int c = len <= 8 ? 0 : len <= 16 ? 1 : len <= 32 ? 2 : len <= 48 ? 3 : 4; g_mg_pieces[c]++; g_mg_rounds[c] += rounds; g_mg_len[len < 16 ? len : 16]++; }
                for (uint64_t al = st.alive; al; al &= al - 1) {
                    const uint32_t v = ids[mg_slot(st.t, (uint32_t)td_ctz64(al))];
                    if ((int32_t)v >= T.pseudo_base) return -TD_E_UNKNOWN_BYTE;
                    tmp.push_back((int32_t)v);
                }
            } else if (merge_piece_host(T, pb, len, tmp) != TD_OK) return -TD_E_UNKNOWN_BYTE;
        }
        for (int32_t v : tmp) {
            if (k >= cap) return -TD_E_CAPACITY;
            out[k++] = v;
        }
        p = e;
    }
    for (int64_t d = 0; d <= n_docs; ++d) out_offs[d] = (offs[d] >= n) ? k : tok_at[(size_t)offs[d]];
    return k;
}

// bit-parallel scanner vs byte scanner on every true piece start, with the 64-byte mask window placed
// at several offsets before the piece.  Returns mismatches; *n_unres counts "-1 (window too short)".
int64_t twin_bits_check(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, int64_t* n_unres,
                        int64_t* n_checked) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    int64_t bad = 0, unres = 0, checked = 0;
    const int backs[4] = {0, 3, 17, 40};
    for (int64_t p = 0; p < n;) {
        const int64_t e = scan_piece(g, p, T.pat_flags);
        for (int bi = 0; bi < 4; ++bi) {
            const int64_t base = p - backs[bi];
            if (base < 0) continue;
            BitWin w;
            for (int k = 0; k < MK_COUNT; ++k) w.m[k] = 0;
            for (int i = 0; i < 64; ++i) {
                const uint32_t v = g.cf(base + i), vp = (base + i > 0) ? g.cf(base + i - 1) : 0u;
                const uint32_t bits = mask_bits_of(vp, v, T.pat_flags);
                for (int k = 0; k < MK_COUNT; ++k) w.m[k] |= (uint64_t)((bits >> k) & 1u) << i;
            }
            {   // the kernel derives SYNC with sync_word() from the other masks: must equal is_sync() per byte
                const uint64_t SL = w.m[MK_TR] & ~w.m[MK_CR];
                const uint32_t vprev = base > 0 ? g.cf(base - 1) : 0u;
                const uint32_t pf = base > 0 ? (feature_of_class(vprev & CLS_MASK) | ((vprev & F_CONT) ? FB_C : 0u)) : 0u;
                const uint64_t sy = sync_word(w.m[MK_U], w.m[MK_W], w.m[MK_X], w.m[MK_S], w.m[MK_N], w.m[MK_CR], SL,
                                              w.m[MK_C], w.m[MK_D], w.m[MK_A], pf, T.pat_flags);
                if (sy != w.m[MK_SYNC]) ++bad;
                // and the feature byte must reproduce the class-set masks
                for (int i = 0; i < 64; ++i) {
                    const uint32_t v = g.cf(base + i);
                    const uint32_t fb = feature_of_class(v & CLS_MASK);
                    const bool okf = (((w.m[MK_U] >> i) & 1) == ((fb & FB_U) != 0)) && (((w.m[MK_W] >> i) & 1) == ((fb & FB_W) != 0)) &&
                                     (((w.m[MK_X] >> i) & 1) == ((fb & FB_X) != 0)) && (((w.m[MK_S] >> i) & 1) == ((fb & FB_S) != 0)) &&
                                     (((w.m[MK_N] >> i) & 1) == fb_is_num(fb)) && (((w.m[MK_CR] >> i) & 1) == ((fb & FB_CR) != 0)) &&
                                     (((SL >> i) & 1) == ((fb & FB_SL) != 0)) && (((w.m[MK_A] >> i) & 1) == fb_is_apos(fb)) &&
                                     (((w.m[MK_SP] >> i) & 1) == fb_is_sp(fb));
                    if (!okf) ++bad;
                }
            }
            {   // the kernel's per-8-byte path: feature bytes -> 8x8 transpose -> mask slices -> sync_byte
                for (int sl = 0; sl < 8; ++sl) {
                    uint64_t F = 0;
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t v = g.cf(base + sl * 8 + i);
                        F |= (uint64_t)(feature_of_class(v & CLS_MASK) | ((v & F_CONT) ? FB_C : 0u)) << (8 * i);
                    }
                    const uint64_t P = transpose8x8(F);
                    const uint32_t pU = P & 0xFF, pW = (P >> 8) & 0xFF, pX = (P >> 16) & 0xFF, pS = (P >> 24) & 0xFF;
                    const uint32_t pNr = (P >> 32) & 0xFF, pCR = (P >> 40) & 0xFF, pSL = (P >> 48) & 0xFF, pC = (P >> 56) & 0xFF;
                    const uint32_t mN = pNr & ~pX & ~pS, mA = pNr & pX, mSP = pNr & pS;
                    const int64_t pos = base + sl * 8;
                    const uint32_t vprev = pos > 0 ? g.cf(pos - 1) : 0u;
                    const uint32_t pf = pos > 0 ? (feature_of_class(vprev & CLS_MASK) | ((vprev & F_CONT) ? FB_C : 0u)) : 0u;
                    const uint32_t D = (uint32_t)(w.m[MK_D] >> (8 * sl)) & 0xFF;
                    const uint32_t sy = sync_byte(pU, pW, pX, pS, mN, pCR, pSL, pC, D, mA, pf, T.pat_flags);
                    auto sl8 = [&](int k) { return (uint32_t)(w.m[k] >> (8 * sl)) & 0xFFu; };
                    if (pU != sl8(MK_U) || pW != sl8(MK_W) || pX != sl8(MK_X) || pS != sl8(MK_S) || mN != sl8(MK_N) ||
                        pCR != sl8(MK_CR) || (pCR | pSL) != sl8(MK_TR) || pC != sl8(MK_C) || mA != sl8(MK_A) ||
                        mSP != sl8(MK_SP) || sy != sl8(MK_SYNC))
                        ++bad;
                }
            }
            int avail = 64;
            auto bytes = [&](int i) { return g.byte(base + i); };
            if (p - base < 32) {  // the kernel's hot path: a 32-bit view cut out of the 64-bit window at the piece start
                BitWin32 v;
                const int off = (int)(p - base);
                for (int k = 0; k < MK_COUNT; ++k) v.m[k] = (uint32_t)(w.m[k] >> off);
                auto b32 = [&](int i) { return g.byte(p + i); };
                const int r32 = scan_piece_p(WinP32(v, 32), b32, T.pat_flags);
                if (r32 >= 0 && p + r32 != e) ++bad;
                // the branch-free form of the common alternatives: whenever it answers, it answers the same
                for (int av : {32, 24, 9}) {
                    const int f32 = scan_piece_fast32(v, av, b32, T.pat_flags);
                    if (f32 >= 0 && p + f32 != e) ++bad;
                    if (av == 32) { ++g_fast_total; g_fast_hit += f32 >= 0; }
                }
            }
            const int r = scan_piece_bits(w, bytes, (int)(p - base), avail, T.pat_flags);
            ++checked;
            if (r < 0) { ++unres; if (e - base < 58 - 3) { /* short piece must resolve unless look-ahead is long */ } }
            else if (base + r != e) ++bad;
        }
        p = e;
    }
    if (n_unres) *n_unres = unres;
    if (n_checked) *n_checked = checked;
    return bad;
}

// ArrMaskP (mask words in an array, unlimited run length) vs byte scanner on every true piece start of the text.
int64_t twin_arrmask_check(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, int64_t* n_checked) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    const int64_t nw = (n + 4 + 63) / 64 + 1;
    std::vector<uint64_t> arr((size_t)nw * MK_COUNT, 0);
    for (int64_t i = 0; i < nw * 64; ++i) {
        const uint32_t v = g.cf(i), vp = i > 0 ? g.cf(i - 1) : 0u;
        const uint32_t bits = mask_bits_of(vp, v, T.pat_flags);
        for (int k = 0; k < MK_COUNT; ++k)
            if ((bits >> k) & 1u) arr[(size_t)(i >> 6) * MK_COUNT + k] |= 1ull << (i & 63);
    }
    int64_t bad = 0, checked = 0;
    auto bytes = [&](int i) { return g.byte(i); };
    for (int64_t p = 0; p < n;) {
        const int64_t e = scan_piece(g, p, T.pat_flags);
        if (n + 4 < 0x7FFFFFFF) {
            ArrMaskP mp(arr.data(), (int)p, (int)(n + 4));
            const int r = scan_piece_p(mp, bytes, T.pat_flags);
            ++checked;
            if (r != (int)e) ++bad;
        }
        p = e;
    }
    if (n_checked) *n_checked = checked;
    return bad;
}

// split_unresolved_heads (the whole-word boundary rules) on 64-byte windows at every multiple of 32: a head the rules
// call resolved must be a piece whose end is the next synchronisation point.  stats: [0] heads, [1] unresolved heads,
// [2] pieces, [3] pieces inside unresolved regions (what the piece-by-piece matcher still has to do).
int64_t twin_word_rules_check(void* h, const uint8_t* text, int64_t n, const int64_t* offs, int64_t n_docs, int64_t* stats,
                              uint8_t* head_state /* NULL or [n]: 1 = resolved head, 2 = unresolved head */) {
    Twin* t = (Twin*)h;
    const Tables T = t->H.view();
    std::vector<uint8_t> cls;
    classify_all(t->H.view(), text, n, offs, n_docs, cls);
    GAcc g{cls.data(), text, n, n + 4, T.pat_flags};
    const int64_t m = n + 96;
    std::vector<uint16_t> bits((size_t)m);
    for (int64_t i = 0; i < m; ++i) bits[(size_t)i] = (uint16_t)mask_bits_of(i > 0 ? g.cf(i - 1) : 0u, g.cf(i), T.pat_flags);
    std::vector<uint8_t> start((size_t)n + 1, 0);
    for (int64_t p = 0; p < n;) { start[(size_t)p] = 1; p = scan_piece(g, p, T.pat_flags); }
    start[(size_t)n] = 1;
    int64_t bad = 0, heads = 0, unres = 0, pieces = 0, upieces = 0;
    for (int64_t base = 0; base < n; base += 32) {
        BitWin w;
        for (int k = 0; k < MK_COUNT; ++k) w.m[k] = 0;
        for (int i = 0; i < 64; ++i)
            for (int k = 0; k < MK_COUNT; ++k) w.m[k] |= (uint64_t)((bits[(size_t)(base + i)] >> k) & 1u) << i;
        uint64_t extra = 0;
        const uint64_t un = split_unresolved_heads(w, T.pat_flags, &extra);
        if (un & ~w.m[MK_SYNC]) ++bad;
        for (int i = 0; i < 64 && base + i < n; ++i)  // an extra start is a piece start, whatever its region counts as
            if (((extra >> i) & 1ull) && !start[(size_t)(base + i)]) ++bad;
        for (int i = 0; i < 32 && base + i < n; ++i) {
            if (!((w.m[MK_SYNC] >> i) & 1ull)) continue;
            const int64_t hd = base + i;
            if (!start[(size_t)hd]) { ++bad; continue; }  // a synchronisation point is a piece start
            int64_t nx = hd + 1;
            while (nx < n && !((bits[(size_t)nx] >> MK_SYNC) & 1u)) ++nx;
            int64_t np = 0;
            for (int64_t q = hd; q < nx; ++q) np += start[(size_t)q];
            ++heads; pieces += np;
            if (head_state) head_state[hd] = ((un >> i) & 1ull) ? 2 : 1;
            if ((un >> i) & 1ull) { ++unres; upieces += np; }
            else {
                // resolved: the pieces of the region are its head and the extra starts the rules placed inside it
                int64_t ne = 0;
                for (int64_t q = hd + 1; q < nx && q < base + 64; ++q) ne += (extra >> (q - base)) & 1ull;
                if (np != 1 + ne) ++bad;
            }
        }
    }
    if (stats) { stats[0] = heads; stats[1] = unres; stats[2] = pieces; stats[3] = upieces; }
    return bad;
}

// generic split patterns: compile `pattern` (td_regex.cpp) and run the reference's piece loop (rx_next_piece) over one
// document.  -> number of pieces (their [start, end) in starts / ends), -1 when the pattern is not supported (err says why)
int64_t twin_rx_split(const char* pattern, const uint8_t* text, int64_t n, int64_t* starts, int64_t* ends, int64_t cap, char* err, int errcap) {
    static thread_local RxProgram P;
    std::string e;
    if (!rx_compile(pattern, P, e)) {
        if (err && errcap > 0) { strncpy(err, e.c_str(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        return -1;
    }
    struct Acc { const uint8_t* t; uint32_t byte(int64_t i) const { return t[i]; } } acc{text};
    const RxTables T = rx_host_tables();
    int64_t k = 0;
    for (int64_t pos = 0; pos < n;) {
        int64_t ms, me;
        rx_next_piece(P, T, acc, pos, n, ms, me);
        if (k >= cap) return -2;
        starts[k] = ms; ends[k] = me; ++k;
        pos = me;
    }
    return k;
}

}  // extern "C"
