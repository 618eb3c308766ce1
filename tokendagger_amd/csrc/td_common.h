// tokendagger_amd — shared host/device definitions for the MI355X tokenizer hot path.
//
// Everything in here is written once and compiled twice: by hipcc for the gfx950 kernels
// (td_kernels.hip) and by the host compiler for the table builder (td_tables.cpp) and the
// CPU "twin" used by the not-gpu tests (tests/twin/td_twin.cpp).  No code here comes from the
// reference; it re-expresses what the reference's hot path computes
// (/root/reference/src/tiktoken/tiktoken.cpp:70-128 split_text, :169-234 encode, :282-378 merge)
// in a form that maps onto 64-wide wavefronts:
//   * the PCRE2 split pattern becomes a deterministic byte-wise scanner over a 1-byte/byte
//     class+flag array (scan_piece), restartable at provable synchronisation points (is_sync);
//   * the byte-string keyed emhash8 encoder map (tiktoken.hpp:41) becomes two open-addressing
//     tables in HBM: piece bytes -> rank (whole-piece fast path, tiktoken.cpp:209-215) and
//     (left id, right id) -> rank (the merge loop's get_rank, tiktoken.cpp:282-296).
#pragma once
#include <stdint.h>

#include "../../include/tokendagger_hip.h"  // TD_OK / TD_E_* codes shared by kernels, host library and C ABI

#if defined(__HIP__)  // HIP language mode (hipcc compiles .cpp as HIP too)
#define TD_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TD_HD inline
#endif

namespace td {

// ------------------------------------------------------------------ character classes -------
// 4-bit class of a code point (tools/gen_unicode_classes.py, probed from PCRE2 10.39 / Unicode 14)
enum : uint32_t {
    C_OTHER = 0,  // anything else: punctuation, symbols, controls, unassigned (in X and P)
    C_APOS = 1,   // U+0027 (in X and P; starts a contraction)
    C_SLASH = 2,  // U+002F (in X and P; also in the [\r\n/]* trailer)
    C_SP = 3,     // U+0020
    C_WS = 4,     // other \s that is not CR/LF
    C_CRLF = 5,   // U+000A, U+000D
    C_UP = 6,     // Lu | Lt
    C_LW = 7,     // Ll
    C_LB = 8,     // Lm | Lo   (member of both letter classes of the pattern)
    C_MK = 9,     // M         (both letter classes, and also prefix / punctuation class)
    C_NUM = 10,   // N
};
// flag bits stored next to the class in the per-byte class array
enum : uint32_t {
    CLS_MASK = 0x0F,
    F_CONT = 0x10,   // byte continues the character of the previous byte (carries that char's class)
    F_MISS = 0x20,   // piece starting here missed the whole-piece table -> needs the merge loop
    F_START = 0x40,  // a regex piece starts at this byte
    F_DOC = 0x80,    // a document starts at this byte (== end of subject for the previous document)
};

#define TD_BIT(c) (1u << (c))
constexpr uint32_t M_U = TD_BIT(C_UP) | TD_BIT(C_LB) | TD_BIT(C_MK);                  // [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]
constexpr uint32_t M_W = TD_BIT(C_LW) | TD_BIT(C_LB) | TD_BIT(C_MK);                  // [\p{Ll}\p{Lm}\p{Lo}\p{M}]
constexpr uint32_t M_L = TD_BIT(C_UP) | TD_BIT(C_LW) | TD_BIT(C_LB);                  // \p{L}
constexpr uint32_t M_S = TD_BIT(C_SP) | TD_BIT(C_WS) | TD_BIT(C_CRLF);                // \s
constexpr uint32_t M_P = 0x7FFu & ~(TD_BIT(C_CRLF) | M_L | TD_BIT(C_NUM));            // [^\r\n\p{L}\p{N}]
constexpr uint32_t M_X = 0x7FFu & ~(M_S | M_L | TD_BIT(C_NUM));                       // [^\s\p{L}\p{N}]
constexpr uint32_t M_TRAIL = TD_BIT(C_CRLF) | TD_BIT(C_SLASH);                        // [\r\n/]
TD_HD bool in_set(uint32_t mask, uint32_t c) { return (mask >> c) & 1u; }

// ------------------------------------------------------------------ table layouts -----------
constexpr int32_t NO_RANK = 0x7FFFFFFF;       // the reference's INT_MAX "no such pair" (tiktoken.cpp:293)
constexpr uint32_t TOK_NONE = 0xFFFFFFFFu;    // empty slot in the byte-indexed token array
constexpr uint32_t TOK_LONGREF = 0x80000000u; // | index into the long-piece list
constexpr uint32_t TOK_MISS = 0x40000000u;    // | tile position << 7 | length: a piece of 2..64 bytes that is not a token; td_merge_tiles
                                              // replaces the slot by the ids its byte-pair merge produces
constexpr uint32_t TOK_MERGED = 0x20000000u;  // in a TOK_MISS slot: the piece is merged, the low 7 bits count its ids (position << 7 stays)
constexpr uint32_t TOK_DUPREF = 0x10000000u;  // in a merged TOK_MISS slot (td_copy_dups): the piece repeats another one — bits 7..27 name the seat of the table of
                                              // distinct pieces (EncodeArgs::dd_table) whose record says where that piece's ids are (instead of a tile position):
                                              // the pack kernels copy them from there, nothing was copied in between
constexpr uint32_t TOK_OVF = 0x08000000u;     // in a TOK_MISS slot: its record found no room on a list; td_merge_pieces' scan behind the rows merges it
// tile_count[]: slots of the tile (bits 0..12) | length classes that found no room on td_collect_misses' lists (bits 13..17) | flags
constexpr uint32_t TILE_HAS_LONG = 0x80000000u, TILE_HAS_MISS = 0x40000000u, TILE_COUNT_MASK = 0x1FFFu;
constexpr uint32_t TILE_DIRECT = 0x10000000u;       // the fused tile loop wrote the tile's ids and document offsets itself: td_pack_tokens skips it
constexpr uint32_t TILE_MISS_LISTED = 0x20000000u;  // the tile's (few) missed pieces are on the global miss list: td_merge_pieces need not scan its slots
constexpr int K_MISS_LISTED_MAX = 6;                // more missed pieces than this in a tile: TILE_HAS_MISS instead
constexpr int COLL_SUBS = 11;                       // td_collect_misses: lists per length class (a wavefront appends to list gw % COLL_SUBS: same-address
constexpr int COLL_STRIDE = 32;                     //   atomics are served one after the other), their counters COLL_STRIDE words apart
constexpr uint32_t TILE_OVF_SHIFT = 13;             // tile_count bits 13..17: length classes of the tile whose records found no room on a list
constexpr int K_MISS_CLASSES = 5;                   // one list per length class (<= 8, 16, 32, 48, 64 bytes): a row of a list is one batch
constexpr int ID_BITS = 21;                   // ids / ranks must be < 2^21 (pair slots pack 2 ids + rank in 64 bit)
constexpr uint64_t PAIR_EMPTY = ~0ull;
constexpr uint64_t PAIR_KEY_MASK = (1ull << (2 * ID_BITS)) - 1;  // (left id << ID_BITS | right id) of a slot >> ID_BITS
// bit 63 of a pair slot, set: NO pair whose first seat (hash_pair) is this slot sits in its second one (hash_pair2) — a probe of
// the first seat that does not find its pair there is final, and nine in ten are: the merge rounds are bound by the number of
// probes the vector L1 takes, not by their latency.  Clear: look at the second seat too.
constexpr uint64_t PAIR_FINAL = 1ull << 63;

struct PieceSlot {  // 16 B; len == 0 marks an empty slot
    uint64_t key;   // len <= 8: the bytes, little-endian, zero padded; len > 8: hash_bytes()
    uint32_t rank;
    uint32_t len;
};

struct Piece12Slot {  // 16 B: every token of 1..12 bytes with its exact key: ONE 16-byte load answers a probe, no verification
    uint32_t k0, k1, k2;  // the bytes, little-endian, zero padded
    uint32_t meta;        // rank | len << 24 | 1 << 31; 0 marks an empty slot
};
constexpr uint32_t P12_MAXLEN = 12;
TD_HD uint32_t p12_meta(uint32_t rank, uint32_t len) { return rank | (len << 24) | 0x80000000u; }

struct Tables {
    const uint8_t* ascii_cls;     // [128]
    const uint16_t* ucls1;        // [4352]   code point >> 8 -> block
    const uint8_t* ucls2;         // [nblocks*256]
    const int32_t* byte_id;       // [256]    id of the 1-byte token, or pseudo id (>= pseudo_base) if absent
    const int32_t* byte_pair;     // [65536]  rank of the 2-byte token (b0<<8|b1) or NO_RANK
    const uint64_t* byte_pair_id; // [65536]  the same | id of the 1-byte token b0 << 32: what the lane-per-piece merge needs to set a part up, in ONE load
    const PieceSlot* piece_slots; // open addressing, linear probing
    const uint64_t* pair_slots;   // cuckoo table: (left<<42 | right<<21 | rank), PAIR_EMPTY if empty
    const Piece12Slot* piece12_slots;  // open addressing, linear probing, inserted in rank order (tokens of 1..12 bytes; they are also in piece_slots)
    const uint32_t* tok_off;      // [max_id+2] byte offsets of token id's bytes (decode + long-key verify)
    const uint8_t* tok_bytes;
    uint32_t piece_mask;
    uint32_t pair_mask;
    int32_t max_id;               // largest real id
    int32_t pseudo_base;          // ids >= pseudo_base stand for single bytes that are not tokens
    uint32_t max_token_len;
    uint32_t piece12_mask;
    uint32_t pat_flags;           // PV_* bits: which member of the split-pattern family the scanners implement
    // character seeds (round 5, below): null = none
    const uint64_t* cseed;        // [65536] by code point (characters of 2 and 3 bytes): CS_VALID | row of cseed_nm << 29 | row of cseed_pm << 21 | id
    const uint32_t* cseed_pm;     // [rows][8] 256-bit sets: a byte that must not stand in FRONT of the character for it to be seeded
    const uint32_t* cseed_nm;     // [rows][8] ... BEHIND it
};

// The split patterns the scanners implement are one family: the Llama-4 / o200k pattern (reference src/main.cpp:114)
// and its Mistral "tekken" sibling (tekken.json config.pattern, reference tests/throughput_test.py:118), which drops the
// (?i:'s|'t|'re|'ve|'m|'ll|'d)? suffix of the two letter alternatives and matches \p{N} instead of \p{N}{1,3}.
// Third member: the cl100k_base / Llama-3 pattern
//   (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
// : the contraction is an alternative of its own in front (not a suffix), letters are plain \p{L}+ (no case structure;
// marks are punctuation) and '/' does not extend a punctuation piece.  The last two are class REMAPS done when the
// tables are built (C_MK -> C_OTHER, C_SLASH -> C_OTHER), so only the first two reach the scanners as flags.
// Fourth member: the GPT-2 pattern (r50k_base / p50k_base)
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
// : case-sensitive contractions in front, the optional prefix is U+0020 only (also before digits), digit runs are not
// cut, no [\r\n/]* trailer and no \s*[\r\n]+ alternative.  It has a scanner of its own (scan_piece_gpt2*), selected
// by PV_GPT2, and the same two class remaps as cl100k.
// PV_WS_EOS_FIRST: current tiktoken releases spell cl100k_base with `\s++$` IN FRONT of `\s*[\r\n]`: a whitespace run that
// reaches the end of the subject is one piece even when it contains CR/LF ("\r\t" at the end: one piece, not two).
enum : uint32_t { PV_NO_CONTRACTION = 1, PV_SINGLE_DIGIT = 2, PV_LEADING_CONTRACTION = 4, PV_PLAIN_LETTERS = 8, PV_GPT2 = 16, PV_WS_EOS_FIRST = 32,
                  PV_GENERIC = 64 };  // PV_GENERIC: not a member of the family: the compiled pattern (td_regex.h) is matched document by document (td_generic.hip)

// ------------------------------------------------------------------ hashing -----------------
TD_HD uint32_t hash_piece(uint64_t key, uint32_t len) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = lo * 0x9E3779B1u;
    h ^= (hi * 0x85EBCA77u) >> 7 | (hi * 0x85EBCA77u) << 25;
    h ^= len * 0xC2B2AE3Du;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}
TD_HD uint32_t hash_pair(uint32_t left, uint32_t right) {
    uint32_t h = left * 0x9E3779B1u + right * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}
// 64-bit key of a piece longer than 8 bytes; `get(i)` returns byte i.
template <class Get>
TD_HD uint64_t hash_bytes(const Get& get, uint32_t len) {
    uint64_t k = 0x243F6A8885A308D3ull ^ len;
    for (uint32_t i = 0; i < len; i += 8) {
        uint64_t w = 0;
        for (uint32_t j = 0; j < 8 && i + j < len; ++j) w |= (uint64_t)get(i + j) << (8 * j);
        k = ((k << 23) | (k >> 41)) ^ w;
        k *= 0x9E3779B97F4A7C15ull;
    }
    return k ^ (k >> 31);
}

TD_HD uint32_t hash_piece12(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len) {
    uint32_t h = (k0 ^ (len * 0xC2B2AE3Du)) * 0x9E3779B1u;
    h ^= h >> 15;
    h += k1 * 0x85EBCA77u;
    h ^= k2 * 0x27D4EB2Fu;
    h *= 0x2C1B3C6Du;
    return h ^ (h >> 13);
}

// (left id, right id) -> rank of the concatenation, NO_RANK if it is not a token.  The pair table is a CUCKOO table
// (two hash functions, one entry per slot): a lookup is exactly two independent 8-byte loads and no loop, so the
// loads of several lookups can be in flight together (the merge rounds are bound by this latency).
TD_HD uint32_t hash_pair2(uint32_t left, uint32_t right) {
    uint32_t h = (left ^ 0x68E31DA4u) * 0xB5297A4Du + (right ^ 0x1B56C4E9u) * 0x68E31DA5u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    return h;
}
TD_HD int32_t pair_match(uint64_t e1, uint64_t e2, uint32_t left, uint32_t right) {
    // (selects, no early return: with one, the compiler sinks the second slot's load behind the first compare and a
    // lookup becomes two dependent round trips)
    const uint64_t key = ((uint64_t)left << ID_BITS) | right;
    const bool m1 = ((e1 >> ID_BITS) & PAIR_KEY_MASK) == key, m2 = ((e2 >> ID_BITS) & PAIR_KEY_MASK) == key;
    const uint32_t v = (uint32_t)(m1 ? e1 : e2) & ((1u << ID_BITS) - 1);
    return (m1 || m2) ? (int32_t)v : NO_RANK;  // (an empty slot is all ones and matches no key: ids are < 2^21 - 1)
}
// one slot against a pair: does it hold it?
TD_HD bool pair_slot_is(uint64_t e, uint32_t left, uint32_t right) { return ((e >> ID_BITS) & PAIR_KEY_MASK) == (((uint64_t)left << ID_BITS) | right); }
TD_HD int32_t pair_lookup(const Tables& T, uint32_t left, uint32_t right) {
    const uint64_t e1 = T.pair_slots[hash_pair(left, right) & T.pair_mask];
    const uint64_t e2 = T.pair_slots[hash_pair2(left, right) & T.pair_mask];
    return pair_match(e1, e2, left, right);
}

// ------------------------------------------------------------------ character seeds (round 5) ----
// The merge loop (bpe_merge, tiktoken.cpp:298-368) starts from single bytes: a CJK sentence of 27 characters is 81 parts and ~75
// rounds, most of which only put the characters together.  A multi-byte character c whose bytes merge to ONE token C may be entered
// as the single part C when that provably changes nothing.  Conditions (td_tables.cpp: build_char_seeds checks 1, 3, 4 per vocabulary;
// 2 is checked per occurrence, here):
//   1. the merge loop on c's bytes alone ends with [C];
//   2. no token of the vocabulary OCCURS in the piece overlapping c partially (some of c's bytes and at least one byte outside): such
//      a token ends with a proper prefix of c preceded by the byte in front of c, or starts with a proper suffix of c followed by the
//      byte behind c — per character two 256-bit sets of bytes that must not stand in front of / behind it (exact in the adjacent
//      byte, conservative beyond it);
//   3. every pair of the pair table with C as its left or right part ranks ABOVE every merge inside c (rho(c) = the highest rank
//      among the merges of 1.; in a vocabulary that BPE training produced, rank(C) itself);
//   4. C's id is its rank (a regular token).
// Why that is exact.  Every part the loop ever holds is a token (or a byte) that occurs in the piece where the part stands.  By 2. no
// part covers c partially, so until c is complete its bytes merge only among themselves, in the order of 1., whatever happens around
// them, and the loop cannot end before c is complete.  Let R be the reference's run on bytes and S the run with every seeded character
// entered whole.  Claim: R's merges that are not inside an incomplete seeded character are S's merges, in S's order.  Induction: let e
// be S's next merge (lowest rank, leftmost).  R's ranked pairs outside incomplete characters are a subset of S's (pairs between a part
// and a piece of an incomplete character have no rank by 2.), so none of them ranks below e or ties left of it.  If e's parts exist
// in R, R merges e next, possibly after merges inside incomplete characters; a character C' they complete only adds pairs S had
// all along.  If e involves a character C that R has not completed, e is C's first merge in S, so rank(e) > rho(c) by 3., i.e. above
// every pending merge inside c: R, which takes the globally lowest rank, completes c (and nothing ranked >= rank(e) elsewhere) first,
// and then is in the first case.  When S ends, R has only merges inside characters left.  Same parts at the end.
// tests/test_char_seeds.py: the CPU twin's seeded merge against the reference's loop on a few hundred thousand pieces of all scripts.
constexpr uint64_t CS_VALID = 1ull << 63;
// Does a seeded character start at byte q of the piece get(0..len)?  -> its length in k, its id in id.
template <class Get>
TD_HD bool cseed_char_at(const Tables& T, Get get, uint32_t len, uint32_t q, uint32_t& k, uint32_t& id) {
    const uint32_t b0 = get(q);
    if (b0 < 0xC2u || b0 >= 0xF0u) return false;
    k = b0 < 0xE0u ? 2u : 3u;
    if (q + k > len) return false;
    const uint32_t b1 = get(q + 1u);
    if ((b1 & 0xC0u) != 0x80u) return false;
    uint32_t cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
    if (k == 3u) {
        const uint32_t b2 = get(q + 2u);
        if ((b2 & 0xC0u) != 0x80u) return false;
        cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
        if (cp < 0x800u) return false;  // (an overlong form is not the character's bytes)
    }
    const uint64_t e = T.cseed[cp];
    if (!(e & CS_VALID)) return false;
    if (q > 0u) {
        const uint32_t pb = get(q - 1u);
        if ((T.cseed_pm[(uint32_t)((e >> 21) & 0xFFu) * 8u + (pb >> 5)] >> (pb & 31u)) & 1u) return false;
    }
    if (q + k < len) {
        const uint32_t nb = get(q + k);
        if ((T.cseed_nm[(uint32_t)((e >> 29) & 0xFFu) * 8u + (nb >> 5)] >> (nb & 31u)) & 1u) return false;
    }
    id = (uint32_t)e & 0x1FFFFFu;
    return true;
}
// What stands at byte q: kind 0 = inside a seeded character (no part starts here), 1 = a single byte is a part, 2 = a seeded
// character of k bytes with id `id` starts here.  back = bytes from q back to the start of the part that holds q (kind 0) or 0.
struct SeedPart { uint32_t kind, k, id, back; };
template <class Get>
TD_HD SeedPart cseed_part_at(const Tables& T, Get get, uint32_t len, uint32_t q) {
    SeedPart sp{1u, 1u, 0u, 0u};
    if (!T.cseed) return sp;
    const uint32_t b = get(q);
    if (b >= 0xC2u) {
        uint32_t k = 0, id = 0;
        if (cseed_char_at(T, get, len, q, k, id)) { sp.kind = 2u; sp.k = k; sp.id = id; }
    } else if ((b & 0xC0u) == 0x80u) {
        for (uint32_t d = 1; d <= 2u && d <= q; ++d) {
            const uint32_t lb = get(q - d);
            if ((lb & 0xC0u) == 0x80u) continue;  // (another continuation byte: the lead may be one further back)
            uint32_t k = 0, id = 0;
            if (lb >= 0xC2u && cseed_char_at(T, get, len, q - d, k, id) && k > d) { sp.kind = 0u; sp.back = d; }
            break;
        }
    }
    return sp;
}

// piece bytes -> rank, NO_RANK if the piece is not a token.  `get(i)` returns byte i of the piece.
template <class Get>
TD_HD int32_t piece_lookup(const Tables& T, uint64_t key, uint32_t len, const Get& get) {
    uint32_t h = hash_piece(key, len) & T.piece_mask;
    for (;;) {
        const PieceSlot s = T.piece_slots[h];
        if (s.len == 0) return NO_RANK;
        if (s.key == key && s.len == len) {
            if (len <= 8) return (int32_t)s.rank;
            const uint8_t* tb = T.tok_bytes + T.tok_off[s.rank];
            bool same = true;
            for (uint32_t i = 0; i < len; ++i)
                if (tb[i] != get(i)) { same = false; break; }
            if (same) return (int32_t)s.rank;
        }
        h = (h + 1) & T.piece_mask;
    }
}

// ------------------------------------------------------------------ UTF-8 classification ----
// Declared sequence length of a lead byte (1 for ASCII and for bytes that cannot lead).
TD_HD uint32_t utf8_declared_len(uint32_t b) {
    if (b >= 0xC2 && b <= 0xDF) return 2;
    if (b >= 0xE0 && b <= 0xEF) return 3;
    if (b >= 0xF0 && b <= 0xF4) return 4;
    return 1;
}
TD_HD uint32_t class_of_cp(const Tables& T, uint32_t cp) {
    if (cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) return C_OTHER;
    return T.ucls2[(uint32_t)T.ucls1[cp >> 8] * 256u + (cp & 255u)];
}
// Class (+F_CONT) of the byte at index i.  S provides: lo (first readable index), hi (one past the
// last readable index), byte(i), doc(i) (document starts at i).  A lead byte followed by fewer
// continuation bytes than it declares forms one C_OTHER character with the ones that are there;
// stray continuation bytes and invalid lead bytes are 1-byte C_OTHER characters; characters never
// cross a document start.  (The reference assumes valid UTF-8: PCRE2_NO_UTF_CHECK, tiktoken.cpp:91.)
template <class S>
TD_HD uint32_t classify_at(const Tables& T, const S& s, int64_t i) {
    const uint32_t b = s.byte(i);
    if (b < 0x80) return T.ascii_cls[b];
    int64_t lead = i;
    if ((b & 0xC0) == 0x80) {
        bool found = false;
        if (!s.doc(i)) {
            for (int k = 1; k <= 3; ++k) {
                const int64_t j = i - k;
                if (j < s.lo) break;
                const uint32_t c = s.byte(j);
                if ((c & 0xC0) != 0x80) {
                    if (utf8_declared_len(c) > (uint32_t)k) { found = true; lead = j; }
                    break;
                }
                if (s.doc(j)) break;  // a continuation byte that starts a document is a stray
            }
        }
        if (!found) return C_OTHER;
    }
    const uint32_t lb = s.byte(lead);
    const uint32_t need = utf8_declared_len(lb) - 1;
    if (need == 0) return C_OTHER;
    uint32_t cp = lb & (0xFFu >> (need + 2));
    uint32_t got = 0;
    while (got < need) {
        const int64_t j = lead + 1 + got;
        if (j >= s.hi || s.doc(j)) break;
        const uint32_t c = s.byte(j);
        if ((c & 0xC0) != 0x80) break;
        cp = (cp << 6) | (c & 0x3F);
        ++got;
    }
    const uint32_t cont = (lead != i) ? F_CONT : 0u;
    if (got < need) return C_OTHER | cont;
    return class_of_cp(T, cp) | cont;
}

// ------------------------------------------------------------------ synchronisation points --
// A position whose class/flag byte is `v` (previous byte's: `vp`) is PROVABLY a piece start,
// whatever came before, when one of these holds (each follows from which alternatives of the
// Llama-4 pattern can contain the two characters; verified by tests/test_twin.py against PCRE2):
//   R1 a document starts here;
//   R2 a non-CR/LF whitespace char follows a non-whitespace char (whitespace is only ever consumed
//      at the start of a piece or inside a whitespace-only piece);
//   R3 any non-whitespace char other than '/' follows CR/LF (CR/LF cannot be a prefix char; it only
//      occurs inside whitespace pieces or in the [\r\n/]* trailer, which continues only with \r \n /);
//   R4 a digit run starts or ends here (digits only occur in \p{N}{1,3} pieces);
//   R5 punctuation other than ' follows a true letter (letters only occur in alternatives 1-2,
//      which continue only with letter-class chars or a contraction).
// (GPT-2 pattern: " 12" is one piece, so the START of a digit run proves nothing there; its end still does.)
TD_HD bool is_sync(uint32_t vp, uint32_t v, uint32_t pv = 0) {
    if (v & F_CONT) return false;
    if (v & F_DOC) return true;
    const uint32_t c = v & CLS_MASK, p = vp & CLS_MASK;
    if ((c == C_SP || c == C_WS) && !in_set(M_S, p)) return true;
    if (p == C_CRLF && !in_set(M_S, c) && c != C_SLASH) return true;
    if ((pv & PV_GPT2) ? (p == C_NUM && c != C_NUM) : ((c == C_NUM) != (p == C_NUM))) return true;
    if ((c == C_OTHER || c == C_SLASH) && in_set(M_L, p)) return true;
    return false;
}

// ------------------------------------------------------------------ the scanner -------------
// Accessor A: pos_t; lim (positions >= lim are not readable); cf(i) class+flags; byte(i) raw byte.
// Positions past the end of the text read as F_DOC sentinels, so "end of subject" is always a
// set F_DOC flag at a position > pos.  Every function returns the piece end (> pos), 0 for "this
// alternative does not match" (helpers only), or -1 when the answer depends on bytes at/after lim.

template <class A>
TD_HD typename A::pos_t scan_contraction(const A& a, typename A::pos_t e, bool leading = false) {
    // (?i:'s|'t|'re|'ve|'m|'ll|'d)?  — caseless under UTF+UCP, so U+017F also matches the s.
    // leading: the apostrophe is the piece start itself (cl100k's first alternative), where a document may begin.
    if (e + 3 > a.lim) return -1;
    uint32_t v = a.cf(e);
    if (((v & F_DOC) && !leading) || (v & CLS_MASK) != C_APOS) return e;
    v = a.cf(e + 1);
    if (v & F_DOC) return e;
    const uint32_t b1 = a.byte(e + 1);
    const uint32_t v2 = a.cf(e + 2);
    if (b1 < 0x80) {
        const uint32_t l1 = b1 | 0x20;
        if (l1 == 's' || l1 == 't' || l1 == 'm' || l1 == 'd') return e + 2;
        if (v2 & F_DOC) return e;
        const uint32_t b2 = a.byte(e + 2);
        if (b2 < 0x80) {
            const uint32_t l2 = b2 | 0x20;
            if ((l1 == 'r' && l2 == 'e') || (l1 == 'v' && l2 == 'e') || (l1 == 'l' && l2 == 'l')) return e + 3;
        }
        return e;
    }
    if (b1 == 0xC5 && !(v2 & F_DOC) && a.byte(e + 2) == 0xBF) return e + 3;
    return e;
}

// alternatives 1 (U* W+) and 2 (U+ W*) from `st` (== pos, or the byte after the 1-char prefix)
template <class A>
TD_HD typename A::pos_t scan_letters(const A& a, typename A::pos_t pos, typename A::pos_t st, int alt, uint32_t pv) {
    using P = typename A::pos_t;
    P q = st, lastw_end = 0;
    uint32_t c = 255;
    bool eos = false;
    for (;;) {
        if (q >= a.lim) return -1;
        const uint32_t v = a.cf(q);
        if (q > pos && (v & F_DOC)) { eos = true; break; }
        c = v & CLS_MASK;
        if (!in_set(M_U, c)) break;
        if (in_set(M_W, c)) lastw_end = q + 1;
        ++q;
    }
    P e;
    const bool w_follows = !eos && in_set(M_W, c);
    if (alt == 1) {
        if (w_follows) e = q;
        else if (lastw_end) return (pv & PV_NO_CONTRACTION) ? lastw_end : scan_contraction(a, lastw_end);  // greedy U* gives back to its last W-class char
        else return 0;
    } else {
        if (q == st) return 0;
        e = q;
    }
    if (w_follows) {
        for (;;) {
            if (e >= a.lim) return -1;
            const uint32_t v = a.cf(e);
            if ((e > pos && (v & F_DOC)) || !in_set(M_W, v & CLS_MASK)) break;
            ++e;
        }
    }
    return (pv & PV_NO_CONTRACTION) ? e : scan_contraction(a, e);
}

// The GPT-2 pattern (classes remapped: marks and '/' are C_OTHER).
template <class A>
TD_HD typename A::pos_t scan_piece_gpt2(const A& a, typename A::pos_t pos) {
    using P = typename A::pos_t;
    const uint32_t c0 = a.cf(pos) & CLS_MASK;
    if (c0 == C_APOS) {  // 's|'t|'re|'ve|'m|'ll|'d  (case-sensitive)
        if (pos + 3 > a.lim) return -1;
        const uint32_t v1 = a.cf(pos + 1), v2 = a.cf(pos + 2);
        if (!(v1 & F_DOC)) {
            const uint32_t b1 = a.byte(pos + 1);
            if (b1 == 's' || b1 == 't' || b1 == 'm' || b1 == 'd') return pos + 2;
            if (!(v2 & F_DOC)) {
                const uint32_t b2 = a.byte(pos + 2);
                if ((b1 == 'r' && b2 == 'e') || (b1 == 'v' && b2 == 'e') || (b1 == 'l' && b2 == 'l')) return pos + 3;
            }
        }
    }
    //  ?\p{L}+ |  ?\p{N}+ |  ?[^\s\p{L}\p{N}]+ : the optional space is taken when a run of one of the three kinds follows
    P st = pos;
    uint32_t cls = c0;
    if (c0 == C_SP) {
        if (pos + 1 >= a.lim) return -1;
        const uint32_t v1 = a.cf(pos + 1);
        const uint32_t c1 = v1 & CLS_MASK;
        if (!(v1 & F_DOC) && (in_set(M_U | M_W, c1) || c1 == C_NUM || in_set(M_X, c1))) { st = pos + 1; cls = c1; }
    }
    const uint32_t run_set = in_set(M_U | M_W, cls) ? (M_U | M_W) : (cls == C_NUM) ? (1u << C_NUM) : in_set(M_X, cls) ? M_X : 0u;
    if (run_set) {
        P e = st;
        for (;;) {
            if (e >= a.lim) return -1;
            const uint32_t v = a.cf(e);
            if ((e > pos && (v & F_DOC)) || !in_set(run_set, v & CLS_MASK)) break;
            ++e;
        }
        return e;
    }
    // \s+(?!\S) | \s+ on the maximal whitespace run
    {
        P q = pos, last_lead = pos;
        bool eos = false;
        for (;;) {
            if (q >= a.lim) return -1;
            const uint32_t v = a.cf(q);
            if (q > pos && (v & F_DOC)) { eos = true; break; }
            if (!in_set(M_S, v & CLS_MASK)) break;
            if (!(v & F_CONT)) last_lead = q;
            ++q;
        }
        if (q == pos) {  // not reachable (every class is covered); mirrors the no-progress rule
            P p1 = pos + 1;
            while (p1 < a.lim && (a.cf(p1) & F_CONT)) ++p1;
            return p1;
        }
        if (eos) return q;
        if (last_lead > pos) return last_lead;
        return q;
    }
}

// End of the piece that starts at `pos` (a character start, pos < lim).  pv: PV_* pattern variant bits.
template <class A>
TD_HD typename A::pos_t scan_piece(const A& a, typename A::pos_t pos, uint32_t pv = 0) {
    using P = typename A::pos_t;
    if (pv & PV_GPT2) return scan_piece_gpt2(a, pos);
    const uint32_t c0 = a.cf(pos) & CLS_MASK;
    P p1 = pos + 1;  // end of the first character
    for (;;) {
        if (p1 >= a.lim) return -1;
        if (!(a.cf(p1) & F_CONT)) break;
        ++p1;
    }
    if ((pv & PV_LEADING_CONTRACTION) && c0 == C_APOS) {
        const P r = scan_contraction(a, pos, true);
        if (r < 0) return -1;
        if (r != pos) return r;
    }
    if (c0 != C_CRLF && c0 != C_NUM) {
        const bool prefixable = in_set(M_P, c0);
        // only worth trying when a letter-class char is at pos or right after the prefix char
        const uint32_t v1 = a.cf(p1);
        const bool l0 = in_set(M_U | M_W, c0);
        const bool l1 = prefixable && !(v1 & F_DOC) && in_set(M_U | M_W, v1 & CLS_MASK);
        if ((pv & PV_PLAIN_LETTERS) && (l0 || l1)) {  // [^\r\n\p{L}\p{N}]?\p{L}+
            P e = l1 ? p1 : pos;
            for (;;) {
                if (e >= a.lim) return -1;
                const uint32_t v = a.cf(e);
                if ((e > pos && (v & F_DOC)) || !in_set(M_U | M_W, v & CLS_MASK)) break;
                ++e;
            }
            return e;
        }
        if (l0 || l1) {
            for (int alt = 1; alt <= 2; ++alt) {
                if (l1) {
                    const P r = scan_letters(a, pos, p1, alt, pv);
                    if (r != 0) return r;
                }
                if (l0) {
                    const P r = scan_letters(a, pos, pos, alt, pv);
                    if (r != 0) return r;
                }
            }
        }
    }
    if (c0 == C_NUM) {  // \p{N}{1,3}  (tekken: \p{N})
        P e = p1;
        const int nmax = (pv & PV_SINGLE_DIGIT) ? 1 : 3;
        for (int k = 1; k < nmax; ++k) {
            if (e >= a.lim) return -1;
            const uint32_t v = a.cf(e);
            if ((v & F_DOC) || (v & CLS_MASK) != C_NUM) break;
            ++e;
            for (;;) {
                if (e >= a.lim) return -1;
                if (!(a.cf(e) & F_CONT)) break;
                ++e;
            }
        }
        return e;
    }
    //  ?[^\s\p{L}\p{N}]+[\r\n/]*
    for (int wsp = 1; wsp >= 0; --wsp) {
        if (wsp && c0 != C_SP) continue;
        const P st = wsp ? pos + 1 : pos;
        P e = st;
        for (;;) {
            if (e >= a.lim) return -1;
            const uint32_t v = a.cf(e);
            if (e > pos && (v & F_DOC)) break;
            if (!in_set(M_X, v & CLS_MASK)) break;
            ++e;
        }
        if (e == st) continue;
        for (;;) {
            if (e >= a.lim) return -1;
            const uint32_t v = a.cf(e);
            if ((v & F_DOC) || !in_set(M_TRAIL, v & CLS_MASK)) break;
            ++e;
        }
        return e;
    }
    // \s*[\r\n]+ | \s+(?!\S) | \s+   on the maximal whitespace run starting at pos
    if (in_set(M_S, c0)) {
        P q = pos, last_crlf_end = 0, last_lead = pos;
        bool eos = false;
        for (;;) {
            if (q >= a.lim) return -1;
            const uint32_t v = a.cf(q);
            if (q > pos && (v & F_DOC)) { eos = true; break; }
            const uint32_t c = v & CLS_MASK;
            if (!in_set(M_S, c)) break;
            if (c == C_CRLF) last_crlf_end = q + 1;
            if (!(v & F_CONT)) last_lead = q;
            ++q;
        }
        if (eos && (pv & PV_WS_EOS_FIRST)) return q;
        if (last_crlf_end) return last_crlf_end;
        if (eos) return q;
        if (last_lead > pos) return last_lead;
        return q;
    }
    return p1;  // not reachable for this pattern; mirrors the no-progress rule (tiktoken.cpp:120-122)
}

// ------------------------------------------------------------------ bit-parallel scanner ----
// The same matcher on per-class BITMASKS (one bit per byte) instead of per-byte class codes: run ends
// become count-trailing-zero operations, so a piece costs a few dozen ALU ops instead of a dependent
// LDS read per byte.  A BitWin holds, for 64 consecutive window bytes starting at `base`, one 64-bit
// word per class set; bit i <-> byte base+i.  scan_piece_bits answers for a piece starting at offset
// `o` inside the window, or returns -1 when it would need bits at/after `avail` (caller reloads the
// window at the piece start, and falls back to scan_piece when a single piece outgrows 64 bytes).
enum : int {
    MK_U = 0,   // M_U
    MK_W,       // M_W
    MK_X,       // M_X
    MK_S,       // M_S
    MK_N,       // C_NUM
    MK_CR,      // C_CRLF
    MK_TR,      // M_TRAIL
    MK_C,       // F_CONT
    MK_D,       // F_DOC   (end of subject when seen at an offset > o)
    MK_A,       // C_APOS
    MK_SP,      // C_SP
    MK_SYNC,    // is_sync(prev, this)
    MK_COUNT
};
struct BitWin {
    uint64_t m[MK_COUNT];
};

TD_HD int td_ctz64(uint64_t x) {  // 64 for x == 0
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? (int)__builtin_ctzll(x) : 64;
#else
    return x ? (int)__builtin_ctzll(x) : 64;
#endif
}
TD_HD int td_top64(uint64_t x) { return 64 - (int)__builtin_clzll(x); }  // index of highest set bit + 1 (x != 0)
// first index >= f whose bit in m is 0 (64 if none below 64)
TD_HD int td_run_end(uint64_t m, int f) { return f >= 64 ? 64 : f + td_ctz64(~(m >> f)); }
TD_HD uint64_t td_bits_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// Mask bits of one byte from its class+flag byte `v` and its predecessor's `vp` (the kernel turns these
// into the 64-bit words with one wavefront ballot per mask; the CPU twin ORs them bit by bit).
TD_HD uint32_t mask_bits_of(uint32_t vp, uint32_t v, uint32_t pv = 0) {
    const uint32_t c = v & CLS_MASK;
    uint32_t r = 0;
    r |= in_set(M_U, c) ? (1u << MK_U) : 0u;
    r |= in_set(M_W, c) ? (1u << MK_W) : 0u;
    r |= in_set(M_X, c) ? (1u << MK_X) : 0u;
    r |= in_set(M_S, c) ? (1u << MK_S) : 0u;
    r |= (c == C_NUM) ? (1u << MK_N) : 0u;
    r |= (c == C_CRLF) ? (1u << MK_CR) : 0u;
    r |= in_set(M_TRAIL, c) ? (1u << MK_TR) : 0u;
    r |= (v & F_CONT) ? (1u << MK_C) : 0u;
    r |= (v & F_DOC) ? (1u << MK_D) : 0u;
    r |= (c == C_APOS) ? (1u << MK_A) : 0u;
    r |= (c == C_SP) ? (1u << MK_SP) : 0u;
    r |= is_sync(vp, v, pv) ? (1u << MK_SYNC) : 0u;
    return r;
}
// Mask providers.  A provider answers bit / run questions about the class masks around one piece start `o`;
// end-of-subject marks (MK_D) count only at positions > o.  Positions >= lim are unreadable: run_end may
// return a value >= lim, which the scanner reports as -1 ("needs more look-ahead").
//   WinP : the 64-bit register window (BitWin) — run ends are single ctz operations
//   (the kernel's LdsMaskP walks the mask words in LDS and has no length limit below the staged window)
struct WinP {
    const BitWin& w;
    int o, lim;
    uint64_t E;  // end-of-subject bits after o
    TD_HD WinP(const BitWin& w_, int o_, int avail) : w(w_), o(o_), lim(avail), E(w_.m[MK_D] & ~td_bits_below(o_ + 1)) {}
    TD_HD bool bit(int k, int i) const { return (w.m[k] >> i) & 1ull; }
    TD_HD bool ebit(int i) const { return (E >> i) & 1ull; }
    TD_HD int run_end(int k, int from) const { return td_run_end(w.m[k] & ~E, from); }
    TD_HD int run_end2(int k1, int k2, int from) const { return td_run_end((w.m[k1] | w.m[k2]) & ~E, from); }
    TD_HD int last_and(int k1, int k2, int lo, int hi) const {  // highest i in [lo,hi) with both bits set, -1 if none
        const uint64_t m = w.m[k1] & w.m[k2] & td_bits_below(hi) & ~td_bits_below(lo);
        return m ? td_top64(m) - 1 : -1;
    }
    TD_HD int last_set(int k, int lo, int hi) const {
        const uint64_t m = w.m[k] & td_bits_below(hi) & ~td_bits_below(lo);
        return m ? td_top64(m) - 1 : -1;
    }
    TD_HD int last_clear(int k, int lo, int hi) const {
        const uint64_t m = ~w.m[k] & td_bits_below(hi) & ~td_bits_below(lo);
        return m ? td_top64(m) - 1 : -1;
    }
};

// 32-bit register window whose bit 0 is the piece start itself (o == 0): all run searches are single 32-bit
// ctz operations.  The kernel cuts it out of its 64-bit window with one funnel shift per mask.
struct BitWin32 {
    uint32_t m[MK_COUNT];
};
TD_HD int td_ctz32(uint32_t x) { return x ? (int)__builtin_ctz(x) : 32; }
struct WinP32 {
    const BitWin32& w;
    int o, lim;
    uint32_t E;
    TD_HD WinP32(const BitWin32& w_, int avail) : w(w_), o(0), lim(avail), E(w_.m[MK_D] & ~1u) {}
    TD_HD bool bit(int k, int i) const { return (w.m[k] >> i) & 1u; }
    TD_HD bool ebit(int i) const { return (E >> i) & 1u; }
    TD_HD int run_end(int k, int from) const { return from >= 32 ? 32 : from + td_ctz32(~((w.m[k] & ~E) >> from)); }
    TD_HD int run_end2(int k1, int k2, int from) const {
        return from >= 32 ? 32 : from + td_ctz32(~(((w.m[k1] | w.m[k2]) & ~E) >> from));
    }
    TD_HD static uint32_t below(int n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }
    TD_HD int last_and(int k1, int k2, int lo, int hi) const {
        const uint32_t m = w.m[k1] & w.m[k2] & below(hi) & ~below(lo);
        return m ? 31 - (int)__builtin_clz(m) : -1;
    }
    TD_HD int last_set(int k, int lo, int hi) const {
        const uint32_t m = w.m[k] & below(hi) & ~below(lo);
        return m ? 31 - (int)__builtin_clz(m) : -1;
    }
    TD_HD int last_clear(int k, int lo, int hi) const {
        const uint32_t m = ~w.m[k] & below(hi) & ~below(lo);
        return m ? 31 - (int)__builtin_clz(m) : -1;
    }
};

// The common pieces of the o200k / Llama-4 and tekken patterns WITHOUT divergent control flow: every alternative is a few
// lines of mask arithmetic on the 32-bit view (bit 0 = the piece start), the result is selected at the end.  Returns the
// piece end, or -1 for everything it does not cover (runs that reach `avail`, a first character that is both a prefix and a
// letter class, multi-byte digits, the other patterns): the caller then runs scan_piece_p, which is the definition.  In a
// wavefront every lane sits in a different alternative, so the branchy matcher costs the SUM of all paths per piece.
// `b1`, `b2`: text bytes at the two positions behind a candidate apostrophe are fetched by `bytes(i)` only there.
template <class B>
TD_HD int scan_piece_fast32(const BitWin32& v, int avail, const B& bytes, uint32_t pv) {
    if (pv & (PV_GPT2 | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS | PV_WS_EOS_FIRST)) return -1;
    const uint32_t U = v.m[MK_U], W = v.m[MK_W], X = v.m[MK_X], S = v.m[MK_S], N = v.m[MK_N], CR = v.m[MK_CR], TR = v.m[MK_TR];
    const uint32_t C = v.m[MK_C], A = v.m[MK_A], SP = v.m[MK_SP];
    const uint32_t E = v.m[MK_D] & ~1u;                    // end of subject behind the start
    const uint32_t lim = avail >= 32 ? 0u : ~0u << avail;  // unknown positions
    const uint32_t Ue = U & ~E, We = W & ~E, Xe = X & ~E, Se = S & ~E, TRe = TR & ~E;
    const int p1 = 1 + td_ctz32(~(C >> 1));               // end of the first character
    const bool u0 = U & 1u, w0 = W & 1u, x0 = X & 1u, s0 = S & 1u, n0 = N & 1u, cr0 = CR & 1u;
    bool bad = p1 >= avail || p1 > 4;
    const int p1c = p1 > 31 ? 31 : p1;
    // ---- letters: [prefix] U* W+ | [prefix] U+ W*, then the contraction ----
    const bool l0 = u0 || w0;
    const bool l1 = (x0 || s0) && !((E >> p1c) & 1u) && (((U | W) >> p1c) & 1u);
    const bool letters = !cr0 && !n0 && (l0 || l1);
    bad = bad || (letters && l0 && l1);
    const int st = l1 ? p1c : 0;
    const int q = st + td_ctz32(~(Ue >> st));             // end of the U run (bits shifted in from the top are 0: a run never passes 32)
    const int qc = q > 31 ? 31 : q;
    const bool wf = ((We >> qc) & 1u) && q < 32;
    const int ew = wf ? qc + td_ctz32(~(We >> qc)) : q;
    const uint32_t uw = U & W & ((qc == 0 ? 0u : (~0u >> (32 - qc)))) & (~0u << st);  // letters of both classes in [st, q)
    const int e1 = wf ? ew : (uw ? 32 - (int)__builtin_clz(uw) : 0);
    const int el = e1 ? e1 : (q > st ? ew : 0);
    int e_let = el;
    if (!(pv & PV_NO_CONTRACTION)) {
        // (?i:'s|'t|'re|'ve|'m|'ll|'d)? behind the letters
        const int ec = el > 29 ? 29 : el;
        const bool apo = ((A >> ec) & 1u) && !((v.m[MK_D] >> ec) & 1u) && !((v.m[MK_D] >> (ec + 1)) & 1u);
        if (letters && el > 0 && apo) {
            if (el + 3 > avail) bad = true;
            else {
                const uint32_t b1 = bytes(el + 1), b2 = bytes(el + 2);
                const bool d2 = (v.m[MK_D] >> (ec + 2)) & 1u;
                const uint32_t c1 = b1 | 0x20u, c2 = b2 | 0x20u;
                if (b1 < 0x80u && (c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd')) e_let = el + 2;
                else if (b1 < 0x80u && !d2 && b2 < 0x80u && ((c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e') || (c1 == 'l' && c2 == 'l'))) e_let = el + 3;
                else if (b1 == 0xC5u && !d2 && b2 == 0xBFu) e_let = el + 3;
            }
        }
    }
    bad = bad || (letters && (el == 0 || q >= avail || ew >= avail || el + 3 > avail));
    // ---- \p{N}{1,3} (tekken: one digit); ASCII digits only ----
    const int nmax = (pv & PV_SINGLE_DIGIT) ? 1 : 3;
    const int nrun = td_ctz32(~(N & ~E) | (1u << nmax));   // consecutive digits from the start, at most nmax
    const int e_num = nrun < 1 ? 1 : nrun;
    bad = bad || (n0 && ((C & 0xEu) || e_num + 1 > avail));
    // ----  ?[^\s\p{L}\p{N}]+[\r\n/]* ----
    const bool psp = (SP & 1u) && ((X >> 1) & 1u) && !((E >> 1) & 1u);
    const bool punct = psp || x0;
    const int pst = psp ? 1 : 0;
    const int px = pst + td_ctz32(~(Xe >> pst));
    const int pxc = px > 31 ? 31 : px;
    const int e_p = px >= 32 ? 32 : pxc + td_ctz32(~(TRe >> pxc));
    bad = bad || (!letters && !n0 && punct && (px >= avail || e_p >= avail));
    // ---- \s*[\r\n]+ | \s+(?!\S) | \s+ ----
    const int sq = td_ctz32(~Se);
    const uint32_t below = sq >= 32 ? ~0u : ((1u << sq) - 1u);
    const uint32_t crs = CR & below;
    const uint32_t leads = ~C & below & ~1u;              // character starts inside the run, behind the first
    const int e_s = crs ? 32 - (int)__builtin_clz(crs)
                        : (sq < 32 && ((E >> sq) & 1u)) ? sq
                        : leads ? 31 - (int)__builtin_clz(leads) : sq;
    bad = bad || (!letters && !n0 && !punct && s0 && sq >= avail);
    (void)lim;
    const int e = letters ? e_let : n0 ? e_num : punct ? e_p : s0 ? e_s : p1;
    return (bad || e <= 0 || e > avail) ? -1 : e;
}

// ------------------------------------------------------------------ whole-word boundary rules ----
// Most regions between two consecutive synchronisation points ARE one piece (" word", ",", "2024"[:3], ".\n\n").  That can
// be proven for all heads of a 64-byte mask window at once with carry arithmetic: adding a start bit to a class mask runs
// through the run of ones it starts and lands on the first byte behind it (td_land).  A head is RESOLVED when the piece
// the pattern's alternatives give it lands exactly on the next synchronisation point; only the other heads (and what
// follows them up to the next synchronisation point) need the piece-by-piece matcher.  Every rule errs to "unresolved":
//   letters   [prefix char] pure-upper* lower-class+ | pure-upper+     (greedy U* W+ / U+ W* have this extent, see scan_letters)
//   digits    a run of at most 3 bytes (tekken: 1)
//   other     [space] non-letter punctuation+ [\r\n/]*
//   space     whitespace ending in \r or \n, or one whitespace byte
// Heads at document starts are left to the matcher (their masks are cut at F_DOC so that no run crosses a document).
TD_HD uint64_t td_land(uint64_t M, uint64_t A) { return ((M + (A & M)) & ~M) | (A & ~M); }
TD_HD uint64_t td_brev64(uint64_t x) {
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
constexpr uint32_t PV_WORD_RULES_OK = PV_NO_CONTRACTION | PV_SINGLE_DIGIT;  // patterns the rules below are written for
// -> the heads (MK_SYNC bits) of the window whose region is NOT proven to be exactly one piece.  A region that is not
// closed by a synchronisation point inside the window is unresolved.
// `extra` (optional): piece starts INSIDE regions that the rules can place as well — so far the second piece of a run of four to six
// ASCII digits (\p{N}{1,3} takes three, what is left is one more piece): half of the unresolved heads of the reference's code file set
// were such numbers.  The positions are piece starts whether or not their region counts as resolved.
TD_HD uint64_t split_unresolved_heads(const BitWin& w, uint32_t pv, uint64_t* extra = nullptr) {
    const uint64_t SY = w.m[MK_SYNC];
    if (extra) *extra = 0;
    if (pv & ~PV_WORD_RULES_OK) return SY;
    const uint64_t nD = ~w.m[MK_D];
    const uint64_t U = w.m[MK_U] & nD, W = w.m[MK_W] & nD, X = w.m[MK_X] & nD, S = w.m[MK_S] & nD, N = w.m[MK_N] & nD;
    const uint64_t CR = w.m[MK_CR] & nD, TR = w.m[MK_TR] & nD, C = w.m[MK_C] & nD, SP = w.m[MK_SP] & nD;
    const uint64_t H = SY & nD;
    const uint64_t L = U | W, Up = U & ~W, Xn = X & ~L;
    // letters
    const uint64_t pfx = H & (X | S) & ~CR & ~L;
    const uint64_t p1 = td_land(C, pfx << 1);  // end of the prefix character
    const uint64_t a1 = (H & L) | (p1 & L);
    uint64_t good = td_land(W, td_land(Up, a1));
    // digits
    const uint64_t hn = H & N;
    const uint64_t near = (pv & PV_SINGLE_DIGIT) ? (hn << 1) : ((hn << 1) | (hn << 2) | (hn << 3));
    const uint64_t ln = td_land(N, hn);
    good |= ln & near;
    if (extra && hn && !(pv & PV_SINGLE_DIGIT)) {
        // head + 3 when the head and the two bytes behind it are one-byte digits and head + 3 is a digit's first byte: the piece that
        // starts at the head is exactly those three, the next one starts there; it ends the region when what is left is at most three
        // bytes (the rule above, from the new start)
        const uint64_t NC = N & ~C;
        const uint64_t e3 = (hn << 3) & NC & (NC << 1) & (NC << 2);
        good |= ln & ((e3 << 1) | (e3 << 2) | (e3 << 3));
        *extra = e3;
    }
    // other
    const uint64_t b1 = (H & Xn) | ((H & SP & (Xn >> 1)) << 1);
    good |= td_land(TR, td_land(Xn, b1));
    // whitespace
    const uint64_t hs = H & S;
    good |= td_land(S, hs) & ((CR << 1) | (hs << 1));
    // regions whose end carries no landing: back to their heads (the same carry trick on the reversed window; the bit
    // shifted in closes the last region, whose end lies outside)
    const uint64_t bad_end = SY & ~good;
    const uint64_t zr = td_brev64(~SY);
    const uint64_t start = (td_brev64(bad_end) << 1) | 1ull;
    return td_brev64(td_land(zr, start));
}

// Provider over a word-major mask array (word i>>6 of mask k at arr[(i>>6)*MK_COUNT + k]): no limit on run
// length below `lim`.  Used for pieces / look-ahead that do not fit a 64-bit register window.
struct ArrMaskP {
    const uint64_t* arr;
    int o, lim;
    TD_HD ArrMaskP(const uint64_t* a, int o_, int lim_) : arr(a), o(o_), lim(lim_) {}
    TD_HD uint64_t word(int k, int wi) const { return arr[wi * MK_COUNT + k]; }
    TD_HD uint64_t eword(int wi) const {  // end-of-subject bits after o
        const int wo = o >> 6;
        if (wi < wo) return 0;
        const uint64_t d = word(MK_D, wi);
        return wi == wo ? d & ~td_bits_below((o & 63) + 1) : d;
    }
    TD_HD bool bit(int k, int i) const { return (word(k, i >> 6) >> (i & 63)) & 1ull; }
    TD_HD bool ebit(int i) const { return i > o && bit(MK_D, i); }
    TD_HD int run_end(int k, int from) const {
        int i = from;
        while (i < lim) {
            const int wi = i >> 6, sh = i & 63;
            const uint64_t m = (word(k, wi) & ~eword(wi)) >> sh;
            const int t = td_ctz64(~m);
            if (t < 64 - sh) return i + t;
            i += 64 - sh;
        }
        return i;
    }
    TD_HD int run_end2(int k1, int k2, int from) const {
        int i = from;
        while (i < lim) {
            const int wi = i >> 6, sh = i & 63;
            const uint64_t m = ((word(k1, wi) | word(k2, wi)) & ~eword(wi)) >> sh;
            const int t = td_ctz64(~m);
            if (t < 64 - sh) return i + t;
            i += 64 - sh;
        }
        return i;
    }
    template <class F>
    TD_HD int last_where(const F& f, int lo, int hi) const {  // highest i in [lo,hi) whose bit in f(word index) is set
        if (hi <= lo) return -1;
        for (int wi = (hi - 1) >> 6; wi >= (lo >> 6); --wi) {
            uint64_t m = f(wi);
            const int base = wi << 6;
            if (hi - base < 64) m &= td_bits_below(hi - base);
            if (lo > base) m &= ~td_bits_below(lo - base);
            if (m) return base + td_top64(m) - 1;
        }
        return -1;
    }
    TD_HD int last_and(int k1, int k2, int lo, int hi) const {
        return last_where([&](int wi) { return word(k1, wi) & word(k2, wi); }, lo, hi);
    }
    TD_HD int last_set(int k, int lo, int hi) const {
        return last_where([&](int wi) { return word(k, wi); }, lo, hi);
    }
    TD_HD int last_clear(int k, int lo, int hi) const {
        return last_where([&](int wi) { return ~word(k, wi); }, lo, hi);
    }
};

// (?i:'s|'t|'re|'ve|'m|'ll|'d)? on masks; `bytes(i)` returns the raw byte at position i.
template <class P, class B>
TD_HD int scan_contraction_p(const P& p, const B& bytes, int e, bool leading = false) {
    if (e + 3 > p.lim) return -1;
    if (!p.bit(MK_A, e) || (p.bit(MK_D, e) && !leading)) return e;
    if (p.bit(MK_D, e + 1)) return e;
    const uint32_t b1 = bytes(e + 1);
    const bool d2 = p.bit(MK_D, e + 2);
    if (b1 < 0x80) {
        const uint32_t l1 = b1 | 0x20;
        if (l1 == 's' || l1 == 't' || l1 == 'm' || l1 == 'd') return e + 2;
        if (d2) return e;
        const uint32_t b2 = bytes(e + 2);
        if (b2 < 0x80) {
            const uint32_t l2 = b2 | 0x20;
            if ((l1 == 'r' && l2 == 'e') || (l1 == 'v' && l2 == 'e') || (l1 == 'l' && l2 == 'l')) return e + 3;
        }
        return e;
    }
    if (b1 == 0xC5 && !d2 && bytes(e + 2) == 0xBF) return e + 3;
    return e;
}

// The GPT-2 pattern on the masks.
template <class P, class B>
TD_HD int scan_piece_gpt2_p(const P& p, const B& bytes) {
    const int o = p.o, lim = p.lim;
    if (o + 3 > lim) return -1;
    if (p.bit(MK_A, o) && !p.bit(MK_D, o + 1)) {  // 's|'t|'re|'ve|'m|'ll|'d  (case-sensitive)
        const uint32_t b1 = bytes(o + 1);
        if (b1 == 's' || b1 == 't' || b1 == 'm' || b1 == 'd') return o + 2;
        if (!p.bit(MK_D, o + 2)) {
            const uint32_t b2 = bytes(o + 2);
            if ((b1 == 'r' && b2 == 'e') || (b1 == 'v' && b2 == 'e') || (b1 == 'l' && b2 == 'l')) return o + 3;
        }
    }
    int st = o;
    if (p.bit(MK_SP, o) && !p.ebit(o + 1) &&
        (p.bit(MK_U, o + 1) || p.bit(MK_W, o + 1) || p.bit(MK_N, o + 1) || p.bit(MK_X, o + 1)))
        st = o + 1;
    int e = -1;
    if (p.bit(MK_U, st) || p.bit(MK_W, st)) e = p.run_end2(MK_U, MK_W, st);
    else if (p.bit(MK_N, st)) e = p.run_end(MK_N, st);
    else if (p.bit(MK_X, st)) e = p.run_end(MK_X, st);
    if (e >= 0) return e >= lim ? -1 : e;
    if (p.bit(MK_S, o)) {  // \s+(?!\S) | \s+
        const int q = p.run_end(MK_S, o);
        if (q >= lim) return -1;
        if (p.ebit(q)) return q;
        const int last_lead = p.last_clear(MK_C, o, q);
        if (last_lead > o) return last_lead;
        return q;
    }
    int p1 = o + 1;
    while (p1 < lim && p.bit(MK_C, p1)) ++p1;
    return p1 >= lim ? -1 : p1;
}

// End of the piece starting at p.o, in the provider's coordinates, or -1 (needs positions >= p.lim).
template <class P, class B>
TD_HD int scan_piece_p(const P& p, const B& bytes, uint32_t pv = 0) {
    if (pv & PV_GPT2) return scan_piece_gpt2_p(p, bytes);
    const int o = p.o, lim = p.lim;
    const bool u0 = p.bit(MK_U, o), w0 = p.bit(MK_W, o), x0 = p.bit(MK_X, o), s0 = p.bit(MK_S, o), n0 = p.bit(MK_N, o);
    const bool cr0 = p.bit(MK_CR, o);
    int p1 = o + 1;  // end of the first character
    while (p1 < lim && p.bit(MK_C, p1)) ++p1;
    if (p1 >= lim) return -1;
    if ((pv & PV_LEADING_CONTRACTION) && p.bit(MK_A, o)) {
        const int r = scan_contraction_p(p, bytes, o, true);
        if (r < 0) return -1;
        if (r != o) return r;
    }
    if (!cr0 && !n0) {
        const bool prefixable = x0 || s0;  // [^\r\n\p{L}\p{N}] = X or non-CR/LF whitespace
        const bool l0 = u0 || w0;
        const bool l1 = prefixable && !p.ebit(p1) && (p.bit(MK_U, p1) || p.bit(MK_W, p1));
        if ((pv & PV_PLAIN_LETTERS) && (l0 || l1)) {  // [^\r\n\p{L}\p{N}]?\p{L}+
            const int e = p.run_end2(MK_U, MK_W, l1 ? p1 : o);
            return e >= lim ? -1 : e;
        }
        if (l0 || l1) {
            // candidates in backtracking order: alt1 from p1, alt1 from o, alt2 from p1, alt2 from o
            int e1 = 0, e2 = 0;  // first successful alt-1 / alt-2 end (0 = none)
            for (int k = 0; k < 2; ++k) {
                const bool use = k == 0 ? l1 : l0;
                if (!use) continue;
                const int st = k == 0 ? p1 : o;
                const int q = p.run_end(MK_U, st);
                if (q >= lim) return -1;
                const bool wf = p.bit(MK_W, q) && !p.ebit(q);
                int ew = q;
                if (wf) {
                    ew = p.run_end(MK_W, q);
                    if (ew >= lim) return -1;
                }
                if (!e1) {
                    if (wf) e1 = ew;
                    else {
                        const int lw = p.last_and(MK_U, MK_W, st, q);
                        if (lw >= 0) e1 = lw + 1;
                    }
                }
                if (!e2 && q > st) e2 = ew;
                if (e1) break;  // alt 1 with the earlier candidate wins over everything that follows
            }
            const int e = e1 ? e1 : e2;
            if (e) return (pv & PV_NO_CONTRACTION) ? e : scan_contraction_p(p, bytes, e);
        }
    }
    if (n0) {  // \p{N}{1,3}  (tekken: \p{N})
        int e = p1;
        const int nmax = (pv & PV_SINGLE_DIGIT) ? 1 : 3;
        for (int k = 1; k < nmax; ++k) {
            if (e >= lim) return -1;
            if (!p.bit(MK_N, e) || p.ebit(e)) break;
            ++e;
            while (e < lim && p.bit(MK_C, e)) ++e;
            if (e >= lim) return -1;
        }
        return e;
    }
    {  //  ?[^\s\p{L}\p{N}]+[\r\n/]*
        int st = -1;
        if (p.bit(MK_SP, o) && p.bit(MK_X, o + 1) && !p.ebit(o + 1)) st = o + 1;
        else if (x0) st = o;
        if (st >= 0) {
            int e = p.run_end(MK_X, st);
            if (e >= lim) return -1;
            e = p.run_end(MK_TR, e);
            if (e >= lim) return -1;
            return e;
        }
    }
    if (s0) {  // \s*[\r\n]+ | \s+(?!\S) | \s+
        const int q = p.run_end(MK_S, o);
        if (q >= lim) return -1;
        if ((pv & PV_WS_EOS_FIRST) && p.ebit(q)) return q;
        const int lc = p.last_set(MK_CR, o, q);
        if (lc >= 0) return lc + 1;
        if (p.ebit(q)) return q;
        const int last_lead = p.last_clear(MK_C, o, q);
        if (last_lead > o) return last_lead;
        return q;
    }
    return p1;
}

// register-window form (the fast path of td_split_tiles)
template <class B>
TD_HD int scan_piece_bits(const BitWin& w, const B& bytes, int o, int avail, uint32_t pv = 0) {
    const WinP p(w, o, avail);
    return scan_piece_p(p, bytes, pv);
}

// Feature byte: the class-set memberships of one byte, one bit each (what phase 1 of the kernel keeps per byte;
// a wavefront ballot per bit turns 64 of them into the mask words).  FB_N doubles as a tag: X|N = apostrophe,
// S|N = U+0020 (neither X nor S ever coincides with a real number class), so no second LDS read of the text is
// needed for those two masks.
enum : uint32_t { FB_U = 1, FB_W = 2, FB_X = 4, FB_S = 8, FB_N = 16, FB_CR = 32, FB_SL = 64, FB_C = 128 };
TD_HD uint32_t feature_of_class(uint32_t c) {
    switch (c) {
        case C_OTHER: return FB_X;
        case C_APOS: return FB_X | FB_N;
        case C_SLASH: return FB_X | FB_SL;
        case C_SP: return FB_S | FB_N;
        case C_WS: return FB_S;
        case C_CRLF: return FB_S | FB_CR;
        case C_UP: return FB_U;
        case C_LW: return FB_W;
        case C_LB: return FB_U | FB_W;
        case C_MK: return FB_U | FB_W | FB_X;
        case C_NUM: return FB_N;
    }
    return FB_X;
}
TD_HD bool fb_is_num(uint32_t f) { return (f & (FB_N | FB_X | FB_S)) == FB_N; }
TD_HD bool fb_is_apos(uint32_t f) { return (f & (FB_N | FB_X)) == (FB_N | FB_X); }
TD_HD bool fb_is_sp(uint32_t f) { return (f & (FB_N | FB_S)) == (FB_N | FB_S); }
// 8x8 bit-matrix transpose: input = 8 bytes (byte j = row j), output byte k = column k (bit j of it = bit k of
// input byte j).  Turns 8 feature bytes into the 8 per-feature bit planes of those 8 text bytes.
TD_HD uint64_t transpose8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}
// 8-bit slice of the SYNC mask from the 8-bit slices of the class masks (same predicate as sync_word / is_sync);
// `pf` = feature byte of the byte in front of the slice.
TD_HD uint32_t sync_byte(uint32_t U, uint32_t W, uint32_t X, uint32_t S, uint32_t N, uint32_t CR, uint32_t SL,
                         uint32_t C, uint32_t D, uint32_t A, uint32_t pf, uint32_t pv = 0) {
    const uint32_t L = (U | W) & ~X;
    const uint32_t pS = (S << 1) | ((pf & FB_S) ? 1u : 0u);
    const uint32_t pCR = (CR << 1) | ((pf & FB_CR) ? 1u : 0u);
    const uint32_t pN = (N << 1) | (fb_is_num(pf) ? 1u : 0u);
    const uint32_t pL = (L << 1) | (((pf & (FB_U | FB_W)) && !(pf & FB_X)) ? 1u : 0u);
    const uint32_t num = (pv & PV_GPT2) ? (~N & pN) : (N ^ pN);
    const uint32_t sy = (S & ~CR & ~pS) | (pCR & ~S & ~SL) | num | (X & ~(U | W) & ~A & pL);
    return ((sy & ~C) | D) & 0xFFu;
}

// SYNC mask word from the class mask words of the same 64 bytes; `pf` = feature byte of the byte just
// before the word (0 if none).  Bit-for-bit the same predicate as is_sync().
TD_HD uint64_t sync_word(uint64_t U, uint64_t W, uint64_t X, uint64_t S, uint64_t N, uint64_t CR, uint64_t SL,
                         uint64_t C, uint64_t D, uint64_t A, uint32_t pf, uint32_t pv = 0) {
    const uint64_t L = (U | W) & ~X;
    const uint64_t pS = (S << 1) | ((pf & FB_S) ? 1ull : 0ull);
    const uint64_t pCR = (CR << 1) | ((pf & FB_CR) ? 1ull : 0ull);
    const uint64_t pN = (N << 1) | (fb_is_num(pf) ? 1ull : 0ull);
    const uint64_t pL = (L << 1) | (((pf & (FB_U | FB_W)) && !(pf & FB_X)) ? 1ull : 0ull);
    const uint64_t num = (pv & PV_GPT2) ? (~N & pN) : (N ^ pN);
    uint64_t sy = (S & ~CR & ~pS) | (pCR & ~S & ~SL) | num | (X & ~(U | W) & ~A & pL);
    return (sy & ~C) | D;
}

// ------------------------------------------------------------------ lane-per-piece merge ----
// The byte-pair merge of one piece (bpe_merge, tiktoken.cpp:298-368) as ONE LANE runs it; 64 pieces per wavefront advance
// together, one merge each per round.  A piece of `len` <= 64 bytes owns ceil(len/16) consecutive 16-slot units of two
// arrays (LDS on the device): ids[] = the id of the part that STARTS at byte j (stale for absorbed parts), keys[] =
// rank(part j, next part) << 6 | j, or MG_DEAD when that pair is no token or j is not a part start.  A round: minimum
// over the piece's keys (lowest rank, leftmost on ties = the reference's strict '<' scan, tiktoken.cpp:334-342), the
// right part is absorbed, the two pairs that touch the merged part are looked up again (tiktoken.cpp:324-331).
// Slots are XOR-swizzled by unit so that 16-byte reads of different lanes fall into different LDS banks.
constexpr uint32_t MG_DEAD = 0xFFFFFFFFu;
constexpr int MG_UNIT = 16;
TD_HD uint32_t mg_slot(uint32_t t, uint32_t j) { return (t << 4) + (j ^ (((t >> 2) & 3u) << 2)); }  // (bits 2..3 of j swizzled by the owner lane)
struct MergeState {
    uint64_t alive;  // bit j: a part starts at byte j
    uint32_t t;      // first unit of the piece
    uint32_t len;    // bytes; 0 = this lane has no piece
};
struct U4 { uint32_t x, y, z, w; };
// part j of the piece: byte b, next byte bn (ignored for the last part)
TD_HD void mg_put(const Tables& T, const int32_t* byte_id, uint32_t* keys, uint32_t* ids, const MergeState& st, uint32_t j, uint32_t b,
                  uint32_t bn) {
    const uint32_t sl = mg_slot(st.t, j);
    ids[sl] = (uint32_t)byte_id[b];
    uint32_t key = MG_DEAD;
    if (j + 1 < st.len) {
        const int32_t r = T.byte_pair[(b << 8) | bn];
        if (r != NO_RANK) key = ((uint32_t)r << 6) | j;
    }
    keys[sl] = key;
}
TD_HD void mg_pad(uint32_t* keys, const MergeState& st) {  // key slots behind the last part of the last unit
    const uint32_t end = ((st.len + 15u) >> 4) << 4;
    for (uint32_t j = st.len; j < end; ++j) keys[mg_slot(st.t, j)] = MG_DEAD;
}
TD_HD uint32_t td_min3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t ab = a < b ? a : b;
    return ab < c ? ab : c;
}
// one merge; false when the piece has no mergeable pair left.  MaskT = uint32_t for pieces of at most 32 bytes (the part
// mask fits one register), uint64_t otherwise.
template <class MaskT>
TD_HD bool mg_round_t(const Tables& T, uint32_t* keys, uint32_t* ids, MergeState& st) {
    const uint32_t units = (st.len + 15u) >> 4;
    const uint32_t base = st.t << 4, sx = ((st.t >> 2) & 3u) << 2;  // slot of part j = base + (j ^ sx)
    uint32_t m = MG_DEAD;
    for (uint32_t c = 0; c < units; ++c) {
        // the unit's sixteen keys: four 16-byte LDS reads issued back to back, THEN the minimum (the compiler, left alone, reused
        // one register quadruple and waited for every read before the next: four LDS round trips in a row per unit and round)
        U4 k0 = *reinterpret_cast<const U4*>(keys + base + ((16u * c + 0u) ^ sx));
        U4 k1 = *reinterpret_cast<const U4*>(keys + base + ((16u * c + 4u) ^ sx));
        U4 k2 = *reinterpret_cast<const U4*>(keys + base + ((16u * c + 8u) ^ sx));
        U4 k3 = *reinterpret_cast<const U4*>(keys + base + ((16u * c + 12u) ^ sx));
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(k0.x), "+v"(k1.x), "+v"(k2.x), "+v"(k3.x));
#endif
        m = td_min3(td_min3(k0.x, k0.y, k0.z), k0.w, m);
        m = td_min3(td_min3(k1.x, k1.y, k1.z), k1.w, m);
        m = td_min3(td_min3(k2.x, k2.y, k2.z), k2.w, m);
        m = td_min3(td_min3(k3.x, k3.y, k3.z), k3.w, m);
    }
    if (m == MG_DEAD) return false;
    const uint32_t w = m & 63u, r = m >> 6;
    constexpr uint32_t NB = sizeof(MaskT) * 8;
    const MaskT one = 1;
    MaskT alive = (MaskT)st.alive;
    MaskT above = alive & ~(((one << w) << 1) - one);    // part starts behind w (there is one: the pair at w has a right part)
    const uint32_t nx = sizeof(MaskT) == 8 ? (uint32_t)td_ctz64((uint64_t)above) : (uint32_t)td_ctz32((uint32_t)above);
    above &= above - one;
    const uint32_t nn = above ? (sizeof(MaskT) == 8 ? (uint32_t)td_ctz64((uint64_t)above) : (uint32_t)td_ctz32((uint32_t)above)) : 64u;
    const MaskT below = alive & ((one << w) - one);
    const uint32_t pw = below ? (sizeof(MaskT) == 8 ? (uint32_t)(td_top64((uint64_t)below) - 1) : 31u - (uint32_t)__builtin_clz((uint32_t)below)) : 64u;
    (void)NB;
    alive &= ~(one << nx);
    st.alive = (uint64_t)alive;
    const uint32_t id_nn = nn < 64u ? ids[base + (nn ^ sx)] : 0u;
    const uint32_t id_pw = pw < 64u ? ids[base + (pw ^ sx)] : 0u;
    ids[base + (w ^ sx)] = r;  // a merged part's id is its rank
    // both pair lookups at once: four independent 8-byte probes in flight.  A neighbour that does not exist is not looked
    // up (the loads sit under the lanes' execution mask: every lane of a divergent load is a separate request to the
    // vector L1, and their number is what bounds td_merge_pieces)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint64_t __attribute__((address_space(1)))* gpair_t;  // (global loads, not flat ones)
    gpair_t const ps = (gpair_t)(uintptr_t)T.pair_slots;
    uint64_t e1 = PAIR_EMPTY, e2 = PAIR_EMPTY, e3 = PAIR_EMPTY, e4 = PAIR_EMPTY;
#ifdef TD_MG_FOUR_PROBES  // (rounds 2-4: both seats of both pairs in flight at once)
    if (nn < 64u) { e1 = ps[hash_pair(r, id_nn) & T.pair_mask]; e2 = ps[hash_pair2(r, id_nn) & T.pair_mask]; }
    if (pw < 64u) { e3 = ps[hash_pair(id_pw, r) & T.pair_mask]; e4 = ps[hash_pair2(id_pw, r) & T.pair_mask]; }
    asm volatile("" : "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4));  // all four are in flight before the first one is looked at
#else
    // the first seats of both pairs; a second seat only where the first one neither holds the pair nor says it is final
    if (nn < 64u) e1 = ps[hash_pair(r, id_nn) & T.pair_mask];
    if (pw < 64u) e3 = ps[hash_pair(id_pw, r) & T.pair_mask];
    asm volatile("" : "+v"(e1), "+v"(e3));
    if (nn < 64u && !(e1 & PAIR_FINAL) && !pair_slot_is(e1, r, id_nn)) e2 = ps[hash_pair2(r, id_nn) & T.pair_mask];
    if (pw < 64u && !(e3 & PAIR_FINAL) && !pair_slot_is(e3, id_pw, r)) e4 = ps[hash_pair2(id_pw, r) & T.pair_mask];
#endif
#else
    const uint64_t e1 = T.pair_slots[hash_pair(r, id_nn) & T.pair_mask], e2 = T.pair_slots[hash_pair2(r, id_nn) & T.pair_mask];
    const uint64_t e3 = T.pair_slots[hash_pair(id_pw, r) & T.pair_mask], e4 = T.pair_slots[hash_pair2(id_pw, r) & T.pair_mask];
#endif
    const int32_t r1 = pair_match(e1, e2, r, id_nn), r2 = pair_match(e3, e4, id_pw, r);
    const uint32_t kw = (nn < 64u && r1 != NO_RANK) ? (((uint32_t)r1 << 6) | w) : MG_DEAD;
    const uint32_t kp = (pw < 64u && r2 != NO_RANK) ? (((uint32_t)r2 << 6) | pw) : MG_DEAD;
    keys[base + (w ^ sx)] = kw;
    keys[base + (nx ^ sx)] = MG_DEAD;
    if (pw < 64u) keys[base + (pw ^ sx)] = kp;
    return true;
}
TD_HD bool mg_round(const Tables& T, uint32_t* keys, uint32_t* ids, MergeState& st) { return mg_round_t<uint64_t>(T, keys, ids, st); }

// ------------------------------------------------------------------ tile geometry -----------
// One workgroup of td_encode_tiles handles one tile of text at a time.
#ifndef TD_K_THREADS
#define TD_K_THREADS 256
#endif
constexpr int K_THREADS = TD_K_THREADS;        // 4 wavefronts
constexpr int K_CHUNK = 16;                    // text bytes whose boundaries one lane is responsible for
constexpr int K_TILE = K_THREADS * K_CHUNK;    // 4096 text bytes per tile
#ifndef TD_K_HL
#define TD_K_HL 128
#endif
constexpr int K_HL = TD_K_HL;                  // left halo (sync-point back-search)
constexpr int K_HR = 192;                      // right halo (piece overrun / look-ahead)
// The pre-tokenizer (td_split_tiles and the CPU twin's tile loop) has its own geometry: a lane applies the whole-word
// rules to a stride of KS_CHUNK = 32 bytes (one 32-bit word of every class mask, + 32 bytes of look-ahead).
#ifndef TD_KS_CHUNK
#define TD_KS_CHUNK 32
#endif
constexpr int KS_CHUNK = TD_KS_CHUNK;
constexpr int KS_TILE = K_THREADS * KS_CHUNK;  // text bytes per pre-tokenizer tile
constexpr int K_WIN = K_HL + KS_TILE + K_HR;   // bytes the pre-tokenizer stages in LDS per tile
constexpr int K_LIM = K_WIN - 4;               // the scanner may read window positions < K_LIM
constexpr int K_MAXSHORT = 64;                 // pieces up to this many bytes merge inside one wavefront
constexpr int K_STAGE = K_TILE + K_MAXSHORT;   // staging slots per tile: a tile owns the tokens of the pieces that START in it,
                                               // and its last piece may end up to K_MAXSHORT - 1 bytes into the next tile

// One head's share of the boundary scan of a tile (phase 2 (b) of td_split_tiles; the CPU twin runs the same rules head
// by head).  Window coordinates: index i <-> global byte wg0 + i; the tile owns [K_HL, tile_hi).  W: window accessor
// (pos_t = int; cf, byte, lim) plus mark(i) (set F_START) and set_ext(i, global_end) (the one piece that leaves the
// window).  G: accessor over the whole text in HBM (pos_t = int64_t; cf, byte, lim, scan(pos)) for what the window
// cannot answer (on the device: the td_split_far_* kernels).
//   heads = the synchronisation points of the tile the whole-word rules (split_unresolved_heads) do not resolve,
//           the last synchronisation point before the tile start (its pieces lead into the tile) and the tile's last head
//           (its pieces delimit the tile's last piece);
//   a head's pieces are matched one after the other until a piece ends on a synchronisation point or leaves the tile.
template <class W, class G>
TD_HD void scan_chain(W& w, const G& g, int head, int tile_hi, int64_t wg0, uint32_t pv = 0) {
    int p = head;  // a piece start; marked by the caller when it lies in the tile
    for (;;) {
        int e = scan_piece(w, p, pv);
        if (e < 0) {
            w.note_far(p);  // (the device hands this piece start to td_split_far_pieces)
            const int64_t ge = g.scan(wg0 + p);
            if (ge - wg0 > (int64_t)K_LIM) {  // piece leaves the window: one per tile at most
                if (p >= K_HL) w.set_ext(p, ge);
                return;
            }
            e = (int)(ge - wg0);
        }
        if (e >= tile_hi) { w.mark(e); return; }  // delimits the last owned piece
        if (is_sync(w.cf(e - 1), w.cf(e), pv)) return;
        if (e >= K_HL) w.mark(e);
        p = e;
    }
}

}  // namespace td
