// Host-side construction of the tokenizer tables that live in HBM (see td_common.h for layouts).
// Replaces what the reference builds in CoreBPE::CoreBPE (tiktoken.hpp:48-67): four emhash8 maps
// + a JIT-compiled PCRE2 pattern become: Unicode class tables, a piece-bytes -> rank table, an
// (id,id) -> rank pair table, byte/byte-pair direct tables and a rank -> bytes store.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "td_common.h"

namespace td {

enum PatternKind : int {
    PATTERN_UNSUPPORTED = -1,
    PATTERN_O200K = 0,  // the Llama-4 / o200k_base split pattern (reference src/main.cpp:114)
    PATTERN_TEKKEN = 1, // Mistral tekken.json config.pattern (reference tests/throughput_test.py:118)
    PATTERN_CL100K = 2, // cl100k_base / Llama-3 (tiktoken's pat_str; any pattern is legal input to the reference: wrapper.py:39)
    PATTERN_GPT2 = 3,   // r50k_base / p50k_base (GPT-2)
    PATTERN_CL100K_EOS = 4,  // cl100k_base as current tiktoken releases spell it: `\s++$` ahead of `\s*[\r\n]` (a different language
                             // on trailing whitespace that contains CR/LF)
    PATTERN_GENERIC = 6,  // any other pattern td_regex.cpp can compile: matched by the generic engine, td_regex.h
    PATTERN_QWEN2 = 5,  // Qwen2 / Qwen2.5 / Qwen3 (tokenizer.json pre_tokenizer): the cl100k_base pattern with single-digit number pieces (`\p{N}`)
};
const char* cl100k_pattern();
uint32_t pattern_flags(PatternKind k);  // PV_* bits for the scanners
const char* tekken_pattern();
PatternKind classify_pattern(const std::string& pat);
const char* o200k_pattern();

struct HostTables {
    PatternKind pattern_kind = PATTERN_UNSUPPORTED;
    std::string pattern;
    std::vector<uint8_t> rx_program;  // PATTERN_GENERIC: the compiled pattern (one RxProgram, td_regex.h)
    bool rx_left_context = false;     // ... uses ^ \\A \\b \\B: a match depends on what stands in front of its subject
    std::vector<uint8_t> ascii_cls;
    std::vector<uint8_t> ucls2_remap;  // stage-2 class table with the pattern's class remaps applied (empty: the static one)
    std::vector<int32_t> byte_id;
    std::vector<int32_t> byte_pair;
    std::vector<uint64_t> byte_pair_id;
    std::vector<PieceSlot> piece_slots;
    std::vector<Piece12Slot> piece12_slots;
    uint32_t piece12_mask = 0;
    std::vector<uint64_t> pair_slots;
    std::vector<uint32_t> tok_off;
    std::vector<uint8_t> tok_bytes;
    uint32_t piece_mask = 0, pair_mask = 0;
    int32_t max_id = -1;        // over regular AND special ids (decode range)
    int32_t max_rank = -1;      // over regular ids
    int32_t pseudo_base = 0;
    uint32_t max_token_len = 0;
    uint64_t n_pairs = 0;
    uint64_t n_pairs_second_seat = 0;  // pairs that sit in their second seat (hash_pair2): their first seats are not PAIR_FINAL
    bool merge_closed = false;  // every multi-byte token is what the merge loop produces from its own bytes
    // character seeds (td_common.h: cseed_char_at): empty = none
    std::vector<uint64_t> cseed;      // [65536]
    std::vector<uint32_t> cseed_pm, cseed_nm;  // [rows][8]
    uint32_t n_char_seeds = 0;        // characters that may be entered whole (statistics)
    std::vector<std::string> special_strs;
    std::vector<int32_t> special_ids;

    Tables view() const;  // Tables whose pointers are the host vectors (CPU twin / table self-check)
};

// Returns TD_OK or a TD_E_* code with a message in err.
int build_tables(const char* pattern, int64_t n_vocab, const uint8_t* token_bytes, const int64_t* token_offsets,
                 const int32_t* ranks, int64_t n_special, const uint8_t* special_bytes,
                 const int64_t* special_offsets, const int32_t* special_ranks, HostTables& out, std::string& err);

// Host reference of the device merge (ids + pair table): appends ids for piece[0,n) to out.
// Returns TD_OK or TD_E_UNKNOWN_BYTE.  Used for the merge_closed self-check and by the CPU twin.
int merge_piece_host(const Tables& T, const uint8_t* piece, uint32_t n, std::vector<int32_t>& out);

// 64-bit key of a piece as the tables store it.
uint64_t piece_key_host(const uint8_t* p, uint32_t len);

}  // namespace td
