// Launch interface between the host library (td_api.cpp) and the gfx950 kernels (td_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "td_common.h"

namespace td {

struct RxProgram;       // td_regex.h: a compiled generic split pattern

constexpr int TD_GP_SCRATCH_BYTES = 8192;  // td_giant_pieces: (1024 listed pieces + 2 x 256 x 2 slots) x 4
struct LongEntry {      // a piece longer than K_MAXSHORT bytes, merged by td_long_pieces
    int64_t gs;         // global byte offset of the piece
    uint32_t len;       // bytes
    uint32_t ntok;      // out: number of tokens
    uint64_t pool_off;  // out: offset (in u32) of its tokens inside the pool
};

// The allowed special literals of a call, sorted bytewise (td_special.hip), and the two bitmaps of the cut search.
struct SpecialTable {
    const uint8_t* bytes;     // the literals, each dword-aligned and zero-padded to a multiple of 4 bytes
    const uint32_t* off;      // [n] byte offset of literal i (a multiple of 4)
    const uint32_t* len;      // [n] its length in bytes
    const int32_t* id;        // [n]
    const int32_t* parent;    // [n] the longest other literal that is a proper prefix of literal i, or -1
    const uint32_t* first2;   // [2048] bit (b0 << 8 | b1): a literal starts with these two bytes
    int64_t* cand_pos;        // [cand_cap] positions whose first two bytes start an allowed literal
    int32_t* cand_lit;        // [cand_cap] the literal matched there, or -1
    uint32_t* cand_count;
    uint32_t cand_cap;
    uint32_t* hitbits;        // [(n_text+31)/32] an allowed literal matches at this byte
    uint32_t* accbits;        // ... and is cut out (not inside a literal cut out in front of it)
    uint32_t n, maxlen;       // literals; bytes of the longest one.  n == 0: no cuts in this call
};

// Generic split patterns: text bytes per lane of td_generic_chunks.  A lane needs about 3 microseconds per byte (a
// backtracking matcher, 64 lanes in 64 different states), so a chunk sets the floor of the kernel's duration: 1 KiB chunks
// took 3 ms however small the text was (8 MiB: 2.6 GB/s; with 256-byte chunks 8.7).  Small enough for two chunks per lane
// the GPU can hold (256 CUs x 8 waves x 64 lanes), not smaller than 64 bytes (a chunk's first piece or two are speculation
// and matched twice), not larger than 1 KiB; a multiple of 32: a lane owns whole words of the bitmaps.
inline int gx_chunk_for(int64_t n) {
#ifdef TD_GX_CHUNK
    (void)n;
    return TD_GX_CHUNK;  // (A/B builds)
#else
    int c = 64;
    while (c < 1024 && n / (2 * c) >= 262144) c *= 2;
    return c;
#endif
}

struct EncodeArgs {
    const Tables* Tp;           // the table descriptor lives in device memory (keeps the kernel argument block small
                                // and lets out-of-line helpers take a pointer without spilling kernargs to scratch)
    const uint8_t* text;        // [n] UTF-8, all documents concatenated
    int64_t n;
    const int64_t* doc_offsets; // [n_docs+1], doc_offsets[0]==0, doc_offsets[n_docs]==n
    int64_t n_docs;
    uint32_t* docbits;          // [(n+31)/32+1] bit i set <=> a document starts at byte i
    uint32_t* startbits;        // [(n+31)/32+8] bit i set <=> a regex piece starts at byte i (td_split_tiles -> td_probe_tiles)
    int64_t* slow_list;         // [slow_cap] piece starts whose end td_split_tiles could not see (td_split_far_pieces)
    uint32_t* tile_flag;        // [n_stiles+1] pre-tokenizer tile has no sync point in its left halo (td_split_far_tiles)
    int64_t* tile_carry;        // [n_stiles+1] first piece start at/after the tile start, where a scan from the left knows it
    uint32_t slow_cap;
    uint32_t* slow_count;
    uint32_t* stage;            // [n_tiles*K_STAGE] per-tile slots, one per piece (td_probe_tiles): id | TOK_MISS.. | TOK_LONGREF..
    uint32_t* merge_out;        // [n_tiles*K_STAGE] ids of the merged pieces, at tile * K_STAGE + the piece's tile position (td_merge_pieces)
    uint32_t* tile_count;       // [n_tiles] slots in the tile | TILE_HAS_LONG | TILE_HAS_MISS
    uint32_t* tile_extra;       // [n_tiles] sum(ntok-1) over the tile's long pieces
    int64_t* tile_base;         // [n_tiles+1] exclusive scan of count+extra
    uint32_t* doc_slot;         // [n_docs] slot index (inside its tile) of each document's first token
    uint32_t* tile_first_doc;   // [n_tiles] index of the first document starting in the tile (0xFFFFFFFF: none)
    LongEntry* long_list;
    uint32_t long_cap;
    uint32_t* long_count;
    uint32_t* giant_count;      // entries of long_list above 1 KiB (td_giant_pieces)
    uint32_t* gp_ctl;           // td_giant_pieces over all workgroups (round 5): [0] grid-barrier arrivals, [1] pieces listed, [2] a barrier gave up, [3] pieces above gp_coop_min (counted by the lookups) — zero at launch
    uint32_t* gp_scratch;       // [TD_GP_SCRATCH_BYTES / 4] the listed pieces | two slots per workgroup and parity
    uint32_t gp_coop_min;       // pieces above this many bytes are swept by all workgroups of the launch together
    uint32_t* tile_draw;        // fused tile loop: counter the workgroups draw their tiles from (0 at launch)
    uint32_t* far_count;        // pre-tokenizer tiles flagged in tile_flag (td_split_far_tiles looks for chains only when there is one; 0 at launch)
    uint32_t* ph_bar;           // td_far_probe / td_tail: arrivals at their grid barriers (0 at launch)
    int pack_dense;             // the dense sequence: td_pack_dense takes the tiles with merged / long pieces (<= 1024 slots), td_pack_rest what is left (TD_PACK_DENSE=0: off, A/B)
    int far_light;              // td_far_probe / td_tail: the far phases' barriers without cache maintenance (TD_FAR_LIGHT=0: with, A/B)
    uint32_t* lp_next;          // td_long_pieces: chunks of 64 list entries handed out beyond every wavefront's first (0 at launch)
    uint32_t* gs_done;          // td_giant_scan: workgroups that have left the giant pieces (0 at launch)
    int sparse;                 // the sparse launch sequence (td_tail + td_giant_scan instead of nine kernels): launch_encode
    uint32_t* pool;             // long-piece scratch + token store
    uint64_t pool_cap;          // in u32
    unsigned long long* pool_used;
    uint32_t* scan_done;        // chunks of td_scan_tiles finished (the last one scans the chunk totals)
    uint32_t* merge_next;       // td_merge_pieces: next tile nobody has taken yet (wavefronts draw runs of tiles)
    unsigned long long* miss_list;  // missed pieces of tiles that have only a few (tile << 32 | slot << 19 | tile position << 7 | length),
                                    // one list per length class, miss_cap entries apart
    uint32_t* miss_count;       // [K_MISS_CLASSES] entries on them
    uint32_t miss_cap;          // (room for K_MISS_LISTED_MAX per tile on every list)
    // ... and of the flagged tiles (many missed pieces): td_collect_misses puts them on COLL_SUBS more lists per length class, in the
    // same buffer behind the five above (list (c, s) starts coll_base[c] + s * coll_cap[c] records into miss_list), so that
    // td_merge_pieces only ever takes full rows from lists.  coll_count[(c * COLL_SUBS + s) * COLL_STRIDE] = entries on list (c, s).
    uint32_t* coll_count;
    uint32_t coll_cap[K_MISS_CLASSES];
    unsigned long long coll_base[K_MISS_CLASSES];
    uint16_t* rest_mask;        // [(n_tiles + 15) / 16] bit k of word g: tile 16 g + k is td_pack_rest's whatever its base (td_scan_tiles -> td_pack_rest)
    uint32_t* ovf_count;        // tiles with a class whose records found no room (tile_count bits TILE_OVF_SHIFT..: td_merge_pieces scans those)
    // round 5: a missed piece whose bytes another missed piece of the call has is not merged again.  dd_table: open addressing, a slot
    // = the record (tile << 32 | slot << 19 | position << 7 | length) of the FIRST piece with those bytes to arrive | hash tag << 56;
    // zeroed by td_prepare; the bytes are compared, the tag only saves most of the comparisons.  0 entries: off.
    unsigned long long* dd_table;
    uint32_t dd_mask;           // entries - 1 (a power of two)
    uint32_t dd_seat_bits;      // bits of a seat number in an entry of dup_list (>= log2(entries); the tile number gets the other 39 - this)
    int dedupe;
    uint32_t dd_minlen;         // pieces below this many bytes are not looked up (merged themselves)
    uint32_t* dd_stats;         // [2] (statistics) repeats / pieces listed for the merge by td_collect_misses, written by td_copy_dups
    uint32_t dd_replicas;       // seats a piece may take, one per group of workgroups (a power of two)
    int overlap;                // (host only: the long pieces run beside the short ones in this call — part of the key a captured graph is reused by)
    unsigned long long* dup_list;  // the repeats: COLL_SUBS lists of dup_cap entries (tile | slot (13 bits) | tile position (12) | seat of dd_table that
    uint32_t dup_cap;              // names the piece whose ids it gets); coll_count[(K_MISS_CLASSES * COLL_SUBS + s) * COLL_STRIDE] = entries on list s (td_copy_dups)
    // generic split patterns (PV_GENERIC; td_generic.hip)
    const RxProgram* rx;        // the compiled pattern
    const uint16_t* rx_stage1;  // general-category table (generated/unicode_gc.inc)
    const uint8_t* rx_stage2;
    int64_t* gap_list;          // (as uint32) the chunks that failed td_generic_commit's check
    uint32_t* gap_count;
    uint32_t gap_cap;
    uint32_t* gapbits;          // [(n+31)/32+8] bit i set <=> a stretch of text the pattern skips starts at byte i (it gets no tokens)
    int32_t gx_chunk;           // bytes per chunk (gx_chunk_for(n))
    int64_t* gx_exit;           // [n / gx_chunk + 1] per chunk: the first piece start at or behind the chunk end, as its own run found it
    uint32_t* gx_state;         // [n / gx_chunk + 1] per chunk: 1 = its run agrees with the chunk in front of it
    const uint8_t* gx_prefix;   // [n_docs] or null: the first gx_prefix[d] bytes of document d are LEFT CONTEXT only (the last character of
                                // the special token a segment stands behind, tiktoken.cpp:86-93): matched from behind them, no tokens
    uint32_t* deferred_list;    // fused tile loop (td_split_tiles<.., true>): the token tiles it left to td_probe_tiles
    uint32_t* deferred_count;   // entries on it
    int fused;                  // launch the fused tile loop (pre-tokenizer + lookup in one pass over the text)
    int probe_deferred;         // td_probe_tiles: only the tiles on deferred_list (set by launch_encode)
    SpecialTable sp;            // allowed special tokens to cut out of the text (sp.n == 0: none)
    uint32_t* flagged_list;     // the tiles td_probe_tiles flagged TILE_HAS_MISS, in the order its workgroups appended them
    uint32_t* flagged_count;    // entries on it
    int64_t* chunk_pref;        // [n_tiles/4096 + 2] token base of every 4096-tile chunk (exclusive scan of the chunk totals)
    uint32_t* ctl_reset; uint32_t ctl_reset_words;  // per-call counters td_prepare clears
    int32_t* out_tokens;        // [out_cap]
    int64_t out_cap;
    int64_t* out_offsets;       // [n_docs+1] token offset of each document; [n_docs] = total
    int* err;                   // err[0] = first TD_E_* raised on device (0 = ok)
    long long* err_pos;         // byte offset it refers to
    int n_tiles;                // token-kernel tiles (K_TILE bytes)
    int n_stiles;               // pre-tokenizer tiles (KS_TILE bytes)
    uint32_t pat_flags;         // PV_* scanner flags of the split pattern (selects the td_split_tiles instantiation)
    int use_fastpath;           // whole-piece lookup before the merge loop (CoreBPE::encode) or not
    int text_aligned;           // text pointer is 16-byte aligned
    int stop_after;             // ablation (only in -DTD_ABLATE builds): leave the tile loop after phase N (0 = run everything)
    // direct placement (round 4, td_split_tiles<.., true>): the fused loop writes a tile's ids straight to out_tokens when the
    // output base of the tile is known in time (decoupled look-back over the per-tile id counts)
    int pack_split;             // td_pack_tokens as a pair: the plain tiles at 8 wavefronts per SIMD, then the others (TD_PACK_SPLIT)
    int direct;                 // try it (family patterns, fused loop, no special cuts)
    unsigned long long* tile_state;  // [n_stiles + 1] per pre-tokenizer tile: status << 62 | value (TS_* below); zeroed by td_prepare
    uint32_t* slab;             // [fused grid][SLAB_RING][SLAB_WORDS] a workgroup's slots of the tile whose placement is pending (stays in the L2)
    uint32_t* direct_tiles;     // (statistics) pre-tokenizer tiles placed directly
};
// per-tile state of the look-back
constexpr unsigned long long TS_NONE = 0ull, TS_AGG = 1ull, TS_PREFIX = 2ull, TS_BROKEN = 3ull, TS_VALUE_MASK = (1ull << 62) - 1ull;
constexpr int SLAB_MAX_MISSES = 2 * 6;        // (2 * K_MISS_LISTED_MAX: both token tiles of a pre-tokenizer tile)
constexpr int SLAB_META = 5136;               // behind the slots (FZ_NPC + 8 of them at most): per merged piece its slot | ids << 16, then its
                                              // token tile (0 / 1) << 12 | position in that tile
constexpr int SLAB_MIDS = SLAB_META + 2 * SLAB_MAX_MISSES;  // ... and the ids of the merged pieces, 64 each (a multiple of 4)
constexpr int SLAB_WORDS = SLAB_MIDS + SLAB_MAX_MISSES * 64;
#ifndef TD_SLAB_RING
#define TD_SLAB_RING 3
#endif
constexpr int SLAB_RING = TD_SLAB_RING;       // slabs per workgroup: a tile is placed SLAB_RING - 1/2 iterations after its slots were written

struct DecodeArgs {
    const Tables* Tp;
    const int32_t* tokens;      // [n]
    int64_t n;
    uint32_t* local_off;        // [n] scratch: byte offset of token i inside its 4096-token chunk
    int64_t* chunk_pref;        // [n/4096 + 2] scratch: byte base of every chunk; [nchunks] = total bytes
    uint32_t* scan_done;        // zeroed by the caller's stream before the launch
    uint8_t* out;               // [out_cap]
    int64_t out_cap;
    int64_t* n_bytes;           // device: total decoded bytes (may be null)
    const int64_t* doc_tok_offsets;  // optional [n_docs+1]: token index where each document starts
    int64_t n_docs;
    int64_t* doc_byte_offsets;  // optional [n_docs+1]: filled with the byte offset of each document in `out`
    int* err;
    long long* err_pos;
};

// One-launch path for inputs of at most 4 KiB (td_small_encode): everything the kernel reads and writes except the tables
// lives in pinned host memory.
constexpr int SM_MAXDOCS = 1024;  // documents of a td_small_encode call (td_api.cpp: SMALL_MAX_DOCS)
struct SmallStatus {
    unsigned long long seq;  // written last (system-scope release): the call's sequence number
    long long err_pos;
    int err, fallback;       // TD_E_* or 0; 1 = a piece above 64 bytes: rerun on the general path
    unsigned int n_tokens;
    unsigned int pad;
};
struct SmallArgs {
    const Tables* Tp;
    const uint8_t* text;         // [n] (pinned host memory)
    const int64_t* doc_offsets;  // [n_docs + 1]
    int32_t* out_tokens;         // [n]
    int64_t* out_offsets;        // [n_docs + 1]
    SmallStatus* status;
    unsigned long long seq;
    int n, n_docs, use_fastpath;
};
hipError_t launch_small_encode(const SmallArgs& a, hipStream_t stream);
// ... and as a kernel that stays for a while (td_small_resident): a request = this header at the start of the pinned input buffer (offsets at
// +64, the text behind them), its sequence number written last; the output buffer as for td_small_encode, with the generation of a kernel
// that has left at byte 40
struct SmallMailbox {
    unsigned long long seq;
    int n, n_docs, use_fastpath, offs_bytes;
};
hipError_t launch_small_resident(const Tables* Tp, void* in, void* out, unsigned long long gen, unsigned long long idle_ticks, hipStream_t stream);
// The same for decode_bytes on at most SMALL_DEC_MAX_TOKENS ids (td_small_decode): ids in, bytes + status out, pinned host memory.
constexpr int SMALL_DEC_MAX_TOKENS = 1024, SMALL_DEC_MAX_BYTES = 16384;
struct SmallDecArgs {
    const Tables* Tp;
    const int32_t* tokens;   // [n] (pinned host memory)
    uint8_t* out;            // [SMALL_DEC_MAX_BYTES]
    SmallStatus* status;     // n_tokens = bytes written; err_pos = index of the first id that is no token; fallback = 1: more bytes than the buffer holds
    unsigned long long seq;
    int n;
};
hipError_t launch_small_decode(const SmallDecArgs& a, hipStream_t stream);

// All launches are asynchronous on `stream`; none of them synchronises or allocates.
// ev (optional, TD_PROF_EVENTS events): ev[0] | td_prepare, td_mark_docs | ev[1] | td_split_tiles (fused: + lookups), td_split_far_* |
// ev[2] | td_probe_tiles (fused: the deferred tiles only) | ev[3] | td_merge_pieces | ev[4] | td_long_pieces, td_giant_pieces,
// td_scan_tiles | ev[5] | td_pack_tokens | ev[6]
constexpr int TD_PROF_EVENTS = 7;
// aux (optional; round 5): a second stream and two events of the handle.  td_long_pieces / td_giant_pieces touch nothing the chain
// td_collect_misses -> td_merge_pieces -> td_copy_dups touches (other pieces, other outputs; both add to tile_extra atomically), and both
// sides are bound by dependent round trips, not by issue or bandwidth: with `aux` the long pieces run on aux->s BESIDE that chain (fork
// behind td_probe_tiles, join in front of td_scan_tiles; inside a capture the two become parallel branches of the graph).
struct LaunchAux { hipStream_t s; hipEvent_t fork, join; };
hipError_t launch_encode(const EncodeArgs& a, hipStream_t stream, hipEvent_t* ev = nullptr, const LaunchAux* aux = nullptr);
// generic split patterns (td_generic.hip), called by launch_encode in place of td_split_tiles / ahead of td_scan_tiles
hipError_t launch_generic_split(const EncodeArgs& a, hipStream_t stream);
hipError_t launch_generic_gaps(const EncodeArgs& a, hipStream_t stream);
// allowed special tokens (td_special.hip): the cuts behind td_mark_docs, the ids in front of td_scan_tiles
hipError_t launch_special_cuts(const EncodeArgs& a, hipStream_t stream);
hipError_t launch_special_ids(const EncodeArgs& a, hipStream_t stream);
// phases: 1 = lengths + offsets (td_decode_len, td_decode_chunks, document byte offsets), 2 = gather (td_decode_copy), 3 = both
hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream, int phases = 3);
int encode_grid_blocks();  // persistent grid size of td_probe_tiles
int fused_grid_blocks();   // persistent grid size of the fused tile loop
int direct_grid_blocks();  // ... with direct placement (SLAB_RING slabs per workgroup)
int merge_grid_blocks();   // persistent grid size of td_merge_pieces

}  // namespace td
