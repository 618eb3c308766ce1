// gfx950 (MI355X) kernels of the tokenizer hot path.  Integer / byte work, HBM- and latency-bound:
// no MFMA anywhere.  Pipeline per encode call (all on one stream, no host round trips):
//
//   td_prepare, td_mark_docs   clear the per-call state; doc_offsets -> one bit per document start
//   td_split_tiles<pattern>    pre-tokenizer: 8 KiB text tiles (+halos) in LDS -> class masks (8x8 bit transpose) ->
//                              bit-parallel scanner from provable synchronisation points -> START bitmap in HBM
//   td_split_far_pieces/_tiles pieces longer than the LDS window (normally none): a wavefront per piece, same matcher over HBM
//   td_probe_tiles             4 KiB tiles: dense piece list -> ONE slot per piece: the id when the piece is a token
//                              (whole-piece table probe), a TOK_MISS marker when it is not, TOK_LONGREF above 64 bytes
//   td_collect_misses          the TOK_MISS pieces of the tiles that have many: onto lists by length class — once per DISTINCT piece of the
//                              call (a table of the pieces seen so far, bytes compared); repeats onto a list of their own (round 5)
//   td_merge_pieces            the listed pieces: byte-pair merge, one LANE per piece (every lane advances its own merge
//                              chain, a merge per round), rows of one length class
//   td_copy_dups               the repeats: id count from the piece they repeat; the pack kernels read the ids there (round 5)
//   td_long_pieces             pieces longer than 64 bytes: lane groups / a wavefront per piece, parts in LDS; multi-byte characters that
//                              provably merge first enter as one part (character seeds, td_common.h; round 5).  With td_giant_pieces on a
//                              second stream beside the three kernels above when the handle has seen long pieces (LaunchAux, td_kernels.h)
//   td_giant_pieces            pieces above 1 KiB: rounds over the whole piece (every pair of the lowest rank at once)
//   td_scan_tiles              device-wide exclusive scan of the per-tile id counts
//   td_pack_plain, td_pack_rest   per-tile slots -> densely packed int32 ids + int64 per-document token offsets (td_pack_tokens: both in one)
//
// Reference behaviour being reproduced: /root/reference/src/tiktoken/tiktoken.cpp:70-128
// (split_text), :169-234 (encode), :282-378 (get_rank / bpe_merge / byte_pair_encode).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "td_kernels.h"

namespace td {

// Ablation hook (tools/gpu_ablate.py: "leave the tile loop after phase N"): compiled in only with -DTD_ABLATE (tools/build_variant.sh);
// release builds carry no trace of it in the hot loops.
#ifdef TD_ABLATE
#define TD_STOP(n) (a.stop_after == (n))
#else
#define TD_STOP(n) false
#endif

// ------------------------------------------------------------------ accessors ---------------
struct LdsSrc {
    const uint8_t* txt;
    const uint32_t* docw;
    int64_t lo, hi;
    __device__ __forceinline__ uint32_t byte(int64_t i) const { return txt[i]; }
    __device__ __forceinline__ bool doc(int64_t i) const { return (docw[i >> 5] >> (i & 31)) & 1u; }
};
struct GlobSrc {
    const uint8_t* text;
    const uint32_t* docbits;
    int64_t lo, hi;
    __device__ __forceinline__ uint32_t byte(int64_t i) const { return text[i]; }
    __device__ __forceinline__ bool doc(int64_t i) const { return (docbits[i >> 5] >> (i & 31)) & 1u; }
};
// Slow path: classes computed on the fly from HBM (pieces / look-ahead that leave the LDS window).
struct GlobAcc {
    using pos_t = int64_t;
    const Tables* T;
    GlobSrc s;
    int64_t n;
    int64_t lim;
    __device__ __noinline__ uint32_t cf(int64_t i) const {
        if (i >= n) return F_DOC;
        uint32_t v = classify_at(*T, s, i);
        if (s.doc(i)) v |= F_DOC;
        return v;
    }
    __device__ __forceinline__ uint32_t byte(int64_t i) const { return i < n ? s.text[i] : 0u; }
    __device__ __noinline__ int64_t scan(int64_t pos) const { return scan_piece(*this, pos, T->pat_flags); }
};

// feature byte of a non-ASCII byte of the LDS window (out of line: the UTF-8 / 2-stage-table walk is only needed for
// non-ASCII text and must not be replicated into the hot ASCII path)
// (the window description travels by value: a struct passed by reference to an out-of-line function has to live in
// scratch memory, and the per-tile store of it was 0.5 GB of HBM writes per launch)
__device__ __noinline__ uint32_t feature_at_v(const uint8_t* ascii_cls, const uint16_t* ucls1, const uint8_t* ucls2, const uint8_t* txt,
                                              const uint32_t* docw, int lo, int hi, int idx) {
    Tables T;  // only the class tables are read here
    T.ascii_cls = ascii_cls; T.ucls1 = ucls1; T.ucls2 = ucls2;
    LdsSrc src;
    src.txt = txt; src.docw = docw; src.lo = lo; src.hi = hi;
    if (idx < src.lo || idx >= src.hi) return FB_X;
    const uint32_t c = classify_at(T, src, idx);
    return feature_of_class(c & CLS_MASK) | ((c & F_CONT) ? (uint32_t)FB_C : 0u);
}

// The table descriptor lives in device memory (EncodeArgs::Tp).  Read through the pointer, its fields are ordinary
// global loads the compiler may not hoist: every `T.piece_mask` inside a loop was a vector load + s_waitcnt vmcnt(0)
// (a full L2 round trip that also drains the loads in flight), every table access a dependent pointer load.  So each
// kernel copies the descriptor once into wave-uniform registers (SGPRs) at entry.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class P>
__device__ __forceinline__ P* uni_ptr(P* p) {
    const uint64_t v = (uint64_t)p;
    return (P*)(((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v));
}
__device__ __forceinline__ Tables uniform_tables(const Tables* p) {
    Tables t;
    t.ascii_cls = uni_ptr(p->ascii_cls);
    t.ucls1 = uni_ptr(p->ucls1);
    t.ucls2 = uni_ptr(p->ucls2);
    t.byte_id = uni_ptr(p->byte_id);
    t.byte_pair = uni_ptr(p->byte_pair);
    t.byte_pair_id = uni_ptr(p->byte_pair_id);
    t.piece_slots = uni_ptr(p->piece_slots);
    t.pair_slots = uni_ptr(p->pair_slots);
    t.piece12_slots = uni_ptr(p->piece12_slots);
    t.tok_off = uni_ptr(p->tok_off);
    t.tok_bytes = uni_ptr(p->tok_bytes);
    t.cseed = uni_ptr(p->cseed);
    t.cseed_pm = uni_ptr(p->cseed_pm);
    t.cseed_nm = uni_ptr(p->cseed_nm);
    t.piece_mask = uni32(p->piece_mask);
    t.pair_mask = uni32(p->pair_mask);
    t.max_id = (int32_t)uni32((uint32_t)p->max_id);
    t.pseudo_base = (int32_t)uni32((uint32_t)p->pseudo_base);
    t.max_token_len = uni32(p->max_token_len);
    t.piece12_mask = uni32(p->piece12_mask);
    t.pat_flags = uni32(p->pat_flags);
    return t;
}

// A kernel's arguments, read where they are used.  The fused tile loop touches forty fields of EncodeArgs and six of the table
// descriptor; loaded at entry (what the compiler does with a by-value argument) they were 100 scalar registers the loop had no room
// for: the allocator parked them in the lanes of two vector registers and fetched them back with v_readlane_b32 in front of every
// use — 594 of the loop's 3078 vector instructions, at 4 issue cycles each (profiles/r6_00_valu_issue_rate.txt), in a loop that is
// bound by vector issue.  Read through a pointer the compiler cannot see through (refresh(): an empty asm that "changes" it), a field
// is an s_load_dword from the kernarg segment at its use — the scalar unit's work, not the vector unit's — and does not outlive
// the phase.  The pointer stays in the constant address space so that the loads are scalar ones.
template <class P>
__device__ __forceinline__ void forget_where_from(P& p) {  // (readfirstlane: behind a branch on a lane's value the compiler takes the pointer for a per-lane one)
    uint32_t lo = (uint32_t)(uint64_t)p, hi = (uint32_t)((uint64_t)p >> 32);
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
    asm volatile("" : "+s"(lo), "+s"(hi));
    p = (P)(((uint64_t)hi << 32) | lo);
}
struct FreshArgs {
    const __attribute__((address_space(4))) EncodeArgs* p;
    __device__ __forceinline__ void refresh() { forget_where_from(p); }
    __device__ __forceinline__ const EncodeArgs* operator->() const { return (const EncodeArgs*)p; }
    __device__ __forceinline__ operator const EncodeArgs&() const { return *(const EncodeArgs*)p; }
};
__device__ __forceinline__ const __attribute__((address_space(4))) EncodeArgs* kernarg_args() {  // (EncodeArgs is the kernels' only argument)
    return (const __attribute__((address_space(4))) EncodeArgs*)__builtin_amdgcn_kernarg_segment_ptr();
}
struct FreshTables {  // (the descriptor is written once, at td_create: constant memory as far as any kernel is concerned)
    const __attribute__((address_space(4))) Tables* p;
    __device__ __forceinline__ void refresh() { forget_where_from(p); }
    __device__ __forceinline__ const Tables* operator->() const { return (const Tables*)p; }
    __device__ __forceinline__ operator const Tables&() const { return *(const Tables*)p; }
};
struct KeptArgs {
    const EncodeArgs* p;
    __device__ __forceinline__ const EncodeArgs* operator->() const { return p; }
    __device__ __forceinline__ operator const EncodeArgs&() const { return *p; }
};
struct KeptTables {
    const Tables* p;
    __device__ __forceinline__ const Tables* operator->() const { return p; }
    __device__ __forceinline__ operator const Tables&() const { return *p; }
};

__device__ __forceinline__ void raise(const EncodeArgs& a, int code, int64_t pos) {
    if (atomicCAS(a.err, 0, code) == 0) *a.err_pos = pos;
}

// exclusive block scan over K_THREADS values (4 wavefronts); total returned to every thread
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(x, d);
        if (lane >= d) x += t;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < K_THREADS / 64; ++w) {
        const uint32_t sw = s_wave[w];
        if (w < wave) woff += sw;
        tot += sw;
    }
    __syncthreads();
    total = tot;
    return woff + x - v;
}

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* a, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------ td_mark_docs ------------
__global__ void td_mark_docs(const int64_t* doc_offsets, int64_t n_docs, int64_t n, uint32_t* docbits,
                             uint32_t* tile_first_doc) {
    // Document offsets are non-decreasing, so a document knows from its two neighbours whether it shares its word of the
    // bitmap or is the first one of its tile: the common case (alone in its 32 bytes, the bitmap is zeroed by td_prepare) is
    // a plain store, and the first document of a tile is exactly one thread's plain store.  (One atomicOr + one atomicMin per
    // document were 62 us per GiB of short paragraphs: 1.7 M atomics.)
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = doc_offsets[d];
        if (p >= 0 && p < n) {
            const int64_t pp = d > 0 ? doc_offsets[d - 1] : -1, pn = d + 1 < n_docs ? doc_offsets[d + 1] : -1;
            const bool shared = (pp >= 0 && (pp >> 5) == (p >> 5)) || (pn >= 0 && pn < n && (pn >> 5) == (p >> 5));
            if (shared) atomicOr(&docbits[p >> 5], 1u << (p & 31));
            else docbits[p >> 5] = 1u << (p & 31);
            // first document of each tile (tile_first_doc is preset to 0xFFFFFFFF)
            if (pp < 0 || pp / K_TILE != p / K_TILE) tile_first_doc[p / K_TILE] = (uint32_t)d;
        }
    }
}

// ------------------------------------------------------------------ td_prepare --------------
__global__ void td_prepare(const EncodeArgs a) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t words = (a.n + 31) / 32 + 2;
    uint4* d4 = reinterpret_cast<uint4*>(a.docbits);  // (hipMalloc alignment; the buffer is padded past `words`)
    for (int64_t i = gid; i < (words + 3) / 4; i += gsz) d4[i] = make_uint4(0, 0, 0, 0);
    for (int64_t i = gid; i <= a.n_tiles; i += gsz) {
        a.tile_extra[i] = 0;
        a.tile_first_doc[i] = 0xFFFFFFFFu;
    }
    for (int64_t i = gid; i <= a.n_stiles; i += gsz) {
        a.tile_flag[i] = 0;
        a.tile_carry[i] = -1;
        a.tile_state[i] = TS_NONE;
    }
    if (gid < a.ctl_reset_words) a.ctl_reset[gid] = 0;
    if (gid < (K_MISS_CLASSES + 1) * COLL_SUBS) a.coll_count[gid * COLL_STRIDE] = 0;  // (the lists of td_collect_misses: records by class, repeats)
    if (a.dedupe) {
        uint4* t4 = reinterpret_cast<uint4*>(a.dd_table);
        for (int64_t i = gid; i < ((int64_t)a.dd_mask + 1) / 2; i += gsz) t4[i] = make_uint4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------ td_prepare_mark ---------
// td_prepare + td_mark_docs as ONE launch (round 6: a dependent launch costs ~5 us of a step whatever it does, and the zeroed bitmap
// was written twice).  The bitmap and tile_first_doc are produced by TEXT RANGE, not by document: a workgroup owns PM_RANGE bytes of
// text (1024 words of docbits, 8 token tiles), finds the first document that starts in its range with a 64-ary search over
// doc_offsets (one wavefront, 3-4 dependent loads for 1.7 M documents instead of a binary search's 21), ORs the range's document
// starts into a bitmap in LDS and stores the 1024 words — every word of docbits is written exactly once, whether a document starts
// in it or not: no zero pass, no atomics in HBM, and the cost does not depend on how long a document is (a thread per document that
// also clears the words up to the next document would leave a single 1 GiB document to one wavefront).  The other per-call state
// (tile_extra, tile_flag / tile_carry / tile_state, the counters, the table of distinct pieces) is cleared grid-stride as before.
__device__ __forceinline__ void wave_sync_lds();  // (below)
constexpr int PM_RANGE = 32768;
constexpr int PM_WAVES = 4;  // wavefronts per workgroup, a range each (nothing in here is workgroup-wide: no barrier)
__global__ __launch_bounds__(64 * PM_WAVES) void td_prepare_mark(const EncodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[PM_WAVES][PM_RANGE / 32];
    __shared__ uint32_t s_first[PM_WAVES][PM_RANGE / K_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + tid, gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gid; i <= a.n_tiles; i += gsz) a.tile_extra[i] = 0;
    for (int64_t i = gid; i <= a.n_stiles; i += gsz) {
        a.tile_flag[i] = 0;
        a.tile_carry[i] = -1;
        a.tile_state[i] = TS_NONE;
    }
    if (gid < a.ctl_reset_words) a.ctl_reset[gid] = 0;
    if (gid < (K_MISS_CLASSES + 1) * COLL_SUBS) a.coll_count[gid * COLL_STRIDE] = 0;
    if (a.dedupe) {
        uint4* t4 = reinterpret_cast<uint4*>(a.dd_table);
        for (int64_t i = gid; i < ((int64_t)a.dd_mask + 1) / 2; i += gsz) t4[i] = make_uint4(0, 0, 0, 0);
    }
    const int64_t words = (a.n + 31) / 32 + 2;  // (what td_prepare cleared; the buffer is padded to whole 16-byte pieces)
    const int64_t nranges = (words * 32 + PM_RANGE - 1) / PM_RANGE;
    const int64_t nd = a.n_docs;
    uint32_t* const bits = s_bits[wv];
    uint32_t* const first = s_first[wv];
    static_assert(PM_RANGE / 32 == 16 * 64, "four 16-byte pieces of the range's bitmap per lane");
    for (int64_t r = (int64_t)blockIdx.x * PM_WAVES + wv; r < nranges; r += (int64_t)gridDim.x * PM_WAVES) {
        const int64_t b0 = r * PM_RANGE, b1 = b0 + PM_RANGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(bits)[lane + 64 * q] = make_uint4(0, 0, 0, 0);
        if (lane < PM_RANGE / K_TILE) first[lane] = 0xFFFFFFFFu;
        // first document d with doc_offsets[d] >= b0 (nd: none).  Documents of about one size start near b0 / n of the way through
        // the offsets: the first round looks there, 64 offsets a 128th of a range's documents apart, and only when that misses does
        // the search start from both ends (64-ary: 3-4 dependent loads for 1.7 M documents)
        int64_t lo = 0, hi = nd;
        if (a.n > 0 && nd > 64) {
            // (in floating point: a guess needs no exact arithmetic, and 64- and 128-bit integer divisions are hundreds of instructions
            // each — the first form of this kernel spent 1000 vector instructions per wavefront on them: 59 us per GiB of text)
            const float dens = (float)nd / (float)a.n;
            const int64_t step = (int64_t)(dens * (float)(PM_RANGE / 64)) + 1;
            const int64_t g = (int64_t)((double)b0 * (double)dens);
            int64_t idx = g + (lane - 32) * step;
            idx = idx < 0 ? 0 : idx >= nd ? nd - 1 : idx;
            const uint64_t b = __ballot(a.doc_offsets[idx] < b0);  // (sorted offsets, non-decreasing idx: a prefix of the lanes)
            const int c = (int)__popcll((unsigned long long)b);
            if (c > 0 && c < 64 && !(b & (b + 1ull))) {  // the answer lies between two of the probes
                lo = __shfl((long long)idx, c - 1) + 1;
                hi = __shfl((long long)idx, c);
            }
        }
        while (hi - lo > 64) {
            const int64_t idx = lo + 1 + (((hi - lo - 1) * (int64_t)lane) >> 6);  // lo < idx < hi, non-decreasing in the lane
            const uint64_t b = __ballot(a.doc_offsets[idx] < b0);
            const int c = (int)__popcll((unsigned long long)b);
            const int64_t below = __shfl((long long)idx, c > 0 ? c - 1 : 0), above = __shfl((long long)idx, c < 64 ? c : 63);
            if (c > 0) lo = below + 1;
            if (c < 64) hi = above;
        }
        {
            const bool less = lo + lane < hi && a.doc_offsets[lo + lane] < b0;
            lo += (int64_t)__popcll((unsigned long long)__ballot(less));
        }
        wave_sync_lds();
        for (int64_t d = lo + lane;; d += 64) {
            const int64_t p = d < nd ? a.doc_offsets[d] : b1;
            if (p >= b0 && p < b1 && p < a.n) {
                atomicOr(&bits[(p - b0) >> 5], 1u << (p & 31));
                atomicMin(&first[(p - b0) / K_TILE], (uint32_t)d);
            }
            if (!((__ballot(p < b1 && d < nd) >> 63) & 1ull)) break;  // (sorted: when the LAST lane's document lies behind the range, so does everything of the later rounds — no further round trip to find that out)
        }
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t gw = (b0 >> 5) + 4 * (lane + 64 * q);
            if (gw < words) reinterpret_cast<uint4*>(a.docbits)[gw >> 2] = reinterpret_cast<const uint4*>(bits)[lane + 64 * q];
        }
        const int64_t t = b0 / K_TILE + lane;
        if (lane < PM_RANGE / K_TILE && t <= a.n_tiles) a.tile_first_doc[t] = first[lane];
        wave_sync_lds();
    }
}

// ------------------------------------------------------------------ shared helpers ----------
constexpr int K_MWORDS = K_WIN / 64;  // 64-byte mask words per window

// 64 consecutive bits of a u32 bit array starting at bit `pos`
__device__ __forceinline__ uint64_t bits64(const uint32_t* arr, int pos) {
    const int w = pos >> 5, sh = pos & 31;
    const uint64_t lo = ((uint64_t)arr[w + 1] << 32) | arr[w];
    const uint64_t hi = arr[w + 2];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// pieces / look-ahead longer than a 64-byte register window: the same matcher on the mask words in LDS (cold path,
// kept out of line so that the hot loop stays small)
template <uint32_t PV>
__device__ __noinline__ int scan_piece_lds(const uint64_t* s_mask, const uint8_t* s_txt, int p) {
    const ArrMaskP mp(s_mask, p, K_LIM);
    return scan_piece_p(mp, [s_txt](int q) { return (uint32_t)s_txt[q]; }, PV);
}

// the 64-byte mask window that starts at window byte `base`
__device__ __forceinline__ void load_bitwin(BitWin& w, const uint64_t* s_mask, int base) {
    const int word = base >> 6, sh = base & 63;
    const uint64_t* lo = s_mask + word * MK_COUNT;
    const uint64_t* hi = lo + MK_COUNT;
#pragma unroll
    for (int k = 0; k < MK_COUNT; ++k) w.m[k] = sh ? (lo[k] >> sh) | (hi[k] << (64 - sh)) : lo[k];
}

// 16 text bytes at global offset g (zero outside [0, n)); one coalesced 16 B/lane load when the base is aligned.
// The byte-wise edge path (first / last window of the text, unaligned text pointer) is out of line: inlined, its
// sixteen address computations were carried (and spilled) through every call site.
__device__ __noinline__ uint4 load_text16_edge(const uint8_t* text, int64_t n, int64_t g) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) {
        const int64_t gg = g + k;
        if (gg >= 0 && gg < n) w[k >> 2] |= (uint32_t)text[gg] << ((k & 3) * 8);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 load_text16(const EncodeArgs& a, int64_t g) {
    if (g >= 0 && g + 16 <= a.n && a.text_aligned) return *reinterpret_cast<const uint4*>(a.text + g);
    if (g + 16 > 0 && g < a.n) return load_text16_edge(a.text, a.n, g);
    return make_uint4(0, 0, 0, 0);
}

// Edge windows of the tile loops (the first and the last windows of the text, an unaligned text pointer): staged straight
// into LDS by an out-of-line routine, byte by byte, zeros outside [0, n).  Interior windows are prefetched into registers
// with plain 16-byte loads and no call anywhere near them: as a per-load "aligned and inside? else call" the out-of-line
// edge path made the compiler keep the prefetch addresses alive across the calls in scratch memory — six 8-byte spill
// stores and five reloads per lane and tile, more bytes than the tile's text (the "2 GB of scratch traffic per GiB" of
// round 2's split kernel).  Returns the OR of the bytes the lane staged.
__device__ __noinline__ uint32_t stage_window_edge(uint8_t* s_txt, const uint8_t* text, int64_t n, int64_t w0, int tid, int n16) {
    uint32_t hib = 0;
    for (int v = tid; v < n16; v += K_THREADS) {
        const int64_t g = w0 + (int64_t)v * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; ++k) {
            const int64_t gg = g + k;
            if (gg >= 0 && gg < n) w[k >> 2] |= (uint32_t)text[gg] << ((k & 3) * 8);
        }
        reinterpret_cast<uint4*>(s_txt)[v] = make_uint4(w[0], w[1], w[2], w[3]);
        hib |= w[0] | w[1] | w[2] | w[3];
    }
    return hib;
}

// Streaming accesses (the text, the ids on their way out, the START bits): marked non-temporal so that they do not push the
// lines that ARE used again — the exact-key table's hot lines and the fused loop's slabs — out of the 4 MB of L2 an XCD has
// (the slabs are rewritten every other tile, and between two uses of a slab line about as much text and output as the L2
// holds streams through it).  -DTD_NO_NT: plain accesses (A/B).
#ifndef TD_NO_NT
#define TD_NT_LOAD(p) __builtin_nontemporal_load(p)
#define TD_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define TD_NT_LOAD(p) (*(p))
#define TD_NT_STORE(v, p) (*(p) = (v))
#endif
typedef uint32_t td_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {  // (the builtin takes scalars and vector types, not HIP's uint4 struct)
    const td_u32x4 v = TD_NT_LOAD(reinterpret_cast<const td_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// LDS traffic between the lanes of ONE wavefront: program order is execution order, the fence keeps the compiler from
// moving the accesses and waits for the outstanding LDS operations
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// the same without waiting for the wavefront's outstanding GLOBAL memory operations (a workgroup-scope fence drains them
// all, e.g. the next tile's prefetch): wavefront scope only keeps the compiler from reordering
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// inclusive add-scan across the wavefront with DPP (no LDS round trips): rows of 16 lanes with row_shr 1/2/4/8 (lanes without
// a source add 0), then row_bcast15 (rows 1 and 3 take the total of rows 0 and 2) and row_bcast31 (rows 2, 3 take rows 0-1)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int /*lane*/) {
    constexpr int ROW_SHR1 = 0x111, ROW_SHR2 = 0x112, ROW_SHR4 = 0x114, ROW_SHR8 = 0x118, ROW_BCAST15 = 0x142, ROW_BCAST31 = 0x143;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_SHR1, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_SHR2, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_SHR4, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_SHR8, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_BCAST15, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ROW_BCAST31, 0xC, 0xF, false);
    return x;
}

// ------------------------------------------------------------------ probe helpers (td_probe_tiles and the fused tile loop) ----
constexpr int K_GIANT_MIN = 1024;  // pieces above this many bytes: td_giant_pieces (= LP_MEDIUM)
constexpr int K_BWIN = K_TILE + 2 * K_MAXSHORT;  // text / START bits staged per tile by this kernel

// length classes of the missed pieces (<= 8, 16, 32, 48, 64 bytes = 1, 1, 2, 3, 4 units of 16 key slots in td_merge_pieces)
__device__ __forceinline__ uint32_t mq_class(uint32_t len) { return len <= 8u ? 0u : len <= 16u ? 1u : len <= 32u ? 2u : len <= 48u ? 3u : 4u; }
__device__ __forceinline__ uint32_t mq_units(uint32_t cls) { return cls < 2u ? 1u : cls; }

#ifndef TD_PROBE_MIN_WAVES
#define TD_PROBE_MIN_WAVES 8  // (measured on 256 MiB of English: 0.49 ms at 8 waves/SIMD, 0.66 ms at 5..7)
#endif

// A piece of the text window in LDS as dwords (zero behind its end): what the cold route hashes and compares.  (Until round 4 it
// took the piece byte by byte — hash_bytes over an LDS byte getter, then piece_lookup's check of a matching slot against the
// token's bytes one global byte load after the other, each waiting for the one before it: 20 % of the fused loop's cycles on
// mixed-script text and on code, where identifiers of 13..30 bytes that ARE tokens paid 13..30 dependent round trips.)
#ifndef TD_COLD_CMP
#define TD_COLD_CMP 2  // dwords of a token compared per step
#endif
struct PieceWords {
    const uint32_t* wp;  // the aligned dword that holds the piece's first byte
    uint32_t sh, len;    // bit offset of that byte in it; bytes
    __device__ __forceinline__ uint32_t tail(uint32_t j) const {  // mask of the piece's bytes in its dword j
        const int nb = (int)len - 4 * (int)j;
        return nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : (1u << (8 * nb)) - 1u;
    }
    __device__ __forceinline__ uint32_t word(uint32_t j) const { return __funnelshift_r(wp[j], wp[j + 1], sh) & tail(j); }
};
__device__ __forceinline__ uint64_t hash_piece_words(const PieceWords& P) {  // = hash_bytes (td_common.h) over the piece's bytes
    uint64_t k = 0x243F6A8885A308D3ull ^ P.len;
#pragma unroll 1
    for (uint32_t j = 0; 4u * j < P.len; j += 2) {
        const uint64_t w = (uint64_t)P.word(j) | ((uint64_t)P.word(j + 1) << 32);
        k = ((k << 23) | (k >> 41)) ^ w;
        k *= 0x9E3779B97F4A7C15ull;
    }
    return k ^ (k >> 31);
}
// = piece_lookup (td_common.h); a slot whose key and length match is checked against the token's bytes sixteen at a time
// (independent loads; tok_bytes is padded behind its last token)
__device__ __forceinline__ int32_t piece_lookup_words(const Tables& T, uint64_t key, const PieceWords& P) {
    uint32_t h = hash_piece(key, P.len) & T.piece_mask;
    for (;;) {
        const PieceSlot s = T.piece_slots[h];
        if (s.len == 0) return NO_RANK;
        if (s.key == key && s.len == P.len) {
            if (P.len <= 8) return (int32_t)s.rank;
            const uintptr_t tb = (uintptr_t)(T.tok_bytes + T.tok_off[s.rank]);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(tb & ~(uintptr_t)3);
            const uint32_t tsh = (uint32_t)(tb & 3) * 8u;
            uint32_t diff = 0;
#pragma unroll 1
            for (uint32_t j = 0; 4u * j < P.len; j += TD_COLD_CMP) {  // (rolled: unrolled to sixteen bytes a step the fused loop spilled 12 more VGPRs, and 3 more with the lookup out of line: +2 % on English)
                uint32_t t[TD_COLD_CMP + 1];
#pragma unroll
                for (int u = 0; u <= TD_COLD_CMP; ++u) t[u] = q[j + u];
#pragma unroll
                for (int u = 0; u < TD_COLD_CMP; ++u) diff |= (__funnelshift_r(t[u], t[u + 1], tsh) & P.tail(j + u)) ^ P.word(j + u);
            }
            if (!diff) return (int32_t)s.rank;
        }
        h = (h + 1) & T.piece_mask;
    }
}

__device__ __forceinline__ int32_t cold_lookup(const Tables& T, const uint32_t* wp, uint32_t sh, uint32_t len) {
    PieceWords P;
    P.wp = wp;
    P.sh = sh;
    P.len = len;
    const uint64_t key = len <= 8 ? ((uint64_t)P.word(0) | ((uint64_t)P.word(1) << 32)) : hash_piece_words(P);
    return piece_lookup_words(T, key, P);
}

// everything the hot probe path leaves out: keys longer than 16 bytes (hashed + verified against the token bytes), probe
// sequences longer than one slot, pieces longer than K_MAXSHORT (handed to td_long_pieces).  Returns the piece's slot.
template <bool WORDS>  // (the fused loop: true; td_probe_tiles keeps the byte-wise form — with the other one it spills 94 VGPRs instead of 14)
__device__ __forceinline__ uint32_t probe_piece_cold(const EncodeArgs& a, const Tables& T, const uint8_t* s_txt, const int32_t* s_byteid,
                                                     uint32_t* s_flags, int64_t wg0, int i, uint32_t len) {
    if (len > (uint32_t)K_MAXSHORT) {
        if (len == 0xFFFFFFFFu) { raise(a, TD_E_SCRATCH, wg0 + i); return 0u; }
        const uint32_t idx = atomicAdd(a.long_count, 1u);
        if (idx >= a.long_cap) { raise(a, TD_E_SCRATCH, wg0 + i); return 0u; }
        LongEntry le;
        le.gs = wg0 + i; le.len = len; le.ntok = 0; le.pool_off = 0;
        a.long_list[idx] = le;
        if (len > (uint32_t)K_GIANT_MIN) atomicAdd(a.giant_count, 1u);  // (td_giant_pieces looks at the list only when there is one)
        if (len > (uint32_t)K_GIANT_MIN && len > a.gp_coop_min) atomicAdd(a.gp_ctl + 3, 1u);  // (... and its workgroups meet at a grid barrier only when one is for all of them)
        atomicOr(s_flags, TILE_HAS_LONG);
        return TOK_LONGREF | idx;
    }
    const uint8_t* pb = s_txt + i;
    if (len == 1) {
        const int32_t id = s_byteid[pb[0]];
        if (id >= T.pseudo_base) raise(a, TD_E_UNKNOWN_BYTE, wg0 + i);
        return (uint32_t)id;
    }
    if (a.use_fastpath) {
        int32_t r;
        if constexpr (WORDS) {
            r = cold_lookup(T, reinterpret_cast<const uint32_t*>(s_txt) + (i >> 2), (uint32_t)(i & 3) * 8u, len);  // (s_txt is 16-byte aligned)
        } else {
            auto get = [pb](uint32_t q) { return (uint32_t)pb[q]; };
            uint64_t key;
            if (len <= 8) {
                key = 0;
                for (uint32_t q = 0; q < len; ++q) key |= (uint64_t)pb[q] << (8 * q);
            } else {
                key = hash_bytes(get, len);
            }
            r = piece_lookup(T, key, len, get);
        }
        if (r != NO_RANK) return (uint32_t)r;
    }
    return TOK_MISS | ((uint32_t)i << 7) | len;  // (the caller notes it: note_miss)
}

// ---- (used by td_merge_pieces and, for the few missed pieces of a tile it places itself, by the fused tile loop) ----
// 20 text bytes at global offset g as five dwords (byte k = bits 8(k&3).. of w[k>>2]); zero past the end of the text
typedef uint32_t U32x4a __attribute__((ext_vector_type(4), aligned(4)));  // (a 16-byte global load needs dword alignment only)
typedef uint32_t U32x2a __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void load_piece_window(const uint8_t* text, int64_t n_text, int64_t g, uint32_t (&w)[5]) {
    const uintptr_t addr = (uintptr_t)(text + g);
    const int64_t g4 = g - (int64_t)(addr & 3);  // text offset of the aligned dword that holds byte g
    if (g4 >= 0 && g4 + 24 <= n_text) {
        // six dwords as two loads (round 5; they were six requests to the vector L1 per lane)
        typedef const U32x4a __attribute__((address_space(1)))* g4_t;
        typedef const U32x2a __attribute__((address_space(1)))* g2_t;
        const U32x4a a = *(g4_t)(uintptr_t)(text + g4);
        const U32x2a b = *(g2_t)(uintptr_t)(text + g4 + 16);
        const uint32_t sh = (uint32_t)(addr & 3) * 8;
        w[0] = __funnelshift_r(a.x, a.y, sh); w[1] = __funnelshift_r(a.y, a.z, sh); w[2] = __funnelshift_r(a.z, a.w, sh);
        w[3] = __funnelshift_r(a.w, b.x, sh); w[4] = __funnelshift_r(b.x, b.y, sh);
    } else {
        for (int k = 0; k < 5; ++k) w[k] = 0;
        for (int k = 0; k < 20; ++k)  // (all twenty bytes, as the fast path has them: lp_seed_setup reads bytes 17..19 too)
            if (g + k >= 0 && g + k < n_text) w[k >> 2] |= (uint32_t)text[g + k] << (8 * (k & 3));
    }
}

// parts + keys of the piece text[g, g + st.len) into the lane's units: the piece's bytes 16 at a time in registers (+ the
// byte behind them), so that the sixteen loads of a unit are independent and go out back to back.  One 8-byte load per part:
// the rank of the byte pair that starts there AND the id of its first byte (Tables::byte_pair_id) — the ids from a table of
// their own were sixteen more requests to the vector L1 per unit (+18 % of td_merge_pieces), in LDS the kilobyte that keeps a
// fifth workgroup off the CU.
__device__ __forceinline__ void mg_init_piece(const uint8_t* text, int64_t n_text, const Tables& T, uint32_t* keys, uint32_t* ids,
                                              const MergeState& st, int64_t g) {
    for (uint32_t c = 0; c * 16u < st.len; ++c) {
        uint32_t w[5];
        load_piece_window(text, n_text, g + 16 * c, w);
        // all sixteen loads first (with the LDS stores in between, every load waited for the one before it), then the slots
        uint64_t rk[16];
        typedef const uint64_t __attribute__((address_space(1)))* gbp_t;  // (global loads, not flat ones)
        gbp_t const bp = (gbp_t)(uintptr_t)T.byte_pair_id;
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const uint32_t bn = (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu;
            rk[j] = bp[(b << 8) | bn];
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t jj = 16u * c + j;
            if (jj < st.len) {
                const uint32_t sl = mg_slot(st.t, jj);
                const int32_t r = (int32_t)(uint32_t)rk[j];
                ids[sl] = (uint32_t)(rk[j] >> 32);
                keys[sl] = (jj + 1 < st.len && r != NO_RANK) ? (((uint32_t)r << 6) | jj) : MG_DEAD;  // (= mg_put)
            }
        }
    }
    mg_pad(keys, st);
}

// ---- parts of the fused loop's direct placement: what a tile with missed pieces needs.  (Tried out of line — they are rare on
// plain text, and with them inlined the loop is 56 KB of code — but the calls cost 79 spilled VGPRs on the hot path: 2.7 -> 3.2 ms.) ----
// One lane per missed piece of the tile (first wavefront, all 64 lanes call): merge it (keys / ids in LDS), ids + slot + position
// to the slab; returns the piece's id count (0: this lane has none).
__device__ __forceinline__ uint32_t fz_merge_piece(const Tables* Tp, const uint8_t* text, int64_t n_text, uint32_t* keys, uint32_t* slab, uint32_t rec,
                                                uint32_t hh, uint32_t n0, long long tile_g0, int* err, long long* err_pos) {
    const Tables T = uniform_tables(Tp);
    const int lane = threadIdx.x & 63;
    uint32_t* const ids = keys + SLAB_MAX_MISSES * 4 * MG_UNIT;
    MergeState st;
    st.len = rec & 127u;
    st.alive = st.len >= 64u ? ~0ull : ((1ull << st.len) - 1ull);
    const uint32_t units = (st.len + 15u) >> 4;
    st.t = wave_incl_scan(units, lane) - units;
    const uint32_t pos = (rec >> 7) & 0xFFFu;
    const int64_t gpos = tile_g0 + (int64_t)hh * K_TILE + pos;
    if (st.len) mg_init_piece(text, n_text, T, keys, ids, st, gpos);
    for (;;) {
        const bool more = mg_round_t<uint64_t>(T, keys, ids, st);
        if (!__any(more)) break;
    }
    uint32_t nt = 0;
    if (st.len) {
        uint32_t* const mo = slab + SLAB_MIDS + lane * 64;
        for (uint64_t al = st.alive; al; al &= al - 1ull) {
            const uint32_t jb = (uint32_t)td_ctz64(al);
            const uint32_t id = ids[mg_slot(st.t, jb)];
            if ((int32_t)id >= T.pseudo_base && atomicCAS(err, 0, TD_E_UNKNOWN_BYTE) == 0) *err_pos = gpos + jb;
            mo[nt++] = id;
        }
        slab[SLAB_META + lane] = ((rec >> 19) + (hh ? n0 : 0u)) | (nt << 16);
        slab[SLAB_META + SLAB_MAX_MISSES + lane] = (hh << 12) | pos;
    }
    return nt;
}
// extra ids of the merged pieces in front of slot k; *mine = the merged piece that IS slot k (or -1)
__device__ __forceinline__ uint32_t fz_shift_of(const uint32_t* meta, uint32_t pnm, uint32_t k, int* mine) {
    uint32_t sh = 0;
    int m = -1;
    for (uint32_t q = 0; q < pnm; ++q) {
        const uint32_t e = meta[q], ms = e & 0xFFFFu;
        if (ms < k) sh += (e >> 16) - 1u;
        if (ms == k) m = (int)q;
    }
    *mine = m;
    return sh;
}
// a tile with merged pieces, placed directly: slot k -> dst[k + the extra ids in front of it]
__device__ __forceinline__ void fz_place_with_misses(const uint32_t* slab, int32_t* dst, uint32_t pnp, uint32_t pnm) {
    const uint32_t* const meta = slab + SLAB_META;
    const uint32_t* const mids = slab + SLAB_MIDS;
    for (uint32_t k = threadIdx.x; k < pnp; k += K_THREADS) {
        const uint32_t v = slab[k];
        int mine = -1;
        const uint32_t o = k + fz_shift_of(meta, pnm, k, &mine);
        if ((v & 0xC0000000u) == TOK_MISS && mine >= 0) {
            const uint32_t nq = meta[mine] >> 16;
            for (uint32_t q = 0; q < nq; ++q) dst[o + q] = (int32_t)mids[mine * 64 + q];
        } else {
            dst[o] = (int32_t)v;
        }
    }
}
// ... staged: the layout td_pack_tokens reads (a merged piece: TOK_MISS | position << 7 | ids, its ids at its own bytes' slots of merge_out)
__device__ __forceinline__ void fz_stage_with_misses(const uint32_t* slab, uint32_t* st0, uint32_t* mo0, uint32_t pnp, uint32_t pn0, uint32_t pnm) {
    const uint32_t* const meta = slab + SLAB_META;
    const uint32_t* const mids = slab + SLAB_MIDS;
    for (uint32_t k = threadIdx.x; k < pnp; k += K_THREADS) {
        uint32_t v = slab[k];
        if ((v & 0xC0000000u) == TOK_MISS) {
            for (uint32_t q = 0; q < pnm; ++q) {
                const uint32_t e = meta[q];
                if ((e & 0xFFFFu) == k) {
                    const uint32_t mp = meta[SLAB_MAX_MISSES + q], hq = mp >> 12, pos = mp & 0xFFFu, nq = e >> 16;
                    uint32_t* const mo = mo0 + (size_t)hq * K_STAGE + pos;
                    for (uint32_t r = 0; r < nq; ++r) mo[r] = mids[q * 64 + r];
                    v = TOK_MISS | TOK_MERGED | (pos << 7) | nq;
                }
            }
        }
        if (k < pn0) st0[k] = v; else st0[K_STAGE + (k - pn0)] = v;
    }
}

// ------------------------------------------------------------------ td_split_tiles ----------
// Pre-tokenizer: the regex split of the reference (CoreBPE::split_text, tiktoken.cpp:70-128) as a
// data-parallel boundary detector.  Output: one bit per text byte in HBM (a.startbits), set where a
// piece starts.  One workgroup per 4 KiB tile (+128 B left / 192 B right halo), persistent grid.
#ifndef TD_SPLIT_MIN_WAVES
#define TD_SPLIT_MIN_WAVES 6
#endif
// PV = the pattern's scanner flags as a compile-time constant: one instantiation per member of the pattern family, so
// the hot scan loop of the Llama-4 pattern carries no trace of the others (as run-time flags they cost 5 % of it).
constexpr int KS_HCAP = 768;   // heads the list holds per tile (what does not fit is matched by the lane that found it)
constexpr int KS_CCAP = 128;

// ---- the fused form (FUSED = true) ---------------------------------------------------------------------------------
// The pre-tokenizer's tile loop goes on where td_split_tiles stops: the text of the tile is in LDS and so are its START bits,
// so the pieces are looked up right there and their slots go to the staging regions in the layout td_probe_tiles writes —
// the text is read from HBM ONCE, the START bitmap is not read back, and the split phases (bound by instruction issue) of
// one workgroup overlap the lookups (bound by the latency of one scattered load per piece) of the others on the same CU.
//   - dense piece list of the whole tile from the START bits (DPP scan), piece k -> lane k mod 256;
//   - the lookups of a lane go out TOGETHER, FZ_PB at a time (a lane has 6-7 pieces in running text): one round trip to
//     the exact-key table instead of one per piece — the workgroup has nothing else to overlap them with;
//   - slot k is stored straight to the token tile's staging region (a tile of KS_TILE bytes is two token tiles of K_TILE
//     bytes: pieces that start in the first half are the first tile's slots);
//   - pieces that are no token (2..64 bytes) are handed to td_merge_pieces exactly as td_probe_tiles hands them over: a token
//     tile with at most K_MISS_LISTED_MAX of them puts their records on the global lists, one with more is flagged;
//   - pieces above K_MAXSHORT bytes: TOK_LONGREF + an entry for td_long_pieces / td_giant_pieces, as before.
// What the window cannot decide (a tile that starts inside a piece longer than the left halo, a piece that leaves the
// window: the td_split_far_* cases) or hold (more than FZ_NPC pieces) is DEFERRED: its token tiles go on a list, and
// td_probe_tiles — which otherwise finds nothing to do — looks their pieces up after the far kernels have completed the
// START bits.  Results are the same slots either way.
constexpr int FZ_NPC = 5120;         // pieces per tile the LDS list holds (a tile with more: deferred)
#ifndef TD_FZ_PB
#define TD_FZ_PB 1
#endif
constexpr int FZ_PB = TD_FZ_PB;      // lookups a lane has in flight
constexpr int FZ_R_MASK = (K_MWORDS + 1) * MK_COUNT * 8;  // bytes of the class masks
constexpr int FZ_R_HEADS = FZ_R_MASK;                     // s_heads behind them
constexpr int FZ_R_COLD = FZ_R_HEADS + (KS_HCAP + 2) * 2;
constexpr int FZ_R_SPLIT_END = FZ_R_COLD + KS_CCAP * 2;
// the same bytes during the token phases
constexpr int FZ_R_PLIST = 0;                                 // u16[FZ_NPC + 8] piece starts (window positions) + end delimiter
constexpr int FZ_CCAP = 768;                                  // pieces a tile can put aside for the long route (more: deferred)
constexpr int FZ_R_COLDK = ((FZ_NPC + 8) * 2 + 15) & ~15;     // u16[FZ_CCAP] pieces put aside for the long route
constexpr int FZ_R_PB = FZ_R_COLDK + FZ_CCAP * 2;             // u16[K_THREADS] pieces before each lane's 32 bytes
constexpr int FZ_R_SM = FZ_R_PB + K_THREADS * 2;              // u32[K_THREADS] START bits of each lane's 32 bytes
constexpr int FZ_R_TOK_END = FZ_R_SM + K_THREADS * 4;
static_assert(SLAB_MIDS >= FZ_NPC + 8, "a slab holds the slots of the fullest tile the fused loop takes");
constexpr int FZ_R_BYTES = ((FZ_R_SPLIT_END > FZ_R_TOK_END ? FZ_R_SPLIT_END : FZ_R_TOK_END) + 15) & ~15;
// s_pd[ring slot]: a tile whose placement is pending (written by the token phases of iteration i, read by the placement in
// the middle of iteration i + SLAB_RING: the slot's turn comes round again just before it is overwritten)
constexpr int PD_KIND = 0;      // 0 nothing pending, 1 its id count is settled (look back, then place or stage), 2 verbatim copy to the staging region
constexpr int PD_TILE = 1, PD_NP = 2, PD_N0 = 3, PD_COUNT = 4, PD_NMISS = 5, PD_FDD = 6, PD_EXT0 = 7;
constexpr int PD_WORDS = 8;     // (what the merged pieces of the tile need is in the slab: SLAB_META)
static_assert(SLAB_MAX_MISSES == 2 * K_MISS_LISTED_MAX, "slab layout");
#ifndef TD_LB_MAX_POLLS
#define TD_LB_MAX_POLLS 2       // look-back: rounds that found a predecessor without a count yet before the tile is staged instead
#endif
#ifndef TD_LB_DRAIN_POLLS
#define TD_LB_DRAIN_POLLS 64    // ... when the workgroup has no tile left to work on meanwhile
#endif
#ifndef TD_LB_MAX_ROUNDS
#define TD_LB_MAX_ROUNDS 4      // ... and rounds of 256 predecessors that all had a count but none a prefix
#endif
#ifndef TD_FUSED_MIN_WAVES
#define TD_FUSED_MIN_WAVES 6  // (256 MiB: English 0.66 ms at 5 and at 6, 0.71 at 4; source code 0.98 / 0.96 / 1.09; mixed-script 1.63 / 1.56 / 1.87)
#endif

// DIRECT (only with FUSED): the loop also places the ids of the tiles whose output base it learns in time (TD_OPT_DIRECT,
// see "placement" below).  An instantiation of its own: the code it adds (20 KB) and the registers it takes cost the loop
// 5-7 % even on the tiles that do not use it.
template <uint32_t PV, bool FUSED, bool DIRECT>
__global__ __launch_bounds__(K_THREADS, FUSED ? TD_FUSED_MIN_WAVES : TD_SPLIT_MIN_WAVES) void td_split_tiles(const EncodeArgs a_kern) {
    static_assert(FUSED || !DIRECT, "direct placement is part of the fused tile loop");
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[K_WIN];
    __shared__ __attribute__((aligned(16))) uint8_t s_R[FZ_R_BYTES];  // class masks | heads | cold list; FUSED: then piece list | slots
    uint64_t* const s_mask = reinterpret_cast<uint64_t*>(s_R);       // [(K_MWORDS + 1) * MK_COUNT] class masks, word-major
    uint16_t* const s_heads = reinterpret_cast<uint16_t*>(s_R + FZ_R_HEADS);  // [KS_HCAP + 2] unresolved heads (window positions)
    uint16_t* const s_cold = reinterpret_cast<uint16_t*>(s_R + FZ_R_COLD);    // [KS_CCAP] piece starts the branch-free matcher left open
    __shared__ uint32_t s_start[K_WIN / 32 + 3];  // bit i: a piece starts at window byte i
    __shared__ uint32_t s_doc[K_WIN / 32 + 2];
    __shared__ uint8_t s_lut[128];                // ASCII byte -> feature byte
    __shared__ uint8_t s_fcls[16];                // class -> feature byte
    __shared__ uint32_t s_nh, s_cur, s_ncold, s_nonascii;
    __shared__ int s_last;
    __shared__ int s_cross;                       // FUSED: window position of the first piece start at/after the tile end (-1: unknown)
    __shared__ uint32_t s_defer;                  // FUSED: the window cannot decide this tile (td_split_far_* will)
    __shared__ int s_next[2];                     // FUSED: the tiles this workgroup takes after the current one (next, the one after)

    const int tid = threadIdx.x;
#ifdef TD_FUSED_TIMING
    unsigned long long tt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_readcyclecounter(), t_total0 = t_last, n_tiles_done = 0;
#define FZ_TICK(i) { const unsigned long long t_now = __builtin_readcyclecounter(); tt[i] += t_now - t_last; t_last = t_now; FZ_FRESH }
#else
#define FZ_TICK(i) FZ_FRESH
#endif
    // The arguments and the table descriptor are READ WHERE THEY ARE USED (scalar loads from the kernarg segment / from the descriptor
    // as constant memory) and forgotten at every phase boundary, see FreshArgs.  -DTD_ARGS_IN_REGISTERS: read once at entry (rounds 1-5).
#ifndef TD_ARGS_IN_REGISTERS
    (void)a_kern;
    FreshArgs a{kernarg_args()};
    FreshTables T{(const __attribute__((address_space(4))) Tables*)a->Tp};
#define FZ_FRESH { a.refresh(); T.refresh(); }
#else
    const KeptArgs a{&a_kern};
    const Tables T_kept = uniform_tables(a->Tp);
    const KeptTables T{&T_kept};
#define FZ_FRESH
#endif
    for (int q = tid; q < 128; q += K_THREADS) s_lut[q] = (uint8_t)feature_of_class(T->ascii_cls[q]);
    if (tid < 16) s_fcls[tid] = (uint8_t)feature_of_class((uint32_t)tid);
    if (tid == 0) s_nonascii = 0;
    // FUSED: state of the token phases (declared unconditionally; the plain instantiation never touches it and the
    // compiler drops it)
    __shared__ __attribute__((aligned(16))) uint32_t s_kmask[FUSED ? (P12_MAXLEN + 1) * 4 : 4];  // row len: byte masks of a len-byte key
    __shared__ uint32_t s_wave[8];
    __shared__ uint32_t s_hflags[2], s_nrec[2], s_ncoldp;          // per token tile of the pair: TILE_HAS_LONG; missed pieces
    __shared__ uint32_t s_rec[2][K_MISS_LISTED_MAX];               // the first few: slot << 19 | tile position << 7 | length
    __shared__ unsigned long long s_pend[48];                      // miss-list entries of the last tiles, not appended yet
    __shared__ uint32_t s_npend;
    __shared__ uint32_t s_flagl[32];                               // flagged token tiles of this workgroup, not appended yet
    __shared__ uint32_t s_nflagl;
    // direct placement (see place_prev below): the tile whose ids wait in this workgroup's slab
    __shared__ uint32_t s_pd[DIRECT ? SLAB_RING : 1][PD_WORDS];  // (not referenced without DIRECT: no LDS then)
    __shared__ uint32_t s_stat[2];                                 // tiles placed directly / staged because their base was not known in time
    __shared__ unsigned long long s_lbv[4];                        // look-back: what a wavefront's 64 predecessors end in
    __shared__ uint32_t s_lbs[4][3];
    __shared__ uint32_t s_brk;                                     // this workgroup has seen the chain of counts broken (sticky)
    // first wavefront: the np entries waiting in s_pend go to their class's global list; ONE atomic per class (td_probe_tiles)
    auto append_pending = [&](uint32_t npd) {
        const int ln = tid & 63;
        const bool have = (uint32_t)ln < npd;
        const unsigned long long rec = have ? s_pend[ln] : 0ull;
        const uint32_t c = mq_class((uint32_t)rec & 127u);
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)K_MISS_CLASSES; ++q) {
            const uint64_t b = __ballot(have && c == q);
            if (b) {
                const int leader = (int)td_ctz64(b);
                uint32_t at = 0;
                if (ln == leader) at = atomicAdd(&a->miss_count[q], (uint32_t)__popcll((unsigned long long)b));
                at = (uint32_t)__shfl((int)at, leader);
                if (have && c == q) a->miss_list[(size_t)q * a->miss_cap + at + (uint32_t)__popcll((unsigned long long)(b & ((1ull << ln) - 1ull)))] = rec;
            }
        }
    };
    if constexpr (FUSED) {
        if (tid < (int)(P12_MAXLEN + 1) * 4) {
            const int len = tid >> 2, w = tid & 3, nb = len - 4 * w;  // bytes of dword w that belong to a len-byte key
            s_kmask[tid] = w == 3 || nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
        }
        if (tid == 0) { s_nflagl = 0; s_npend = 0; s_brk = 0; }
        if (DIRECT && tid < SLAB_RING * PD_WORDS) (&s_pd[0][0])[tid] = 0;
        if (tid < 2) s_stat[tid] = 0;
    }
    __syncthreads();

#ifndef TD_FUSED_NPRE
#define TD_FUSED_NPRE 0
#endif
    constexpr int NPF = (K_WIN / 16 + K_THREADS - 1) / K_THREADS;  // 16-byte pieces per lane for one window
    // ... of which the two-kernel form prefetches all into registers one tile ahead, and the FUSED loop NONE: its token
    // phases have no room for them — the compiler parked them in scratch memory behind a wait (load, wait, scratch store at
    // the prefetch; scratch load at the top of the next tile: 16 KiB of scratch traffic per 8 KiB tile, and the latency the
    // prefetch was to hide waited for on the spot).  The tile's text is requested at the top of its iteration; five other
    // workgroups on the CU cover the wait.  1024 MiB of English, same box: 2.15 ms with all three pieces prefetched, 2.07
    // with one, 1.95 with none (-DTD_FUSED_NPRE=3 / 1 / 0).
    constexpr int NPRE = FUSED ? TD_FUSED_NPRE : NPF;
    uint4 pf[NPF];
    const int64_t nwords = (a->n + 31) >> 5;
    // (uniform) the whole window lies inside the text and 16-byte loads are aligned: prefetched into registers; the others
    // (first / last windows, unaligned text) are staged by stage_window_edge when their turn comes
    auto interior = [&](int64_t w0) { return a->text_aligned && w0 >= 0 && w0 + K_WIN <= a->n; };
    // what a tile needs besides its text is requested with it: the document bits of the window (266 words: one per lane and
    // ten more) and, FUSED, the first documents of its two token tiles.  (Loaded when the tile's turn came they were a
    // global-memory round trip at the top of every tile and another one in front of the document slots: 13 % + 5 % of the
    // fused loop.)
    uint32_t pfd0 = 0, pfd1 = 0, pffd0 = 0xFFFFFFFFu, pffd1 = 0xFFFFFFFFu;
#ifdef TD_TEXT_L2_PREFETCH
    uint32_t pft = 0;  // (FUSED) one byte of every 64-byte line of the next window: pulls the text into the L2, see load_window
#endif
    static_assert(K_WIN / 32 <= 2 * K_THREADS, "two prefetched document words per lane cover the window");
    auto load_window = [&](int64_t w0) {
        {
            const int64_t gw0 = (w0 >> 5) + tid, gw1 = gw0 + K_THREADS;
            pfd0 = (gw0 >= 0 && gw0 < nwords) ? a->docbits[gw0] : 0u;
            pfd1 = (tid < K_WIN / 32 - K_THREADS && gw1 >= 0 && gw1 < nwords) ? a->docbits[gw1] : 0u;
            if (FUSED) {
                const int64_t t4 = (w0 + K_HL) / K_TILE;  // first token tile of the window's tile
                pffd0 = t4 < a->n_tiles ? a->tile_first_doc[t4] : 0xFFFFFFFFu;
                pffd1 = t4 + 1 < a->n_tiles ? a->tile_first_doc[t4 + 1] : 0xFFFFFFFFu;
            }
        }
        if (!interior(w0)) return;
#ifdef TD_TEXT_L2_PREFETCH
        // (Measured in round 4 and NOT in, -DTD_TEXT_L2_PREFETCH: the next window's text cannot be prefetched into registers, NPRE
        // above, and waiting for it at the top of its iteration is 14 % of the loop — so ONE byte per 64-byte line and lane, a
        // single register across the token phases that nobody looks at, pulls the window into this XCD's L2.  1.951 -> 2.007 ms
        // per GiB of English, 0.766 -> 0.791 ms per 256 MiB of code: the one register is 1 -> 5 spilled VGPRs at the cap.)
        if (FUSED && NPRE == 0 && tid < (K_WIN + 63) / 64) pft = *reinterpret_cast<const volatile uint8_t*>(a->text + w0 + 64 * tid);
#endif
        const uint4* src16 = reinterpret_cast<const uint4*>(a->text + w0);
#pragma unroll
        for (int q = 0; q < NPRE; ++q)
            if (q < NPF - 1 || tid < K_WIN / 16 - (NPF - 1) * K_THREADS) pf[q] = nt_load16(src16 + q * K_THREADS + tid);
    };
#pragma unroll
    for (int q = 0; q < NPF; ++q) pf[q] = make_uint4(0, 0, 0, 0);
    // The first tile: dealt by blockIdx in the two-kernel form and with TD_DRAW_TWO_AHEAD; DRAWN like every other one otherwise —
    // a workgroup that has not started yet (the grid is sized to what the occupancy query says is resident; when fewer are,
    // the rest start when the first ones are done) then holds no tile the look-back of the others would wait for
    int first_tile = (int)blockIdx.x;
    if constexpr (DIRECT) {
        if (tid == 0) s_next[0] = (int)atomicAdd(a->tile_draw, 1u);
        __syncthreads();
        first_tile = __builtin_amdgcn_readfirstlane(s_next[0]);
        __syncthreads();
    }
    if (first_tile < a->n_stiles) load_window((int64_t)first_tile * KS_TILE - K_HL);
    const int tid_outer = tid;
    // FUSED: the workgroups DRAW their tiles.  The grid is persistent (as many workgroups as fit the chip) and the SIMDs
    // issue from their oldest wavefront first, so the workgroups that came to a CU first run faster than the ones that
    // came last (phase timers: 0.93 M cycles for 22 tiles in workgroup 211, 1.36 M for 21 in workgroup 1477); dealt
    // round-robin the kernel ended with the young workgroups running alone on half-empty CUs.  A workgroup's first two
    // tiles are dealt, the others come from a counter: the draw for the tile AFTER the next one is issued together with the
    // next tile's text prefetch, a whole iteration before it is needed.  1024 MiB of English: 2.33 -> 2.15 ms; 256 MiB of
    // mixed-script text: 1.51 -> 1.34 ms (same box, -DTD_FUSED_STATIC_TILES against the default).
#ifndef TD_FUSED_STATIC_TILES
    constexpr bool DRAW = FUSED;
#else
    constexpr bool DRAW = false;
#endif
    int next_tile = 0, par = 0;
    int ring = 0;  // (uniform) the ring slot of this iteration: iteration number mod SLAB_RING
    if (DRAW && !DIRECT && tid == 0) s_next[0] = (int)(blockIdx.x + gridDim.x);  // (read behind several barriers)
    // (FUSED with direct placement: the loop runs one more time than the workgroup has tiles — the iteration without a tile
    // places the last tile's ids, place_prev below)
    for (int tile = first_tile;; tile = DRAW ? next_tile : tile + (int)gridDim.x) {
        const bool have = tile < a->n_stiles;  // (uniform)
        if (!have) {  // (s_pd: written in front of the barrier that ends an iteration)
            bool pending = false;
            if constexpr (DIRECT)
                for (int r = 0; r < SLAB_RING; ++r) pending = pending || s_pd[r][PD_KIND] != 0u;
            if (!pending) break;
        }
        // The lane index is made opaque once per tile: everything derived from it is then recomputed inside the iteration (a
        // few integer operations) instead of being hoisted out of the tile loop — the compiler hoisted dozens of such per-lane
        // values, ran out of registers and parked them in scratch memory, reloading them in the hot loops.
        int tid_opaque = tid_outer;
        asm volatile("" : "+v"(tid_opaque));
        const int tid = tid_opaque, lane = tid & 63;
        const int64_t tile_g0 = (int64_t)tile * KS_TILE;
        const int64_t wg0 = tile_g0 - K_HL;  // global offset of window index 0 (multiple of 64)
        const int tile_hi = K_HL + (int)((a->n - tile_g0 < KS_TILE) ? (a->n - tile_g0) : KS_TILE);
        static_assert(KS_CHUNK == 32, "a lane's stride is one 32-bit word of the masks");
        const uint32_t fd0 = pffd0, fd1 = pffd1;  // (FUSED) first documents of this tile's token tiles
        // ---- placement of an EARLIER tile of this workgroup, part 1 (direct placement, round 4) ----
        // Rounds 1-3 wrote every id twice: a slot per piece into the staging region, then td_pack_tokens moved the slots to
        // their place once a device-wide scan had the bases (0.59 of 2.65 ms and 40 % of the fabric traffic of a step).  Here
        // the token phases leave a tile's slots in a slab of this workgroup (SLAB_RING of them, used in turn), settle the tile's id
        // count inside the loop (its few missed pieces are merged by the first wavefront) and publish it; SLAB_RING - 1
        // iterations later — the workgroups that have the tiles in front have had that long to publish theirs — the
        // workgroup looks back over the published counts for the tile's base (decoupled look-back: per tile a status word,
        // "count" or "inclusive prefix") and moves the slots from the slab to where they belong, document offsets included.
        // The loads this takes (the slots, the first documents, the first 256 status words) are requested HERE, in front of
        // the text of the tile whose turn it is: one wait covers them all.  The wait for a predecessor is BOUNDED: a tile
        // whose base is not known after TD_LB_MAX_POLLS more rounds is copied to the staging region instead and placed by
        // td_pack_tokens (its count stays published: the chain goes on), and a tile whose count cannot be settled here
        // (pieces above 64 bytes, more missed pieces than the lists take, what the window cannot decide) breaks the chain
        // for good: it and everything behind it are staged as in round 3.
        // (What this does NOT save is fabric traffic: the L2 of this chip does not keep written lines — the slots come back
        // from the memory side, Infinity Cache at best — see DESIGN.md.)
        typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at a dword-aligned address
        const uint32_t pkind = DIRECT ? s_pd[ring][PD_KIND] : 0u;  // (uniform)
        int B = 0;
        uint32_t pnp = 0, pn0 = 0, pcount = 0, pnm = 0, pext0 = 0, pdsl = 0;
        uint4 xs[3];
        int64_t pdm = 0, pdpos = 0;
        unsigned long long stw = 0;
        const uint32_t* const pslab = a->slab + ((size_t)blockIdx.x * SLAB_RING + ring) * SLAB_WORDS;
        if (DIRECT && pkind) {
            B = (int)s_pd[ring][PD_TILE];
            pnp = s_pd[ring][PD_NP]; pn0 = s_pd[ring][PD_N0]; pcount = s_pd[ring][PD_COUNT]; pnm = s_pd[ring][PD_NMISS]; pext0 = s_pd[ring][PD_EXT0];
            const uint32_t nv4 = (pnp + 3u) >> 2;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                xs[q] = make_uint4(0, 0, 0, 0);
                if ((uint32_t)tid + 256u * q < nv4) xs[q] = reinterpret_cast<const uint4*>(pslab)[tid + 256 * q];
            }
            const uint32_t pfdd = s_pd[ring][PD_FDD];
            pdm = (int64_t)pfdd + tid;
            pdpos = a->n;
            if (pkind == 1u && pfdd != 0xFFFFFFFFu && pdm < a->n_docs) { pdpos = a->doc_offsets[pdm]; pdsl = a->doc_slot[pdm]; }
            const int idx0 = B - 1 - tid;  // my predecessor (tile -1: an inclusive prefix of 0)
            stw = (pkind == 1u && idx0 >= 0) ? __hip_atomic_load(&a->tile_state[idx0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (TS_PREFIX << 62);
        }
        if (have) {
#ifdef TD_TEXT_L2_PREFETCH
        if (FUSED) asm volatile("" :: "v"(pft));  // (see load_window: the prefetch byte's register is free from here on)
#endif

        // ---- phase 0: stage the text window and the document bits (two-kernel form: the text was requested one iteration ago,
        //      registers pf[]; FUSED: now, see NPRE) --------------
        uint32_t hib = 0;  // (any byte >= 0x80 in what this lane stages?)
        if (interior(wg0)) {
            // (the pieces that are not prefetched — the window's tail, a few lanes — are requested now and staged last)
#pragma unroll
            for (int q = NPRE; q < NPF; ++q)
                if (q < NPF - 1 || tid < K_WIN / 16 - (NPF - 1) * K_THREADS) pf[q] = nt_load16(reinterpret_cast<const uint4*>(a->text + wg0) + q * K_THREADS + tid);
#pragma unroll
            for (int q = 0; q < NPF; ++q)
                if (q < NPF - 1 || tid < K_WIN / 16 - (NPF - 1) * K_THREADS) {
                    reinterpret_cast<uint4*>(s_txt)[q * K_THREADS + tid] = pf[q];
                    hib |= pf[q].x | pf[q].y | pf[q].z | pf[q].w;
                }
        } else {
            hib = stage_window_edge(s_txt, a->text, a->n, wg0, tid, K_WIN / 16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int w = tid + q * K_THREADS;
            if (w < K_WIN / 32) {
                uint32_t dw = q ? pfd1 : pfd0;
                const int64_t g = wg0 + (int64_t)w * 32;  // bytes past the end of the text: "end of subject" sentinels
                if (g + 32 > a->n) dw |= (g >= a->n) ? 0xFFFFFFFFu : ~((1u << (int)(a->n - g)) - 1u);
                s_doc[w] = dw;
            }
        }
        // DIRECT: the NEXT tile of this workgroup is drawn now and read behind the boundary phases (round 3 drew two tiles ahead: a
        // tile then ran one and a half iterations after its number was handed out, and since the workgroups of a CU do not
        // run at one speed, tiles next to each other were up to 15 us apart — what the look-back of the direct placement waits for)
        if (DRAW && DIRECT && tid == 0) s_next[0] = (int)atomicAdd(a->tile_draw, 1u);
        // next tile of this workgroup.  (FUSED: its document bits are requested behind the boundary phases instead, its text
        // when its turn comes)
        if (!FUSED && tile + (int)gridDim.x < a->n_stiles) load_window(wg0 + (int64_t)gridDim.x * KS_TILE);
        for (int w = tid; w < K_WIN / 32 + 3; w += K_THREADS) s_start[w] = 0;
        if (tid == 0) { s_nh = 0; s_cur = 0; s_ncold = 0; s_last = -1; s_cross = -1; s_defer = 0; }
        if (__ballot((hib & 0x80808080u) != 0) && lane == 0) s_nonascii = 1;  // (reset behind phase 1; __syncthreads_or costs extra barriers)
        }  // (have)
        // ---- placement, part 2: the first round of the look-back goes through the barrier that ends the staging ----
        auto lb_post = [&](unsigned long long w) {  // what a wavefront's 64 predecessors end in -> LDS
            const uint32_t status = (uint32_t)(w >> 62);
            const uint64_t na = __ballot(status != (uint32_t)TS_AGG);
            const int first = na ? (int)td_ctz64(na) : 64;
            const uint32_t mysum = wave_incl_scan(lane < first ? (uint32_t)w : 0u, lane);  // (a count is below 2^16)
            if (lane == 63) { s_lbs[tid >> 6][0] = (uint32_t)first; s_lbs[tid >> 6][1] = mysum; }
            if (lane == first) { s_lbs[tid >> 6][2] = status; s_lbv[tid >> 6] = w & TS_VALUE_MASK; }
        };
        if (DIRECT && pkind == 1u) lb_post(stw);
        __syncthreads();
        if (DIRECT && pkind) {
            const uint32_t* const slab = pslab;
            const uint32_t* const meta = slab + SLAB_META;
            const int64_t pg0 = (int64_t)B * KS_TILE, pg1 = (pg0 + KS_TILE < a->n) ? pg0 + KS_TILE : a->n;
            long long base = -1;  // >= 0: the tile's base; -1: the chain is broken in front of it; -2: not known in time
            if (pkind == 1u) {
                long long acc = 0;
                int j = B - 1, polls = 0, rounds = 0;
                for (;;) {
                    uint32_t tstatus = (uint32_t)TS_AGG;  // what the chain of counts ends in (TS_AGG: not inside these 256)
                    unsigned long long tval = 0;
                    int tdist = 0;
#pragma unroll
                    for (int w = 0; w < K_THREADS / 64; ++w) {
                        if (tstatus == (uint32_t)TS_AGG) {
                            acc += s_lbs[w][1];
                            const int f = (int)s_lbs[w][0];
                            if (f < 64) { tstatus = s_lbs[w][2]; tval = s_lbv[w]; tdist = w * 64 + f; }
                        }
                    }
                    if (tstatus == (uint32_t)TS_PREFIX) { base = acc + (long long)tval; break; }
                    if (tstatus == (uint32_t)TS_BROKEN) { base = -1; break; }
                    if (tstatus == (uint32_t)TS_NONE) {
                        // (an iteration without a tile — the workgroup's last placements — has nothing else to do: it waits longer)
                        if (++polls > (have ? TD_LB_MAX_POLLS : TD_LB_DRAIN_POLLS)) { base = -2; break; }
                        j -= tdist;  // (the counts in front of it are in acc)
                        __builtin_amdgcn_s_sleep(8);
                    } else {
                        if (++rounds >= TD_LB_MAX_ROUNDS) { base = -2; break; }  // (a chain of counts this long: the tiles in front are not being placed)
                        j -= K_THREADS;
                    }
                    __syncthreads();  // (everybody has read s_lbs)
                    const int idx = j - tid;
                    lb_post(idx >= 0 ? __hip_atomic_load(&a->tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (TS_PREFIX << 62));
                    __syncthreads();
                }
            }
            FZ_TICK(8)
            const int tB4 = B * (KS_TILE / K_TILE);
            const bool ptwo = tB4 + 1 < a->n_tiles;
            uint32_t* const st0 = a->stage + (size_t)tB4 * K_STAGE;
            auto shift_of = [&](uint32_t k, int& mine) { return fz_shift_of(meta, pnm, k, &mine); };
            if (pkind == 1u && base >= 0) {
                // -- direct: slot k of the tile -> out_tokens[base + k + the extra ids of the merged pieces in front of it]
                int32_t* const dst = a->out_tokens + base;
                const bool fits = base + (long long)pcount <= (long long)a->out_cap;  // (else: td_scan_tiles raises TD_E_CAPACITY; nothing is written)
                if (fits && pnm == 0u) {  // every slot an id: 16-byte stores (at dword-aligned addresses)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const uint32_t v = (uint32_t)tid + 256u * q;
                        if (4u * v + 4u <= pnp) {
                            u32x4a4 o;
                            o.x = xs[q].x; o.y = xs[q].y; o.z = xs[q].z; o.w = xs[q].w;
                            TD_NT_STORE(o, reinterpret_cast<u32x4a4*>(dst + 4u * v));
                        } else if (4u * v < pnp) {  // the tile's last, partial piece
                            dst[4u * v] = (int32_t)xs[q].x;
                            if (4u * v + 1u < pnp) dst[4u * v + 1u] = (int32_t)xs[q].y;
                            if (4u * v + 2u < pnp) dst[4u * v + 2u] = (int32_t)xs[q].z;
                        }
                    }
                    for (uint32_t k = 3072u + tid; k < pnp; k += K_THREADS) dst[k] = (int32_t)slab[k];  // (a tile of more than 3072 pieces)
                } else if (fits) {
                    fz_place_with_misses(slab, dst, pnp, pnm);
                }
                {   // the documents that start in the tile (their slots: written by the token phases, a.doc_slot)
                    if (pdpos < pg1) {
                        const uint32_t k = pdsl + (pdpos >= pg0 + K_TILE ? pn0 : 0u);
                        int mine = -1;
                        TD_NT_STORE((int64_t)(base + (long long)(k + (pnm ? shift_of(k, mine) : 0u))), &a->out_offsets[pdm]);
                    }
                    if (__all(pdpos < pg1) && tid >= K_THREADS - 64) {  // more than 256 documents start in the tile: the last wavefront takes the rest
                        for (int64_t d = pdm + 64; d < a->n_docs; d += 64) {
                            const int64_t pd = a->doc_offsets[d];
                            if (pd >= pg1) break;
                            const uint32_t k = a->doc_slot[d] + (pd >= pg0 + K_TILE ? pn0 : 0u);
                            int mine = -1;
                            a->out_offsets[d] = base + (long long)(k + (pnm ? shift_of(k, mine) : 0u));
                        }
                    }
                }
                if (tid == 0) {
                    a->tile_count[tB4] = pn0 | TILE_DIRECT;
                    a->tile_extra[tB4] = pext0;
                    if (ptwo) {
                        a->tile_count[tB4 + 1] = (pnp - pn0) | TILE_DIRECT;
                        a->tile_extra[tB4 + 1] = pcount - pnp - pext0;
                    }
                    __hip_atomic_store(&a->tile_state[B], (TS_PREFIX << 62) | (unsigned long long)(base + (long long)pcount), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    s_stat[0] += 1u;
                }
            } else {
                // -- staged: the slots go to the tile's staging regions in the layout td_pack_tokens reads (a merged piece:
                //    TOK_MISS | position << 7 | ids, its ids at its own bytes' slots of merge_out)
                if (pnm == 0u || pkind != 1u) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const uint32_t k0 = 4u * ((uint32_t)tid + 256u * q);
                        const uint32_t xv[4] = {xs[q].x, xs[q].y, xs[q].z, xs[q].w};
#pragma unroll
                        for (uint32_t e = 0; e < 4; ++e) {
                            const uint32_t k = k0 + e;
                            if (k < pnp) { if (k < pn0) st0[k] = xv[e]; else st0[K_STAGE + (k - pn0)] = xv[e]; }
                        }
                    }
                    for (uint32_t k = 3072u + tid; k < pnp; k += K_THREADS) { const uint32_t v = slab[k]; if (k < pn0) st0[k] = v; else st0[K_STAGE + (k - pn0)] = v; }
                } else {
                    fz_stage_with_misses(slab, st0, a->merge_out + (size_t)tB4 * K_STAGE, pnp, pn0, pnm);
                }
                if (pkind == 1u && tid == 0) {
                    uint32_t m0 = 0, m1 = 0;
                    for (uint32_t q = 0; q < pnm; ++q) { if (meta[SLAB_MAX_MISSES + q] >> 12) m1 = 1; else m0 = 1; }
                    a->tile_count[tB4] = pn0 | (m0 ? TILE_MISS_LISTED : 0u);
                    a->tile_extra[tB4] = pext0;
                    if (ptwo) {
                        a->tile_count[tB4 + 1] = (pnp - pn0) | (m1 ? TILE_MISS_LISTED : 0u);
                        a->tile_extra[tB4 + 1] = pcount - pnp - pext0;
                    }
                    if (base == -1) {  // the chain is broken in front of this tile: the tiles behind it need not look further
                        __hip_atomic_store(&a->tile_state[B], TS_BROKEN << 62, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_brk = 1u;
                    } else {
                        s_stat[1] += 1u;  // (statistics: not known in time)
                    }
                }
            }
            if (tid == 0) s_pd[ring][PD_KIND] = 0;  // (read again at the top of the next iteration: barriers in between)
            FZ_TICK(11)
        }
        if (!have) {  // (an iteration without a tile: it placed one pending tile; the others' turns follow)
            __syncthreads();
            ring = __builtin_amdgcn_readfirstlane(ring + 1 == SLAB_RING ? 0 : ring + 1);
            continue;
        }
        {
        FZ_TICK(0)
        const bool tile_ascii = !s_nonascii;

        // ---- phase 1: class masks.  Every lane takes 8 text bytes: feature byte per byte (ASCII: 128-B LUT in LDS;
        //      otherwise UTF-8 decode + 2-stage Unicode table in L2), an 8x8 bit transpose turns the 8 feature bytes
        //      into the 8-bit slices of the class masks, the SYNC slice follows from them and the neighbour's last
        //      byte; 12 byte stores put the slices into the word-major mask array ------------------------------------
        LdsSrc src;
        src.txt = s_txt;
        src.docw = s_doc;
        src.lo = (wg0 < 0) ? -wg0 : 0;
        src.hi = (a->n - wg0 < K_WIN) ? (a->n - wg0) : K_WIN;
        if (tid < MK_COUNT) s_mask[K_MWORDS * MK_COUNT + tid] = 0;  // zero word behind the last one
        // masks of one item from its eight feature bytes and the feature byte in front of them
        auto emit_masks = [&](int it, uint32_t flo, uint32_t fhi, uint32_t pf) {
            const uint64_t P = transpose8x8(((uint64_t)fhi << 32) | flo);  // byte k = bit plane of feature bit k
            const uint32_t plo = (uint32_t)P, phi = (uint32_t)(P >> 32);
            const uint32_t mU = plo & 0xFF, mW = (plo >> 8) & 0xFF, mX = (plo >> 16) & 0xFF, mS = plo >> 24;
            const uint32_t nraw = phi & 0xFF, mCR = (phi >> 8) & 0xFF, mSL = (phi >> 16) & 0xFF, mC = phi >> 24;
            const uint32_t mN = nraw & ~mX & ~mS, mA = nraw & mX, mSP = nraw & mS;
            const uint32_t mD = reinterpret_cast<const uint8_t*>(s_doc)[it];
            const uint32_t mSY = sync_byte(mU, mW, mX, mS, mN, mCR, mSL, mC, mD, mA, pf, PV);
            static_assert(MK_U == 0 && MK_W == 1 && MK_X == 2 && MK_S == 3 && MK_N == 4 && MK_CR == 5 && MK_TR == 6 &&
                          MK_C == 7 && MK_D == 8 && MK_A == 9 && MK_SP == 10 && MK_SYNC == 11, "mask order");
            uint8_t* o = reinterpret_cast<uint8_t*>(s_mask + (it >> 3) * MK_COUNT) + (it & 7);
            o[0 * 8] = (uint8_t)mU; o[1 * 8] = (uint8_t)mW; o[2 * 8] = (uint8_t)mX; o[3 * 8] = (uint8_t)mS;
            o[4 * 8] = (uint8_t)mN; o[5 * 8] = (uint8_t)mCR; o[6 * 8] = (uint8_t)(mCR | mSL); o[7 * 8] = (uint8_t)mC;
            o[8 * 8] = (uint8_t)mD; o[9 * 8] = (uint8_t)mA; o[10 * 8] = (uint8_t)mSP; o[11 * 8] = (uint8_t)mSY;
        };
        constexpr int P1_ITEMS = K_WIN / 8;
        if (tile_ascii) {
            // no byte >= 0x80 in the whole window (most tiles of English text and of source code): LUT only
            for (int it = tid; it < P1_ITEMS; it += K_THREADS) {
                const uint2 t8 = reinterpret_cast<const uint2*>(s_txt)[it];
                const uint32_t flo = (uint32_t)s_lut[t8.x & 0x7F] | ((uint32_t)s_lut[(t8.x >> 8) & 0x7F] << 8) |
                                     ((uint32_t)s_lut[(t8.x >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[t8.x >> 24] << 24);
                const uint32_t fhi = (uint32_t)s_lut[t8.y & 0x7F] | ((uint32_t)s_lut[(t8.y >> 8) & 0x7F] << 8) |
                                     ((uint32_t)s_lut[(t8.y >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[t8.y >> 24] << 24);
                uint32_t pf = __shfl_up(fhi >> 24, 1);
                if (lane == 0) pf = it > 0 ? (uint32_t)s_lut[s_txt[it * 8 - 1]] : 0u;
                emit_masks(it, flo, fhi, pf);
            }
        } else {
        // (a wavefront takes a contiguous quarter of the items, 64 per pass: what lane 0 needs from the item in front of its
        // own — the last character's features — is lane 63's of the pass before, not another wavefront's)
        constexpr int P1_WAVES = K_THREADS / 64, P1_PER_WAVE = (P1_ITEMS / P1_WAVES) & ~63;
        static_assert(P1_ITEMS - P1_WAVES * P1_PER_WAVE <= 64, "the items left over are one more pass of the last wavefront");
        uint32_t carry_st = 0x100u, carry_pf = 0;
        const int n_pass = P1_PER_WAVE / 64 + ((tid >> 6) == P1_WAVES - 1 ? 1 : 0);
        const uint8_t* const t_ascii = T->ascii_cls;  // (read once for the phase, not in front of every character)
        const uint16_t* const t_u1 = T->ucls1;
        const uint8_t* const t_u2 = T->ucls2;
        for (int q = 0; q < n_pass; ++q) {
            const int it = (tid >> 6) * P1_PER_WAVE + q * 64 + lane;
            if (it >= P1_ITEMS) break;
            const uint2 t8 = reinterpret_cast<const uint2*>(s_txt)[it];
            uint32_t flo, fhi;
            // state handed to the next lane: class features of my last character and how many continuation bytes it
            // still claims (0x100 = "unknown": this item held continuation bytes only)
            uint32_t st_out = 0;
            int lead_conts = 0;  // continuation bytes at the start of my item (they belong to the previous lane's char)
            if (!((t8.x | t8.y) & 0x80808080u)) {
                flo = (uint32_t)s_lut[t8.x & 0x7F] | ((uint32_t)s_lut[(t8.x >> 8) & 0x7F] << 8) |
                      ((uint32_t)s_lut[(t8.x >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[t8.x >> 24] << 24);
                fhi = (uint32_t)s_lut[t8.y & 0x7F] | ((uint32_t)s_lut[(t8.y >> 8) & 0x7F] << 8) |
                      ((uint32_t)s_lut[(t8.y >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[t8.y >> 24] << 24);
            } else {
                // non-ASCII.  UTF-8 is self-synchronising, so the item is classified without walking it character by
                // character: every byte gets the ASCII LUT's answer first; the lead bytes (at most four well-formed
                // multi-byte characters fit) are picked out of byte-flag words, decoded from a 16-byte register window, and
                // the two table loads of ALL of them go out together (a loop over the characters with two dependent L2
                // round trips each made this phase 80 % of the kernel on mixed-script text); their features overwrite the
                // bytes they span.  What is left (invalid leads, truncated sequences, stray continuation bytes, a fifth
                // lead) takes the general per-byte route (feature_at_v).
                const uint2 t8n = (it + 1 < K_WIN / 8) ? reinterpret_cast<const uint2*>(s_txt)[it + 1] : make_uint2(0, 0);
                const uint64_t lo8 = ((uint64_t)t8.y << 32) | t8.x;
                const uint32_t docb = (uint32_t)reinterpret_cast<const uint8_t*>(s_doc)[it] |
                                      ((it + 1 < K_WIN / 8) ? (uint32_t)reinterpret_cast<const uint8_t*>(s_doc)[it + 1] << 8 : 0u);
                const int pos0 = it * 8;
                uint64_t F = (uint64_t)((uint32_t)s_lut[t8.x & 0x7F] | ((uint32_t)s_lut[(t8.x >> 8) & 0x7F] << 8) |
                                        ((uint32_t)s_lut[(t8.x >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[(t8.x >> 24) & 0x7F] << 24)) |
                             ((uint64_t)((uint32_t)s_lut[t8.y & 0x7F] | ((uint32_t)s_lut[(t8.y >> 8) & 0x7F] << 8) |
                                         ((uint32_t)s_lut[(t8.y >> 16) & 0x7F] << 16) | ((uint32_t)s_lut[(t8.y >> 24) & 0x7F] << 24)) << 32);
                constexpr uint64_t HB = 0x8080808080808080ull;  // flag words: bit 8k+7 <-> byte k
                const uint64_t na = lo8 & HB;                   // non-ASCII bytes
                const uint64_t ct = na & ~(lo8 << 1);           // continuation bytes (10xxxxxx)
                uint64_t ld = na & (lo8 << 1);                  // bytes >= 0xC0
                auto bytes_below = [](int nb) { return nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull); };
                {   // continuation bytes at the very start belong to the previous lane's character (resolved below); one
                    // that starts a document is a character of its own
                    const int lc0 = td_ctz64(~ct & HB) >> 3, lcd = (int)td_ctz32(docb | 0x100u);
                    lead_conts = lc0 < lcd ? lc0 : lcd;
                }
                uint32_t kj[4], ndj[4], cpj[4], uj[4];
                bool tbj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool have = ld != 0;
                    const int k = have ? td_ctz64(ld) >> 3 : 0;
                    ld &= ld - 1ull;
#ifndef TD_DECODE_BRANCHY
                    // bytes k .. k+3 of the 16-byte window as ONE dword (32-bit funnel shift over the pair of dwords the lead lies in), and
                    // everything below WITHOUT a branch: written with && and ?: the compiler turned the three lengths into three divergent
                    // blocks per character slot, and mixed-script text has all three in every wavefront
                    const uint32_t dlo = k < 4 ? t8.x : t8.y, dhi = k < 4 ? t8.y : t8n.x;
                    const uint32_t x = __funnelshift_r(dlo, dhi, 8u * ((uint32_t)k & 3u));
                    const uint32_t b = x & 0xFFu;
                    const uint32_t need = utf8_declared_len(b) - 1u;  // 0 .. 3
                    const int pos = pos0 + k;
                    const uint32_t cm = (0xC0C0C000u >> (8u * (3u - need))) & 0xC0C0C000u;  // the two top bits of the `need` continuation bytes
                    const bool ok = have & (need > 0) & (pos + (int)need < (int)src.hi) & (pos >= (int)src.lo) & (((docb >> (k + 1)) & ((1u << need) - 1u)) == 0u) &
                                    ((x & cm) == (cm & 0x80808000u));
                    const uint32_t c1 = (x >> 8) & 0x3Fu, c2 = (x >> 16) & 0x3Fu, c3 = (x >> 24) & 0x3Fu;
                    uint32_t c = ((b & (0x3Fu >> need)) << 6) | c1;
                    c = need >= 2u ? (c << 6) | c2 : c;
                    c = need >= 3u ? (c << 6) | c3 : c;
#else
                    const uint64_t hi8 = ((uint64_t)t8n.y << 32) | t8n.x;
                    const uint64_t w = (lo8 >> (8 * k)) | ((hi8 << 8) << (56 - 8 * k));  // bytes k .. k+7
                    const uint32_t b = (uint32_t)w & 0xFFu, c1 = (uint32_t)(w >> 8) & 0xFFu, c2 = (uint32_t)(w >> 16) & 0xFFu, c3 = (uint32_t)(w >> 24) & 0xFFu;
                    const uint32_t need = utf8_declared_len(b) - 1u;
                    const int pos = pos0 + k;
                    const bool ok = have && need > 0 && pos + (int)need < (int)src.hi && pos >= (int)src.lo && !((docb >> (k + 1)) & ((1u << need) - 1u)) &&
                                    (c1 & 0xC0) == 0x80 && (need < 2 || (c2 & 0xC0) == 0x80) && (need < 3 || (c3 & 0xC0) == 0x80);
                    const uint32_t c = (need == 1) ? ((b & 0x1F) << 6) | (c1 & 0x3F)
                                     : (need == 2) ? ((b & 0x0F) << 12) | ((c1 & 0x3F) << 6) | (c2 & 0x3F)
                                                   : ((b & 0x07) << 18) | ((c1 & 0x3F) << 12) | ((c2 & 0x3F) << 6) | (c3 & 0x3F);
#endif
                    tbj[j] = ok && c <= 0x10FFFFu && !(c >= 0xD800u && c <= 0xDFFFu);  // (class_of_cp: everything else is C_OTHER)
                    cpj[j] = tbj[j] ? c : 0u;
                    kj[j] = (uint32_t)k;
                    ndj[j] = ok ? need : 0u;
                    uj[j] = t_u1[cpj[j] >> 8];  // (unconditional, independent loads: they leave back to back)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) uj[j] = t_u2[uj[j] * 256u + (cpj[j] & 255u)];
                uint64_t cov = 0;  // flags of the bytes the well-formed characters span
                int remaining = 0, last_lead = -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ndj[j]) {
                        const uint32_t f = tbj[j] ? (uint32_t)s_fcls[uj[j] & 15u] : (uint32_t)FB_X;
                        const uint32_t bm = ndj[j] == 3u ? 0xFFFFFFFFu : ((1u << (8u * (ndj[j] + 1u))) - 1u);
                        const uint32_t pat = ((f * 0x01010101u) | 0x80808000u) & bm;  // the lead, then its continuation bytes (FB_C)
                        const int sh = 8 * (int)kj[j];
                        F = (F & ~((uint64_t)bm << sh)) | ((uint64_t)pat << sh);
                        cov |= (uint64_t)(bm & 0x80808080u) << sh;
                        last_lead = (int)kj[j];
                        remaining = ((int)kj[j] + (int)ndj[j] > 7) ? (int)kj[j] + (int)ndj[j] - 7 : 0;
                    }
                }
                const uint64_t lcm = bytes_below(lead_conts);
                for (uint64_t gen = na & ~cov & ~lcm; gen; gen &= gen - 1ull) {  // (rare)
                    const int k = td_ctz64(gen) >> 3;
                    const uint32_t f = feature_at_v(t_ascii, t_u1, t_u2, s_txt, s_doc, (int)src.lo, (int)src.hi, pos0 + k);
                    F = (F & ~(0xFFull << (8 * k))) | ((uint64_t)f << (8 * k));
                }
                F &= ~lcm;
                flo = (uint32_t)F;
                fhi = (uint32_t)(F >> 32);
                // the item's last character: its features and the continuation bytes it claims from the next item
                const uint64_t starts = HB & ~(cov & ct) & ~lcm;
                if (starts) {
                    const int last = (63 - (int)__builtin_clzll(starts)) >> 3;
                    st_out = ((uint32_t)(F >> (8 * last)) & 0x7Fu) | ((uint32_t)(last == last_lead ? remaining : 0) << 9);
                } else {
                    st_out = 0x100u;
                }
            }
            {   // continuation bytes at the start of the item: the previous lane's last character claims them
                uint32_t st_in = __shfl_up(st_out, 1);
                if (lane == 0) st_in = carry_st;
                carry_st = (uint32_t)__builtin_amdgcn_readlane((int)st_out, 63);
                if (lead_conts) {
                    const bool known = !(st_in & 0x100u);
                    const int claim = (int)(st_in >> 9);
                    for (int k = 0; k < lead_conts; ++k) {
                        const uint32_t f = (known && k < claim) ? ((st_in & 0xFFu) | FB_C) : feature_at_v(t_ascii, t_u1, t_u2, s_txt, s_doc, (int)src.lo, (int)src.hi, it * 8 + k);
                        if (k < 4) flo |= f << (8 * k); else fhi |= f << (8 * (k - 4));
                    }
                }
            }
            // feature byte of the byte in front of my 8 (for the sync predicate)
            uint32_t pf = __shfl_up(fhi >> 24, 1);
            if (lane == 0 && q > 0) pf = carry_pf;
            carry_pf = (uint32_t)__builtin_amdgcn_readlane((int)(fhi >> 24), 63);
            if (lane == 0 && q == 0) {
                pf = 0;
                if (it > 0) {
                    const uint32_t pb = s_txt[it * 8 - 1];
                    pf = (pb < 0x80) ? (uint32_t)s_lut[pb] : feature_at_v(t_ascii, t_u1, t_u2, s_txt, s_doc, (int)src.lo, (int)src.hi, it * 8 - 1);
                }
            }
            emit_masks(it, flo, fhi, pf);
        }
        }
        __syncthreads();
        FZ_TICK(1)
        if (tid == 0) s_nonascii = 0;
        if (TD_STOP(12)) continue;

        // ---- phase 2: piece boundaries.
        //  (a) whole-word rules: every lane takes the 32 bytes of its stride + 32 bytes of look-ahead as 64-bit masks and
        //      proves with carry arithmetic (split_unresolved_heads) which synchronisation points ("heads") are followed by
        //      exactly one piece up to the next one.  START = the synchronisation points themselves; on English text 98 %
        //      of the heads end here.
        //  (b) the other heads go on a list in LDS (+ the last sync point at or before the tile start, whose pieces lead
        //      into the tile, + the tile's last head, whose pieces find the next tile's first piece start: tile_carry).
        //      Wavefronts draw 64 heads at a time; a lane walks its head's pieces with the branch-free matcher until it
        //      lands on a synchronisation point, then draws the next head: no lane waits for the busiest chunk.
        //  (c) what the branch-free matcher leaves open (1 % of the pieces) is put aside and matched by the general
        //      matcher on the mask words in LDS afterwards; pieces that leave the window go to the td_split_far_* kernels
        //      ("this piece start is known, its end is not"), and a tile without a sync point in its left halo is flagged
        //      ("first piece start unknown": it lies inside a piece longer than the halo) and gets it through tile_carry.
        {
            constexpr bool FAST = !(PV & (PV_GPT2 | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS | PV_WS_EOS_FIRST));
            const uint32_t* s_mask32 = reinterpret_cast<const uint32_t*>(s_mask);
            auto dword_at = [](int j) { return (j >> 1) * (MK_COUNT * 2) + (j & 1); };  // mask 0 of dword j; mask k: + 2 k
            auto defer = [&](int64_t g) {
                if (FUSED) s_defer = 1;
                const uint32_t q = atomicAdd(a->slow_count, 1u);
                if (q < a->slow_cap) a->slow_list[q] = g;
                else raise(a, TD_E_SCRATCH, g);
            };
            auto mark = [&](int q) { atomicOr(&s_start[q >> 5], 1u << (q & 31)); };
            auto sync_at = [&](int q) { return (s_mask32[dword_at(q >> 5) + 2 * MK_SYNC] >> (q & 31)) & 1u; };
            auto crossed = [&](int e) {  // first piece start of the next tile
                if (FUSED) s_cross = e;
                if (tile + 1 < a->n_stiles) a->tile_carry[tile + 1] = wg0 + e;
            };
            // general matcher from the piece start p (marked) to the end of its chain
            auto walk = [&](int p) {
                for (;;) {
                    const int e = scan_piece_lds<PV>(s_mask, s_txt, p);
                    if (e < 0) { if (p >= K_HL) defer(wg0 + p); return; }
                    if (e >= tile_hi) { crossed(e); return; }
                    if (sync_at(e)) return;
                    if (e >= K_HL) mark(e);
                    p = e;
                }
            };
            // (a)
            const int b0 = K_HL + tid * 32;
            uint32_t mine = 0;  // unresolved heads of my stride that found no room on the list
            uint32_t extra_hi = 0;
            {
                const uint32_t* lo = s_mask32 + dword_at(b0 >> 5);
                const uint32_t* hi = s_mask32 + dword_at((b0 >> 5) + 1);
                BitWin wv;
#pragma unroll
                for (int k = 0; k < MK_COUNT; ++k) wv.m[k] = ((uint64_t)hi[2 * k] << 32) | lo[2 * k];
                uint32_t sy = (uint32_t)wv.m[MK_SYNC];
                const int room = tile_hi - b0;
                if (room < 32) sy = room <= 0 ? 0u : sy & ((1u << room) - 1u);
                uint32_t st0 = sy, un;
                if (wv.m[MK_N] != 0ull) {  // digits in my window (no lane of a wavefront of plain prose comes here: the rules twice in the code, once in its time)
                    uint64_t extra = 0;  // piece starts inside the regions of my heads that the rules place too (td_common.h)
                    un = (uint32_t)split_unresolved_heads(wv, PV, &extra) & sy;
                    extra &= (uint64_t)sy << 3;  // (behind MY heads, inside the tile)
                    extra &= room >= 64 ? ~0ull : room <= 0 ? 0ull : ((1ull << room) - 1ull);
                    st0 |= (uint32_t)extra;
                    extra_hi = (uint32_t)(extra >> 32);  // (the next lane's word: behind the barrier)
                } else {
                    un = (uint32_t)split_unresolved_heads(wv, PV) & sy;
                }
                s_start[b0 >> 5] = st0;
                if (sy) atomicMax(&s_last, b0 + 31 - (int)__clz(sy));
                const uint32_t cnt = __popc(un);
                const uint32_t incl = wave_incl_scan(cnt, lane);
                uint32_t base = 0;
                if (lane == 63 && incl) base = atomicAdd(&s_nh, incl);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
                uint32_t idx = base + incl - cnt;
                while (un) {
                    const int bit = __ffs(un) - 1;
                    un &= un - 1u;
                    if (idx < (uint32_t)KS_HCAP) s_heads[idx] = (uint16_t)(b0 + bit); else mine |= 1u << bit;
                    ++idx;
                }
            }
            __syncthreads();
            FZ_TICK(2)
            if (extra_hi) atomicOr(&s_start[(b0 >> 5) + 1], extra_hi);
            if (TD_STOP(13)) continue;
            // (b)
            {
                const uint32_t nh = s_nh < (uint32_t)KS_HCAP ? s_nh : (uint32_t)KS_HCAP;
                int p = -1;
                // two heads are not on the list (a second barrier and a serial stretch of lane 0 put them there): lane 0 starts
                // with the last provable sync point before the tile start (window bytes 4..K_HL) unless the tile starts on one,
                // lane 1 with the tile's last head
                if (tid == 0) {
                    static_assert(K_HL % 64 == 0 && K_HL >= 64, "left halo = whole mask words");
                    if (!(s_mask[(K_HL / 64) * MK_COUNT + MK_SYNC] & 1ull)) {
                        int sp = -1;
                        for (int w = K_HL / 64 - 1; w >= 0 && sp < 0; --w) {
                            uint64_t m = s_mask[w * MK_COUNT + MK_SYNC];
                            if (w == 0) m &= ~0xFull;
                            if (m) sp = w * 64 + td_top64(m) - 1;
                        }
                        if (sp < 0) { a->tile_flag[tile] = 1; atomicAdd(a->far_count, 1u); if (FUSED) s_defer = 1; }
                        else p = sp;
                    }
                }
                if (tid == 1 && s_last >= 0) p = s_last;
                bool more = true;  // (uniform) the list may still hold heads
                for (;;) {
                    const uint64_t nb = __ballot(p < 0);
                    if (nb && more) {
                        uint32_t base = 0;
                        if (lane == __ffsll((unsigned long long)nb) - 1) base = atomicAdd(&s_cur, (uint32_t)__popcll(nb));
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((unsigned long long)nb) - 1);
                        more = base + (uint32_t)__popcll(nb) < nh;
                        if (p < 0) {
                            const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(nb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nb, 0u));
                            if (idx < nh) p = s_heads[idx];
                        }
                    }
                    if (!__ballot(p >= 0)) break;
                    if (p >= 0) {
                        const int sh = p & 31;
                        const uint32_t* lo = s_mask32 + dword_at(p >> 5);
                        const uint32_t* hi = s_mask32 + dword_at((p >> 5) + 1);
                        BitWin32 v;  // 32-bit view of every mask with bit 0 = the piece start
#pragma unroll
                        for (int k = 0; k < MK_COUNT; ++k) v.m[k] = __funnelshift_r(lo[2 * k], hi[2 * k], (uint32_t)sh);
                        const int avail = (K_LIM - p < 32) ? (K_LIM - p) : 32;
                        auto bytes = [&](int q) { return (uint32_t)s_txt[p + q]; };
                        const int r = FAST ? scan_piece_fast32(v, avail, bytes, PV) : scan_piece_p(WinP32(v, avail), bytes, PV);
                        if (r < 0) {  // (c)
                            const uint32_t ci = atomicAdd(&s_ncold, 1u);
                            if (ci < (uint32_t)KS_CCAP) s_cold[ci] = (uint16_t)p; else walk(p);
                            p = -1;
                        } else {
                            const int e = p + r;
                            bool stop = true;
                            if (e >= tile_hi) crossed(e);
                            else {
                                stop = (r < 32) ? ((v.m[MK_SYNC] >> r) & 1u) : sync_at(e);
                                if (!stop && e >= K_HL) mark(e);
                            }
                            p = stop ? -1 : e;
                        }
                    }
                }
            }
            __syncthreads();
            {
                const uint32_t nc = s_ncold < (uint32_t)KS_CCAP ? s_ncold : (uint32_t)KS_CCAP;
                for (uint32_t i = tid; i < nc; i += K_THREADS) walk(s_cold[i]);
                while (mine) {
                    const int bit = __ffs(mine) - 1;
                    mine &= mine - 1u;
                    walk(b0 + bit);
                }
            }
        }
        __syncthreads();
        FZ_TICK(3)
        // ---- publish the tile's own START bits (tile-aligned words: no other workgroup writes them) ----
        if (tid < KS_TILE / 32) {
            const int64_t g = tile_g0 + (int64_t)tid * 32;
            if (g < a->n) {
                uint32_t v = s_start[K_HL / 32 + tid];
                if (g + 32 > a->n) v &= (1u << (int)(a->n - g)) - 1u;
                TD_NT_STORE(v, &a->startbits[(tile_g0 >> 5) + tid]);
            }
        }
        if constexpr (DRAW) {
            // (thread 0 waits for its draw right here and stores it for the iteration after the next one; keeping it in a
            // register until the next tile's text is waited for was no faster and cost spills)
            if constexpr (!DIRECT) {
                next_tile = __builtin_amdgcn_readfirstlane(s_next[par]);  // (written one iteration ago; uniform: a scalar register)
                if (tid == 0) s_next[par ^ 1] = (int)(2u * gridDim.x + atomicAdd(a->tile_draw, 1u));
                par = __builtin_amdgcn_readfirstlane(par ^ 1);
            } else {
                next_tile = __builtin_amdgcn_readfirstlane(s_next[0]);  // (drawn at the top of this iteration; uniform: a scalar register)
            }
            if (next_tile < a->n_stiles) load_window((int64_t)next_tile * KS_TILE - K_HL);
        } else if (FUSED && tile + (int)gridDim.x < a->n_stiles) {
            load_window(wg0 + (int64_t)gridDim.x * KS_TILE);
        }
        }  // (the boundary phases)
        if constexpr (FUSED) {
            // ================= token phases: the tile's pieces -> slots of the staging regions (see the note above) =========
            constexpr int NT4 = KS_TILE / K_TILE;  // token tiles per pre-tokenizer tile
            static_assert(NT4 == 2 && KS_CHUNK == 32, "two token tiles, a 32-bit START word per lane");
            const int cross = s_cross;
            const bool tile_deferred = s_defer != 0 || cross < 0;  // (uniform: both are final since the barrier above)
            uint16_t* const s_plist = reinterpret_cast<uint16_t*>(s_R + FZ_R_PLIST);
            uint16_t* const s_coldk = reinterpret_cast<uint16_t*>(s_R + FZ_R_COLDK);
            uint16_t* const s_pb = reinterpret_cast<uint16_t*>(s_R + FZ_R_PB);
            uint32_t* const s_sm = reinterpret_cast<uint32_t*>(s_R + FZ_R_SM);
            const int wv = tid >> 6;
            const int tile4 = tile * NT4;            // first token tile of the pair
            const bool two = K_HL + K_TILE < tile_hi;  // the second one exists
            // (the documents that start in the tile are consecutive from its first one, fd0 / fd1: prefetched with the window)
            // ---- dense list of the tile's piece starts ----
            const int b0 = K_HL + tid * KS_CHUNK;
            const uint32_t smask0 = s_start[b0 >> 5];  // START bits of my 32 bytes (bits at and behind the tile end are zero)
            // (the region is free: nobody has touched the masks and the lists since the barrier in front of the START bits' way out)
            const uint32_t cnt = __popc(smask0);
            const uint32_t incl = wave_incl_scan(cnt, lane);
            if (lane == 63) s_wave[wv] = incl;
            if (tid == 0) { s_hflags[0] = 0; s_hflags[1] = 0; s_nrec[0] = 0; s_nrec[1] = 0; s_ncoldp = 0; }
            __syncthreads();
            const uint32_t w0s = s_wave[0], w1s = s_wave[1], w2s = s_wave[2], w3s = s_wave[3];
            const uint32_t np = w0s + w1s + w2s + w3s, n0 = w0s + w1s;  // pieces of the tile / of its first token tile (lanes 0..127)
            const uint32_t pbase = (wv > 0 ? w0s : 0u) + (wv > 1 ? w1s : 0u) + (wv > 2 ? w2s : 0u) + incl - cnt;
            // direct placement: this tile's slots go to the workgroup's slab (place_prev above takes them from there half an
            // iteration later) unless the chain of counts is known to be broken already
            const bool cand = DIRECT && !s_brk;  // (uniform; s_brk: written in front of the barriers of the boundary phases)
            auto break_chain = [&]() {  // (thread 0) this tile's id count is not settled inside the loop
                __hip_atomic_store(&a->tile_state[tile], TS_BROKEN << 62, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_brk = 1u;  // (this workgroup's later tiles take the staged path from the start; the others find out when they look back)
            };
            if (tile_deferred || np > (uint32_t)FZ_NPC) {  // (uniform) td_probe_tiles takes these token tiles, behind the far kernels
                if (tid == 0) {
                    const uint32_t at = atomicAdd(a->deferred_count, two ? 2u : 1u);
                    a->deferred_list[at] = (uint32_t)tile4;
                    if (two) a->deferred_list[at + 1] = (uint32_t)tile4 + 1u;
                    if (DIRECT) break_chain();
                }
            } else {
                {
                    uint32_t k = pbase, m = smask0;
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        s_plist[k++] = (uint16_t)(b0 + b);
                    }
                }
                s_pb[tid] = (uint16_t)pbase;
                s_sm[tid] = smask0;
                if (tid == 0) s_plist[np] = (uint16_t)cross;  // end of the tile's last piece: where the boundary scan crossed the tile end
                const uint64_t fdd = fd0 != 0xFFFFFFFFu ? fd0 : fd1;
                // (parked now: what is kept in a register to the end of the tile is spilled, and a reload from scratch memory behind the
                // slot stores waits for every one of them)
                if (cand && tid == 0) { s_pd[ring][PD_FDD] = (uint32_t)fdd; s_pd[ring][PD_TILE] = (uint32_t)tile; s_pd[ring][PD_NP] = np; s_pd[ring][PD_N0] = n0; }
                const int64_t dmine = (int64_t)fdd + tid;
                const int64_t dpos = (fdd != 0xFFFFFFFFu && dmine < a->n_docs) ? a->doc_offsets[dmine] : a->n;  // (used in the last phase)
                __syncthreads();
                FZ_TICK(4)
                // ---- lookups.  Piece k -> lane k mod 256, FZ_PB pieces of a lane at a time: first all their keys and loads
                //      (three dwords cut out of LDS with funnel shifts, masked by length; ONE 16-byte load of the first slot of
                //      the exact-key table each), then the compares and the slot stores.  An empty slot is a miss; a slot held
                //      by another key and longer pieces are put aside (probe_piece_cold behind the loop). ----
                uint32_t* const slab = a->slab + ((size_t)blockIdx.x * SLAB_RING + ring) * SLAB_WORDS;
                uint32_t* const dst0 = cand ? slab : a->stage + (size_t)tile4 * K_STAGE;
                uint32_t* const dst1 = cand ? slab : dst0 + K_STAGE - n0;  // (slot k >= n0 is slot k - n0 of the second token tile)
                auto note_miss = [&](uint32_t k, uint32_t res) {  // res: TOK_MISS | position in ITS token tile << 7 | length
                    const uint32_t h = k >= n0 ? 1u : 0u;
                    const uint32_t mi = atomicAdd(&s_nrec[h], 1u);
                    if (mi < (uint32_t)K_MISS_LISTED_MAX) s_rec[h][mi] = ((k - (h ? n0 : 0u)) << 19) | (res & 0x7FFFFu);
                };
                auto miss_marker = [&](int i, uint32_t len) { return TOK_MISS | ((uint32_t)((i - K_HL) & (K_TILE - 1)) << 7) | len; };
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef const u32x4 __attribute__((address_space(1)))* gslot_t;  // (global loads, not flat ones)
                gslot_t const p12 = (gslot_t)(uintptr_t)T->piece12_slots;
                const uint32_t p12_mask = T->piece12_mask;
                const bool fastpath = a->use_fastpath != 0;
                for (uint32_t kb = tid; kb < np; kb += K_THREADS * FZ_PB) {
                    u32x4 sl[FZ_PB];
                    uint32_t kk0[FZ_PB], kk1[FZ_PB], kk2[FZ_PB], il[FZ_PB];
#pragma unroll
                    for (int q = 0; q < FZ_PB; ++q) {
                        const uint32_t k = kb + (uint32_t)q * K_THREADS;
                        const bool on = k < np;
                        const int i = on ? s_plist[k] : K_HL;
                        const uint32_t len = on ? (uint32_t)s_plist[k + 1] - (uint32_t)i : 0u;
                        const uint32_t* wp = reinterpret_cast<const uint32_t*>(s_txt) + (i >> 2);
                        const uint32_t sh = (i & 3) * 8;
                        const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                        const uint4 km = reinterpret_cast<const uint4*>(s_kmask)[len < P12_MAXLEN ? len : P12_MAXLEN];
                        kk0[q] = __funnelshift_r(w0, w1, sh) & km.x;
                        kk1[q] = __funnelshift_r(w1, w2, sh) & km.y;
                        kk2[q] = __funnelshift_r(w2, w3, sh) & km.z;
                        il[q] = ((uint32_t)i << 16) | (len < 0xFFFFu ? len : 0xFFFFu);
                        if (on) sl[q] = p12[hash_piece12(kk0[q], kk1[q], kk2[q], len) & p12_mask];
                    }
#pragma unroll
                    for (int q = 0; q < FZ_PB; ++q) {
                        const uint32_t k = kb + (uint32_t)q * K_THREADS;
                        if (k < np) {
                            const int i = (int)(il[q] >> 16);
                            const uint32_t len = il[q] & 0xFFFFu;
                            const u32x4 e = sl[q];
                            if (len <= P12_MAXLEN && fastpath &&
                                (e.w == 0u || (e.x == kk0[q] && e.y == kk1[q] && e.z == kk2[q] && (e.w >> 24) == (0x80u | len)))) {
                                const bool miss = e.w == 0u;  // empty slot: not a token (a single byte that is no token is an error, not a merge)
                                const uint32_t res = miss ? miss_marker(i, len) : (e.w & 0x1FFFFFu);
                                if (miss) {
                                    if (len == 1) raise(a, TD_E_UNKNOWN_BYTE, wg0 + i);
                                    note_miss(k, res);
                                }
                                (k < n0 ? dst0 : dst1)[k] = res;
                            } else {
                                const uint32_t j = atomicAdd(&s_ncoldp, 1u);
                                if (j < (uint32_t)FZ_CCAP) s_coldk[j] = (uint16_t)k;  // (more: the tile is deferred after all, below)
                            }
                        }
                    }
                }
                __syncthreads();
                FZ_TICK(5)
                const uint32_t ncold = s_ncoldp;
                const bool cold_overflow = ncold > (uint32_t)FZ_CCAP;  // (uniform)
                if (!cold_overflow) {
                    for (uint32_t c = tid; c < ncold; c += K_THREADS) {
                        const uint32_t k = s_coldk[c];
                        const int i = s_plist[k];
                        const uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)i;
                        const uint32_t h = k >= n0 ? 1u : 0u;
                        uint32_t res = probe_piece_cold<true>(a, T, s_txt, T->byte_id, &s_hflags[h], wg0, i, len);
                        if ((res & 0xC0000000u) == TOK_MISS) { res = miss_marker(i, len); note_miss(k, res); }
                        (h ? dst1 : dst0)[k] = res;
                    }
                }
                __syncthreads();
                FZ_TICK(6)
                // (uniform) the tile's id count can be settled right here: every piece is a token or one of a few missed pieces
                const bool resolved = cand && !cold_overflow && !(s_hflags[0] | s_hflags[1]) && s_nrec[0] <= (uint32_t)K_MISS_LISTED_MAX &&
                                      s_nrec[1] <= (uint32_t)K_MISS_LISTED_MAX;
                if (cold_overflow) {  // more pieces for the long route than the list holds: td_probe_tiles does the tile over
                    if (tid == 0) {
                        const uint32_t at = atomicAdd(a->deferred_count, two ? 2u : 1u);
                        a->deferred_list[at] = (uint32_t)tile4;
                        if (two) a->deferred_list[at + 1] = (uint32_t)tile4 + 1u;
                        if (DIRECT) break_chain();
                    }
                } else if (resolved) {
                    FZ_TICK(12)
                    // ---- the few missed pieces of the tile are merged here, one lane each (the first wavefront; the others go on
                    //      to the documents): ids to the slab behind the slots, count and extras to s_pv ----
                    if (tid < 64) {
                        const uint32_t nr0 = s_nrec[0], nr1 = s_nrec[1], nm = nr0 + nr1;
                        uint32_t ext0 = 0, ext1 = 0;
                        if (nm) {  // (rare on plain text: out of line)
                            static_assert(2 * SLAB_MAX_MISSES * 4 * MG_UNIT * 4 <= FZ_R_COLDK, "keys + ids of the merged pieces fit the (dead) piece list");
                            uint32_t rec = 0, hh = 0;
                            if ((uint32_t)lane < nm) {
                                hh = (uint32_t)lane >= nr0 ? 1u : 0u;
                                rec = s_rec[hh][(uint32_t)lane - (hh ? nr0 : 0u)];
                            }
                            const uint32_t nt = fz_merge_piece(a->Tp, a->text, a->n, reinterpret_cast<uint32_t*>(s_R), slab, rec, hh, n0, tile_g0, a->err, a->err_pos);
                            const uint32_t e0 = wave_incl_scan((nt && !hh) ? nt - 1u : 0u, lane), e1 = wave_incl_scan((nt && hh) ? nt - 1u : 0u, lane);
                            ext0 = (uint32_t)__builtin_amdgcn_readlane((int)e0, 63);
                            ext1 = (uint32_t)__builtin_amdgcn_readlane((int)e1, 63);
                        }
                        FZ_TICK(13)
                        if (tid == 0) {
                            const uint32_t count = s_pd[ring][PD_NP] + ext0 + ext1;
                            s_pd[ring][PD_KIND] = 1u; s_pd[ring][PD_COUNT] = count; s_pd[ring][PD_NMISS] = nm; s_pd[ring][PD_EXT0] = ext0;
                            __hip_atomic_store(&a->tile_state[s_pd[ring][PD_TILE]], (TS_AGG << 62) | (unsigned long long)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    FZ_TICK(7)
                    {   // the slot of every document that starts in this tile (as below)
                        const int64_t tile_end_g = wg0 + tile_hi;
                        auto slot_of = [&](int64_t p) {
                            const int lp = (int)(p - tile_g0);
                            const uint32_t k = (uint32_t)s_pb[lp >> 5] + __popc(s_sm[lp >> 5] & ((1u << (lp & 31)) - 1u));
                            return k >= n0 ? k - n0 : k;
                        };
                        if (dpos < tile_end_g) {
                            a->doc_slot[dmine] = slot_of(dpos);
                            for (int64_t d = dmine + K_THREADS; d < a->n_docs; d += K_THREADS) {
                                const int64_t p = a->doc_offsets[d];
                                if (p >= tile_end_g) break;
                                a->doc_slot[d] = slot_of(p);
                            }
                        }
                    }
                    FZ_TICK(9)
                } else {
                if (DIRECT && tid == 0) {  // a tile whose count is settled by the kernels behind this one: the chain ends here
                    break_chain();
                    if (cand) { s_pd[ring][PD_KIND] = 2u; s_pd[ring][PD_NMISS] = 0; }
                }
                // ---- per token tile: slot count + flags; its missed pieces to the global lists (few) or the tile flagged (many) ----
                if (tid < 64) {  // (first wavefront: program order between its lanes' LDS accesses; td_probe_tiles has the notes)
#pragma unroll
                    for (uint32_t h = 0; h < 2; ++h) {
                        const uint32_t nr = s_nrec[h], np0 = s_npend;
                        const bool listed = nr && nr <= (uint32_t)K_MISS_LISTED_MAX;
                        if (listed && (uint32_t)tid < nr) s_pend[np0 + tid] = ((unsigned long long)((uint32_t)tile4 + h) << 32) | s_rec[h][tid];
                        const uint32_t np1 = listed ? np0 + nr : np0;
                        wave_sync_lds();
                        if (np1 > 48u - (uint32_t)K_MISS_LISTED_MAX) {  // (s_pend holds 48)
                            append_pending(np1);
                            wave_sync_lds();
                            if (tid == 0) s_npend = 0;
                        } else if (tid == 0) {
                            s_npend = np1;
                        }
                        wave_sync_lds();
                    }
                    if (tid == 0) {
#pragma unroll
                        for (uint32_t h = 0; h < 2; ++h) {
                            if (h == 1 && !two) break;
                            uint32_t fl = s_hflags[h];
                            const uint32_t nr = s_nrec[h];
                            if (nr > (uint32_t)K_MISS_LISTED_MAX) {
                                fl |= TILE_HAS_MISS;
                                uint32_t nf = s_nflagl;
                                s_flagl[nf++] = (uint32_t)tile4 + h;
                                if (nf == 32u) {
                                    const uint32_t at = atomicAdd(a->flagged_count, 32u);
                                    for (uint32_t q = 0; q < 32u; ++q) a->flagged_list[at + q] = s_flagl[q];
                                    nf = 0;
                                }
                                s_nflagl = nf;
                            } else if (nr) {
                                fl |= TILE_MISS_LISTED;
                            }
                            a->tile_count[tile4 + h] = (h ? np - n0 : n0) | fl;
                        }
                    }
                }
                FZ_TICK(7)
                {   // the slot of every document that starts in this tile
                    const int64_t tile_end_g = wg0 + tile_hi;
                    auto slot_of = [&](int64_t p) {
                        const int lp = (int)(p - tile_g0);
                        const uint32_t k = (uint32_t)s_pb[lp >> 5] + __popc(s_sm[lp >> 5] & ((1u << (lp & 31)) - 1u));
                        return k >= n0 ? k - n0 : k;
                    };
                    if (dpos < tile_end_g) {
                        a->doc_slot[dmine] = slot_of(dpos);
                        for (int64_t d = dmine + K_THREADS; d < a->n_docs; d += K_THREADS) {  // more than 256 documents start in this tile
                            const int64_t p = a->doc_offsets[d];
                            if (p >= tile_end_g) break;
                            a->doc_slot[d] = slot_of(p);
                        }
                    }
                }
                FZ_TICK(9)
                }
            }
        }
        __syncthreads();
        ring = __builtin_amdgcn_readfirstlane(ring + 1 == SLAB_RING ? 0 : ring + 1);
        FZ_TICK(10)
#ifdef TD_FUSED_TIMING
        ++n_tiles_done;
#endif
    }
#ifdef TD_FUSED_TIMING
    if (FUSED && tid == 0 && (blockIdx.x % 211) == 0)
        printf("fused wg %d: %llu tiles, total %llu cycles | stage %llu masks %llu rules %llu heads %llu | list %llu probe %llu cold %llu bookkeeping %llu lookback %llu docs %llu tail %llu place %llu | t12 %llu t13 %llu\n",
               (int)blockIdx.x, n_tiles_done, (unsigned long long)(__builtin_readcyclecounter() - t_total0), tt[0], tt[1], tt[2], tt[3], tt[4], tt[5], tt[6], tt[7],
               tt[8], tt[9], tt[10], tt[11], tt[12], tt[13]);
#endif
    if constexpr (FUSED) {
        if (DIRECT && tid == 0 && s_stat[0]) atomicAdd(a->direct_tiles, s_stat[0]);
        if (DIRECT && tid == 0 && s_stat[1]) atomicAdd(a->direct_tiles + 1, s_stat[1]);
        if (tid < 64) append_pending(s_npend);  // (what is still waiting in LDS)
        if (tid == 0 && s_nflagl) {
            const uint32_t nf = s_nflagl, at = atomicAdd(a->flagged_count, nf);
            for (uint32_t q = 0; q < nf; ++q) a->flagged_list[at + q] = s_flagl[q];
        }
    }
}

// ------------------------------------------------------------------ td_split_far_* ----------
// What the LDS-window pre-tokenizer could not decide: pieces longer than its window.  A WAVEFRONT per item runs the very
// same matcher (scan_piece_p) on a mask provider that reads the text from HBM: a run search classifies 256 bytes per step
// (four per lane) and ballots the class predicate, so a piece of a megabyte is a few thousand steps, not a million —
// round 1's form (one LANE per item, the byte scanner, and for every tile inside a long piece a walk back to the piece
// start) was quadratic in the piece length.
//   td_split_far_pieces  list of piece starts whose end the window could not see: scan that piece, go on marking piece
//                        starts to the end of its tile (or to where a lane of the fast kernel started), tell the tiles
//                        that begin inside what was scanned their first piece start (tile_carry);
//   td_split_far_tiles   chains of flagged tiles (no sync point in the left halo: lane 0 of the fast kernel did not run):
//                        from tile_carry of the chain's first tile, piece by piece through the flagged tiles.
struct WaveMaskP {  // mask provider over HBM for scan_piece_p; every method is executed by the whole wavefront, uniformly
    const Tables& T;
    const uint8_t* text;
    const uint32_t* docbits;
    int64_t n, g;  // text length; global position of the piece start (provider position 0)
    int o, lim;
    int lane;
    __device__ __forceinline__ uint32_t cf_at(int64_t pos) const {  // class + flags of the byte at global position pos
        if (pos >= n) return F_DOC;
        GlobSrc src;
        src.text = text; src.docbits = docbits; src.lo = 0; src.hi = n;
        uint32_t v = classify_at(T, src, pos);
        if (src.doc(pos)) v |= F_DOC;
        return v;
    }
    __device__ __forceinline__ bool pred(int k, int64_t pos, bool mask_eos) const {  // bit of mask k at pos (end-of-subject bits after o cleared)
        const uint32_t v = cf_at(pos);
        if (mask_eos && pos > g && (v & F_DOC)) return false;
        return (mask_bits_of(0, v) >> k) & 1u;
    }
    __device__ __forceinline__ bool bit(int k, int i) const { return pred(k, g + i, false); }
    __device__ __forceinline__ bool ebit(int i) const { return i > 0 && (cf_at(g + i) & F_DOC); }
    template <class F>
    __device__ __forceinline__ int first_clear(const F& f, int from) const {  // first i >= from with !f(g + i), at most lim
        int64_t w = g + from;
        for (;;) {
            uint64_t b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = __ballot(f(w + 64 * q + lane));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (~b[q]) return (int)(w - g) + 64 * q + td_ctz64(~b[q]);
            w += 256;
            if (w - g >= lim) return lim;
        }
    }
    __device__ __forceinline__ int run_end(int k, int from) const { return first_clear([&](int64_t pos) { return pred(k, pos, true); }, from); }
    __device__ __forceinline__ int run_end2(int k1, int k2, int from) const {
        return first_clear([&](int64_t pos) { return pred(k1, pos, true) || pred(k2, pos, true); }, from);
    }
    template <class F>
    __device__ __forceinline__ int last_where(const F& f, int lo, int hi) const {  // highest i in [lo, hi) with f(g + i), -1 if none
        for (int64_t w = hi; w > lo; w -= 64) {
            const int64_t i = w - 64 + lane;  // window [w - 64, w)
            const uint64_t b = __ballot(i >= lo && f(g + i));
            if (b) return (int)(w - 64) + td_top64(b) - 1;
        }
        return -1;
    }
    __device__ __forceinline__ int last_and(int k1, int k2, int lo, int hi) const {
        return last_where([&](int64_t pos) { return pred(k1, pos, false) && pred(k2, pos, false); }, lo, hi);
    }
    __device__ __forceinline__ int last_set(int k, int lo, int hi) const { return last_where([&](int64_t pos) { return pred(k, pos, false); }, lo, hi); }
    __device__ __forceinline__ int last_clear(int k, int lo, int hi) const { return last_where([&](int64_t pos) { return !pred(k, pos, false); }, lo, hi); }
};

struct FarScan {  // one wavefront's view of the text for the far kernels
    const EncodeArgs& a;
    const Tables& T;
    int lane;
    __device__ __forceinline__ int64_t piece_end(int64_t g) const {  // end of the piece that starts at g
        const int64_t room = a.n - g + 4;
        WaveMaskP mp{T, a.text, a.docbits, a.n, g, 0, room > 0x7FFFFFF0ll ? 0x7FFFFFF0 : (int)room, lane};
        const uint8_t* text = a.text;
        const int64_t n = a.n;
        const int r = scan_piece_p(mp, [=](int q) { return g + q < n ? (uint32_t)text[g + q] : 0u; }, T.pat_flags);
        return r > 0 ? g + r : a.n;  // (r < 0 cannot happen: the provider reads to the end of the text)
    }
    __device__ __forceinline__ bool sync_at(int64_t p) const {
        WaveMaskP mp{T, a.text, a.docbits, a.n, p, 0, 8, lane};
        return is_sync(p > 0 ? mp.cf_at(p - 1) : 0u, mp.cf_at(p), T.pat_flags);
    }
    // does the fast kernel take over at p?  Every synchronisation point inside a tile is a head there.
    __device__ __forceinline__ bool fast_lane_starts_at(int64_t p) const { return sync_at(p); }
    // piece starts from p (a piece start) to the end of p's tile, marked as they are found; returns the first piece start at or
    // behind the tile end, or -1 when it stopped where a lane of the fast kernel took over
    __device__ __forceinline__ int64_t mark_to_tile_end(int64_t p, bool first_is_marked) const {
        const int64_t tile_end = (p - (p % KS_TILE) + KS_TILE < a.n) ? p - (p % KS_TILE) + KS_TILE : a.n;
        bool skip = first_is_marked;
        const int stride = digit_stride();
        while (p < tile_end) {
            if (!skip) {
                if (fast_lane_starts_at(p)) return -1;
                if (lane == 0) atomicOr(&a.startbits[p >> 5], 1u << (p & 31));
            }
            skip = false;
            if (stride) {
                const int64_t q = digit_run_skip(p, tile_end, stride);
                if (q > p) { p = q; continue; }
            }
            p = piece_end(p);
        }
        return p;
    }
    // Inside a run of digits nothing is a synchronisation point (is_sync): under \p{N}{1,3} a piece start is the run's start + 3 k, so
    // the tiles of a long run are a chain walked by ONE wavefront — piece by piece that was ~2 us per piece: 15 ms for 20 KB of
    // digits, 0.75 s for a megabyte.  Round 5: when the piece at p starts with an ASCII digit, the end of the run of ASCII digits
    // (no document start inside) is searched 256 bytes per step, as far as the tile end, and the piece starts p + stride k in
    // front of it are marked a word of START bits per lane (stride 3; 1 for the patterns that take digits one by one; GPT-2's
    // digit run is one piece: td_split_far_pieces).  A digit that is not ASCII ends the search: the walk goes on piece by piece
    // from the last start in front of it.  -> the last of those starts (unmarked: the walk checks and marks it), or p: no shortcut
    __device__ __forceinline__ int digit_stride() const {
        const uint32_t pf = T.pat_flags;
        return (pf & (PV_GPT2 | PV_GENERIC)) ? 0 : (pf & PV_SINGLE_DIGIT) ? 1 : 3;
    }
    __device__ __forceinline__ int64_t digit_run_skip(int64_t p, int64_t tile_end, int stride) const {
        if ((uint32_t)a.text[p] - (uint32_t)'0' > 9u) return p;
        const int64_t lim = tile_end + stride - 1 < a.n ? tile_end + stride - 1 : a.n;
        int64_t e = lim;
        for (int64_t w = p; w < lim; w += 256) {
            uint64_t b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t pos = w + 64 * q + lane;
                bool ok = pos < lim;
                if (ok) ok = (uint32_t)a.text[pos] - (uint32_t)'0' <= 9u && !(pos > p && ((a.docbits[pos >> 5] >> (pos & 31)) & 1u));
                b[q] = __ballot(ok);
            }
            int64_t stop = -1;
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (~b[q]) stop = w + 64 * q + td_ctz64(~b[q]);
            if (stop >= 0) { e = stop < lim ? stop : lim; break; }
        }
        const int64_t K = (e - p) / stride;
        if (K < 2) return p;
        const int64_t first = p + stride, last = p + stride * K;  // starts first, first + stride, ... in front of `last` (and of the tile end)
        const int64_t stop_at = last < tile_end ? last : tile_end;
        for (int64_t w = (first >> 5) + lane; w <= ((stop_at - 1) >> 5); w += 64) {
            uint32_t bits = 0;
            for (int bpos = 0; bpos < 32; ++bpos) {
                const int64_t pos = w * 32 + bpos;
                if (pos >= first && pos < stop_at && (pos - p) % stride == 0) bits |= 1u << bpos;
            }
            if (bits) atomicOr(&a.startbits[w], bits);
        }
        return last;
    }
    __device__ __forceinline__ void carry_to(int64_t from_tile, int64_t p) const {  // tiles (from_tile, tile of p]: first piece start = p
        const int64_t last = p >= a.n ? (int64_t)a.n_stiles - 1 : p / KS_TILE;
        // (agent scope: td_far_probe / td_tail read it behind a grid barrier that does no cache maintenance, ph_sync_light)
        for (int64_t t = from_tile + 1 + lane; t <= last; t += 64) __hip_atomic_store(&a.tile_carry[t], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// (bodies: bid / nb = this workgroup's number and the number of workgroups that share the work — blockIdx.x / gridDim.x in the kernels
// of their own, something else inside td_far_probe / td_tail, where one launch walks several of these phases)
__device__ __forceinline__ void far_pieces_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb) {
    const Tables T = uniform_tables(a.Tp);
    const int lane = threadIdx.x & 63;
    const uint32_t nitems = *a.slow_count < a.slow_cap ? *a.slow_count : a.slow_cap;
    const uint32_t nwaves = nb * (blockDim.x >> 6);
    const FarScan F{a, T, lane};
    for (uint32_t j = bid * (blockDim.x >> 6) + (threadIdx.x >> 6); j < nitems; j += nwaves) {
        const int64_t g = a.slow_list[j];  // a piece start the fast kernel marked; its end was beyond the window
#ifdef TD_FAR_DEBUG
        const unsigned long long t0 = __builtin_readcyclecounter();
        const int64_t pe = F.piece_end(g);
        const unsigned long long t1 = __builtin_readcyclecounter();
#endif
        const int64_t p = F.mark_to_tile_end(g, true);
#ifdef TD_FAR_DEBUG
        if (lane == 0) printf("far piece at %lld (tile %lld + %lld): ends %lld, walked to %lld; piece_end %llu ticks, all %llu ticks\n", (long long)g, (long long)(g / KS_TILE), (long long)(g % KS_TILE), (long long)pe, (long long)p, t1 - t0, (unsigned long long)__builtin_readcyclecounter() - t0);
#endif
        if (p >= 0) F.carry_to(g / KS_TILE, p);
    }
}

__global__ __launch_bounds__(256) void td_split_far_pieces(const EncodeArgs a) { far_pieces_body(a, blockIdx.x, gridDim.x); }

__device__ __forceinline__ void far_tiles_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb) {
    const Tables T = uniform_tables(a.Tp);
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)nb * (blockDim.x >> 6);
    const FarScan F{a, T, lane};
    // (64 tiles per step and wavefront, a lane each, looking for the heads of chains of flagged tiles: one tile per step was
    // 60 us per GiB of text for a kernel that normally finds nothing)
    for (int64_t t0 = ((int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; t0 < a.n_stiles; t0 += nwaves * 64) {
        const int64_t tl = t0 + lane;
        const bool head = tl < a.n_stiles && a.tile_flag[tl] && !(tl > 0 && a.tile_flag[tl - 1]);
        for (uint64_t hb = __ballot(head); hb; hb &= hb - 1ull) {
            const int64_t t = t0 + td_ctz64(hb);
            // walk the chain tile by tile.  The first piece start of a tile is where the walk crossed into it, or — when a lane of
            // the fast kernel took over in the tile before (it then scanned on to that tile's end), or a long piece was resolved by
            // td_split_far_pieces — what that scan recorded in tile_carry
            int64_t p = -1;
            for (int64_t tt = t; tt < a.n_stiles && a.tile_flag[tt]; ++tt) {
                const int64_t tg0 = tt * (int64_t)KS_TILE;
                if (p < tg0) p = __hip_atomic_load(&a.tile_carry[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written by td_split_tiles or by the far pieces' phase of this launch)
                if (p < tg0) { if (lane == 0) raise(a, TD_E_SCRATCH, tg0); break; }  // (cannot happen: nobody told this tile)
                if (p >= tg0 + KS_TILE || p >= a.n) continue;                        // a piece covers the whole tile
                p = F.mark_to_tile_end(p, false);
            }
        }
    }
}
__global__ __launch_bounds__(256) void td_split_far_tiles(const EncodeArgs a) { far_tiles_body(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------ td_probe_tiles ----------
// Token kernel, first half: pieces (from the START bitmap) -> one SLOT per piece, in piece order, in the tile's staging
// region: the id of the piece when it is a token (whole-piece lookup, CoreBPE::encode's fast path, tiktoken.cpp:209-215;
// single bytes through the 256-entry table), TOK_MISS | position | length when it is not (td_merge_tiles expands those),
// TOK_LONGREF | index for pieces above 64 bytes (td_long_pieces).  No per-byte token array and no compaction: slot k is
// piece k, lane k mod 256 stores it, the stores of a wavefront are consecutive.
__device__ __forceinline__ void probe_tiles_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb) {
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[K_BWIN + 16];
    __shared__ uint32_t s_start[K_BWIN / 32 + 3];  // bit i: a piece starts at tile byte i
    __shared__ __attribute__((aligned(16))) uint16_t s_plist[K_TILE + 8];  // tile positions of the piece starts (+ end delimiter)
    // after the probes the (then dead) piece list holds: pieces before each lane's 16-byte chunk / its START bits
    uint32_t* const s_off = reinterpret_cast<uint32_t*>(s_plist);
    uint16_t* const s_valid = s_plist + 2 * K_THREADS;
    __shared__ int32_t s_byteid[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_kmask[(P12_MAXLEN + 1) * 4];  // row len: byte masks of a len-byte key
    __shared__ uint32_t s_wave[8];
    __shared__ long long s_ext_end;                // global end of the tile's last piece when it leaves the window (else 0)
    __shared__ uint32_t s_flags;                   // TILE_HAS_LONG | TILE_HAS_MISS
    __shared__ uint32_t s_nrec;                    // missed pieces of the tile
    __shared__ uint32_t s_rec[K_MISS_LISTED_MAX];  // the first few: slot << 19 | tile position << 7 | length
    __shared__ unsigned long long s_pend[64];      // miss-list entries of the last tiles, not appended yet
    __shared__ uint32_t s_npend;
    __shared__ uint32_t s_flagl[32];               // flagged tiles of this workgroup, not appended yet
    __shared__ uint32_t s_nflagl;
    __shared__ uint32_t s_ncold;                   // pieces put aside for the long route
    __shared__ uint16_t s_coldk[K_THREADS];        // their indices in the piece list

    const int tid = threadIdx.x;
    const Tables T = uniform_tables(a.Tp);
    for (int q = tid; q < 256; q += K_THREADS) s_byteid[q] = T.byte_id[q];
    if (tid == 0) { s_npend = 0; s_nflagl = 0; }
    if (tid < (int)(P12_MAXLEN + 1) * 4) {
        const int len = tid >> 2, w = tid & 3, nb = len - 4 * w;  // bytes of dword w that belong to a len-byte key
        s_kmask[tid] = w == 3 || nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
    }

    uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0;
    uint32_t pfs = 0;  // START bits of window word `tid`
    const uint32_t list_max = TD_STOP(70) ? 0u : (uint32_t)K_MISS_LISTED_MAX;  // (70: tuning aid, no miss lists)
    // first wavefront: the np entries waiting in s_pend go to their class's global list; ONE atomic per class (same-address
    // atomics are served one after the other, tens of nanoseconds each: one per entry was +50 % on the whole kernel)
    auto append_pending = [&](uint32_t np) {
        const int ln = tid & 63;
        const bool have = (uint32_t)ln < np;
        const unsigned long long rec = have ? s_pend[ln] : 0ull;
        const uint32_t c = mq_class((uint32_t)rec & 127u);
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)K_MISS_CLASSES; ++q) {
            const uint64_t b = __ballot(have && c == q);
            if (b) {
                const int leader = (int)td_ctz64(b);
                uint32_t at = 0;
                if (ln == leader) at = atomicAdd(&a.miss_count[q], (uint32_t)__popcll((unsigned long long)b));
                at = (uint32_t)__shfl((int)at, leader);
                if (have && c == q) a.miss_list[(size_t)q * a.miss_cap + at + (uint32_t)__popcll((unsigned long long)(b & ((1ull << ln) - 1ull)))] = rec;
            }
        }
    };
    static_assert(K_BWIN / 32 + 3 <= K_THREADS, "one prefetched START word per lane covers the window");
    const int64_t nwords = (a.n + 31) >> 5;
    auto load_startword = [&](int64_t wg0_) -> uint32_t {
        const int64_t gw = (wg0_ >> 5) + tid;
        if (!(tid < K_BWIN / 32 + 3 && gw < nwords)) return 0u;
        // (the deferred tiles' START bits were completed by the far phases of this very launch, with atomics at agent scope and no cache
        // maintenance at the barrier in between: read them past this XCD's caches)
        return a.probe_deferred ? __hip_atomic_load(&a.startbits[gw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.startbits[gw];
    };
    // the token tiles this launch looks up: all of them, or (behind the fused tile loop) the ones it deferred
    const uint32_t* const tlist = a.probe_deferred ? a.deferred_list : nullptr;
    const int n_items = tlist ? (int)*a.deferred_count : a.n_tiles;
    auto tile_of = [&](int it) { return tlist ? (int)tlist[it] : it; };
    if ((int)bid < n_items) {
        const int64_t w0 = (int64_t)tile_of((int)bid) * K_TILE;
        pf0 = load_text16(a, w0 + (int64_t)tid * 16);
        if (tid < (K_BWIN + 16) / 16 - K_THREADS) pf1 = load_text16(a, w0 + (int64_t)(K_THREADS + tid) * 16);
        pfs = load_startword(w0);
    }
    for (int item = (int)bid; item < n_items; item += (int)nb) {
        const int tile = tile_of(item);
        const int64_t tile_g0 = (int64_t)tile * K_TILE;
        const int64_t wg0 = tile_g0;                 // window index 0 == first byte of the tile
        const int tile_hi = (int)((a.n - tile_g0 < K_TILE) ? (a.n - tile_g0) : K_TILE);
        const int c0 = tid * K_CHUNK, c1 = c0 + K_CHUNK;

        // ---- stage text + START bits of the tile (+128 B look-ahead).  Both were requested one iteration ago ----
        reinterpret_cast<uint4*>(s_txt)[tid] = pf0;
        if (tid < (K_BWIN + 16) / 16 - K_THREADS) reinterpret_cast<uint4*>(s_txt)[K_THREADS + tid] = pf1;
        if (tid < K_BWIN / 32 + 3) {
            uint32_t sw = pfs;
            const int64_t g = wg0 + (int64_t)tid * 32;
            if (a.n >= g && a.n < g + 32) sw |= 1u << (int)(a.n - g);  // the end of the text delimits the last piece
            s_start[tid] = sw;
        }
        if (item + (int)nb < n_items) {
            const int64_t nwg0 = (int64_t)tile_of(item + (int)nb) * K_TILE;
            {
                pf0 = load_text16(a, nwg0 + (int64_t)tid * 16);
                if (tid < (K_BWIN + 16) / 16 - K_THREADS) pf1 = load_text16(a, nwg0 + (int64_t)(K_THREADS + tid) * 16);
                pfs = load_startword(nwg0);
            }
        }
        if (tid == 0) { s_ext_end = 0; s_flags = 0; s_ncold = 0; s_nrec = 0; }
        __syncthreads();
        if (TD_STOP(30)) continue;

        // ---- dense list of the tile's piece starts (so that every lane has a piece to look up) ----
        uint32_t np_total, pbase, smask0;
        {
            uint32_t smask = 0;  // START bits of my 16 bytes
            if (c0 < tile_hi) {
                smask = (s_start[c0 >> 5] >> (c0 & 31)) & 0xFFFFu;
                if (c1 > tile_hi) smask &= (1u << (tile_hi - c0)) - 1u;
            }
            smask0 = smask;
            const uint32_t cnt = __popc(smask);
            pbase = block_excl_scan(cnt, s_wave, np_total);
            uint32_t k = pbase;
            while (smask) {
                const int b = __ffs(smask) - 1;
                smask &= smask - 1;
                s_plist[k++] = (uint16_t)(c0 + b);
            }
            if (tid < 64) {  // (first wavefront)
                // end of the last owned piece: first START at/after the tile end; beyond the staged window (a piece longer
                // than 128 bytes) walk the bitmap in HBM, 64 words per step (a megabyte piece: 500 steps, not 31 000)
                int e = tile_hi;
                while (e < K_BWIN && !((s_start[e >> 5] >> (e & 31)) & 1u)) ++e;
                if (e >= K_BWIN && np_total > 0) {
                    const int64_t g = wg0 + K_BWIN;
                    int64_t found = a.n;
                    for (int64_t gw0 = g >> 5; gw0 < nwords; gw0 += 64) {
                        const int64_t gw = gw0 + tid;
                        uint32_t sw = gw < nwords ? (a.probe_deferred ? __hip_atomic_load(&a.startbits[gw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.startbits[gw]) : 0u;
                        if (gw == (g >> 5)) sw &= ~((1u << (g & 31)) - 1u);
                        const uint64_t b = __ballot(sw != 0);
                        if (b) {
                            const int l = td_ctz64(b);
                            const int64_t f = (gw0 + l) * 32 + (__ffs(__shfl(sw, l)) - 1);
                            if (f < a.n) found = f;
                            break;
                        }
                    }
                    if (tid == 0) s_ext_end = found;
                    e = K_BWIN;  // placeholder; the probe uses s_ext_end for the last piece
                }
                if (tid == 0) s_plist[np_total] = (uint16_t)e;
            }
        }
        __syncthreads();
        if (TD_STOP(31)) continue;
        const long long ext_end = s_ext_end;
        // ---- probe: piece k -> lane k mod 256, one piece per lane at a time (more in flight cost registers, and registers
        //      cost resident wavefronts: both round 1 and round 2 measured 2 and 4 in flight slower).  Hot path = pieces of
        //      1..12 bytes: key = the bytes (three dwords cut out of LDS with funnel shifts, masked by length through a
        //      13-row table), ONE 16-byte load of the first slot of the exact-key table, three compares.  An empty slot is
        //      a miss; a slot held by another key, longer pieces, the piece that leaves the window: probe_piece_cold. ----
        uint32_t* const dst = a.stage + (size_t)tile * K_STAGE;
        // a missed piece (2..64 bytes, not a token): counted; the first few of a tile are remembered for the global list
        auto note_miss = [&](uint32_t k, uint32_t res) {
            const uint32_t mi = atomicAdd(&s_nrec, 1u);
            if (mi < (uint32_t)K_MISS_LISTED_MAX) s_rec[mi] = (k << 19) | (res & 0x7FFFFu);
        };
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const u32x4 __attribute__((address_space(1)))* gslot_t;  // (global loads, not flat ones)
        gslot_t const p12 = (gslot_t)(uintptr_t)T.piece12_slots;
        for (uint32_t k = tid; k < np_total; k += K_THREADS) {
            const int i = s_plist[k];
            uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)i;
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(s_txt) + (i >> 2);
            const uint32_t sh = (i & 3) * 8;
            const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
            const uint4 km = reinterpret_cast<const uint4*>(s_kmask)[len < P12_MAXLEN ? len : P12_MAXLEN];
            const uint32_t k0 = __funnelshift_r(w0, w1, sh) & km.x, k1 = __funnelshift_r(w1, w2, sh) & km.y,
                           k2 = __funnelshift_r(w2, w3, sh) & km.z;
            const u32x4 sl = p12[hash_piece12(k0, k1, k2, len) & (TD_STOP(32) ? 0xFu : T.piece12_mask)];  // (32: tuning aid, 16 slots only)
            const bool last_ext = ext_end && k == np_total - 1;
            uint32_t res;
            if (len <= P12_MAXLEN && a.use_fastpath && !last_ext && (sl.w == 0u || (sl.x == k0 && sl.y == k1 && sl.z == k2 && (sl.w >> 24) == (0x80u | len)))) {
                const bool miss = sl.w == 0u;  // empty slot: not a token (a single byte that is no token is an error, not a merge)
                res = miss ? (TOK_MISS | ((uint32_t)i << 7) | len) : (sl.w & 0x1FFFFFu);
                if (miss) {
                    if (len == 1) raise(a, TD_E_UNKNOWN_BYTE, wg0 + i);
                    note_miss(k, res);
                }
            } else {
                // everything else is put aside and handled densely after the loop, one piece per lane: 7 % of the pieces of
                // mixed-script text are longer than 12 bytes, i.e. nearly every wavefront pass of this loop held one, and the
                // whole wavefront then sat through the long route (hash over the bytes, verify against the token store)
                const uint32_t j = atomicAdd(&s_ncold, 1u);
                if (j < (uint32_t)K_THREADS) { s_coldk[j] = (uint16_t)k; continue; }
                if (last_ext) {
                    const long long l = ext_end - (wg0 + i);
                    len = l > 0x7FFFFFFFll ? 0xFFFFFFFFu : (uint32_t)l;
                }
                res = probe_piece_cold<false>(a, T, s_txt, s_byteid, &s_flags, wg0, i, len);  // (more than 256 of them in one tile)
                if ((res & 0xC0000000u) == TOK_MISS) note_miss(k, res);
            }
            dst[k] = res;
        }
        __syncthreads();
        {
            const uint32_t nc = s_ncold < (uint32_t)K_THREADS ? s_ncold : (uint32_t)K_THREADS;
            if ((uint32_t)tid < nc) {
                const uint32_t k = s_coldk[tid];
                const int i = s_plist[k];
                uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)i;
                if (ext_end && k == np_total - 1) {
                    const long long l = ext_end - (wg0 + i);
                    len = l > 0x7FFFFFFFll ? 0xFFFFFFFFu : (uint32_t)l;
                }
                const uint32_t res = probe_piece_cold<false>(a, T, s_txt, s_byteid, &s_flags, wg0, i, len);
                if ((res & 0xC0000000u) == TOK_MISS) note_miss(k, res);
                dst[k] = res;
            }
        }
        __syncthreads();
        if (TD_STOP(3)) continue;
        // ---- per-tile results: slot count + flags; the slot of every document that starts in this tile (documents are
        //      consecutive from the tile's first one, recorded by td_mark_docs; empty documents share a position) ----
        s_off[tid] = pbase;
        s_valid[tid] = (uint16_t)smask0;
        {
            // tiles with only a few missed pieces (running text: one in three tiles has one or two) put them on global lists,
            // one per length class, so that td_merge_pieces does not read every slot of a third of the tiles to find them and
            // does not end with a handful of pieces of every class per wavefront (a row of a list is a full batch); tiles with
            // many are flagged and scanned there
            // (collected in LDS over several tiles and appended 64 - K_MISS_LISTED_MAX or more at a time: a list position
            // comes from an atomic, and waiting for one per tile cost 5 % of the kernel)
            const uint32_t nr = s_nrec, np0 = s_npend;
            if (tid < 64) {  // (first wavefront: program order between its lanes' LDS accesses)
                const bool listed = nr && nr <= list_max;
                if (listed && (uint32_t)tid < nr) s_pend[np0 + tid] = ((unsigned long long)(uint32_t)tile << 32) | s_rec[tid];
                const uint32_t np1 = listed ? np0 + nr : np0;
                wave_sync_lds();
                if (np1 > 64u - (uint32_t)K_MISS_LISTED_MAX) {
                    append_pending(np1);
                    wave_sync_lds();
                    if (tid == 0) s_npend = 0;
                } else if (tid == 0) {
                    s_npend = np1;
                }
            }
        }
        if (tid == 0) {
            uint32_t fl = s_flags;
            const uint32_t nr = s_nrec;
            if (nr > list_max) {
                // flagged: td_merge_pieces scans the tile.  It draws the flagged tiles from a list (two at a time: runs of
                // consecutive tiles, grown while they were light, sent whole heavy stretches to single wavefronts: 2.7 ms
                // instead of 1.2 on the reference's code file set), appended 32 at a time per workgroup
                fl |= TILE_HAS_MISS;
                uint32_t nf = s_nflagl;
                s_flagl[nf++] = (uint32_t)tile;
                if (nf == 32u) {
                    const uint32_t at = atomicAdd(a.flagged_count, 32u);
                    for (uint32_t q = 0; q < 32u; ++q) a.flagged_list[at + q] = s_flagl[q];
                    nf = 0;
                }
                s_nflagl = nf;
            }
            else if (nr) fl |= TILE_MISS_LISTED;
            a.tile_count[tile] = np_total | fl;
        }
        __syncthreads();
        {
            const int64_t tile_end_g = tile_g0 + tile_hi;
            const int64_t fd = (int64_t)a.tile_first_doc[tile];
            for (int64_t d = fd + tid; d < a.n_docs; d += K_THREADS) {
                const int64_t p = a.doc_offsets[d];
                if (p >= tile_end_g) break;
                const int lp = (int)(p - tile_g0);
                a.doc_slot[d] = s_off[lp >> 4] + __popc((uint32_t)s_valid[lp >> 4] & ((1u << (lp & 15)) - 1u));
            }
        }
        __syncthreads();
    }
    if (tid < 64) append_pending(s_npend);  // (what is still waiting in LDS)
    if (tid == 0 && s_nflagl) {
        const uint32_t nf = s_nflagl, at = atomicAdd(a.flagged_count, nf);
        for (uint32_t q = 0; q < nf; ++q) a.flagged_list[at + q] = s_flagl[q];
    }
}
__global__ __launch_bounds__(K_THREADS, TD_PROBE_MIN_WAVES) void td_probe_tiles(const EncodeArgs a) { probe_tiles_body(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------ byte-pair merge, one lane per piece ----
// Token kernel, second half: the pieces td_probe_tiles marked TOK_MISS (2..64 bytes, not a token).  A piece is merged by ONE
// LANE (mg_round, td_common.h: every lane advances its own merge chain, a merge per round; a round is a minimum over the
// lane's keys in LDS, a handful of bit operations on its part mask and four pair-table probes in flight).
// (Round 1 merged with one lane per BYTE and a segmented min-scan per 64-byte window: ~120 instructions per round for the
// 5-10 pieces of a window, and only as many probes in flight as there were pieces in the window.  A first lane-per-piece
// version that worked tile by tile — a workgroup per tile, the tile's ~80 missed pieces in five class batches one after
// the other — ran 163 rounds per tile at 2 % lane utilisation and was slower than round 1; batches have to be FULL, so they
// are filled across tiles.)
// ------------------------------------------------------------------ td_collect_misses ---------
// The missed pieces of the FLAGGED tiles (more than K_MISS_LISTED_MAX of them in the tile: mixed-script text and code have ~80 per
// tile) go onto lists by length class too, so that td_merge_pieces takes nothing but full rows.  (Rounds 2-4: td_merge_pieces scanned
// the flagged tiles itself and queued their pieces in LDS until a class had a full batch — 28 % of its time on mixed-script text,
// 36 % on the code file set, at the three wavefronts per SIMD its key arrays + queues allowed.)  A WAVEFRONT per tile: pass 1
// counts the tile's misses per class, ONE atomic per class reserves their places (on list gw % COLL_SUBS of the class: the
// counters of one class are COLL_SUBS different cache lines), pass 2 writes the records — no LDS, full occupancy.  A class that
// finds no room (lists are sized for a few times the density of real text, not for the worst case of a miss every two bytes)
// is noted in the tile's count word; td_merge_pieces scans those tiles for those classes after the rows.
// Round 5: ... and a piece is merged ONCE per call however often its bytes occur.  The tile's missed slots are compacted into LDS (a
// group of eight rows at a time; what is left of a group waits for the next one, so the dense part always runs with 64 lanes), then a
// lane per piece: hash of its bytes, a look at the table of the pieces seen so far in this call (EncodeArgs::dd_table), and either the
// piece is the first one with these bytes (its record goes into the table with one compare-and-swap, and onto a list), or the table
// names a piece with the same hash tag and length whose bytes ARE the same (compared, 16 bytes a step): then the pair (this record,
// that record) goes on the list of repeats and td_copy_dups copies the ids when td_merge_pieces is done.  Equal bytes have equal ids
// (bpe_merge reads nothing but the piece, tiktoken.cpp:298-368), so this changes no result; missed words repeat (8 MiB of mixed-script
// text: 168 000 missed pieces, 14 000 distinct; the reference's code file set: 290 000 / 4 500).  A full table (or two occupied seats)
// only means the piece is merged itself.  Records and repeats are held back in LDS until a wavefront has 64 of a kind: one atomic
// per 64 whatever the tiles were.
constexpr int DD_BUF = 512 + 64;   // compacted slots of eight rows + what the group before left over
__device__ __forceinline__ void dd_chunk(const uint8_t* text, int64_t n_text, int64_t g, uint32_t left, uint32_t (&w)[4]) {  // 16 bytes at g, zero from byte `left` on
    uint32_t t[5];
    load_piece_window(text, n_text, g, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int32_t l = (int32_t)left - 4 * i;
        w[i] = l >= 4 ? t[i] : l <= 0 ? 0u : (t[i] & ((1u << (8 * l)) - 1u));
    }
}
__device__ __forceinline__ uint32_t dd_mix(uint32_t h, uint32_t w) {  // (murmur3's mixing)
    uint32_t k = w * 0xCC9E2D51u;
    k = (k << 15) | (k >> 17);
    h ^= k * 0x1B873593u;
    return ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
}
__device__ __forceinline__ uint32_t dd_fmix(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ bool dd_same_from(const uint8_t* text, int64_t n_text, int64_t g, int64_t g2, uint32_t len, uint32_t from) {
    uint32_t diff = 0;
    for (uint32_t c = from; c < len; c += 16u) {
        uint32_t w[4], v[4];
        dd_chunk(text, n_text, g + c, len - c, w);
        dd_chunk(text, n_text, g2 + c, len - c, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) diff |= w[i] ^ v[i];
    }
    return diff == 0u;
}
constexpr unsigned long long DD_REC_MASK = (1ull << 56) - 1ull;  // (a record's tile stays below 2^24: dedupe is off above 64 GiB of text)
constexpr int DD_CTR = K_MISS_CLASSES * COLL_SUBS;  // coll_count[(DD_CTR + s) * COLL_STRIDE]: entries on list s of the repeats

#ifndef TD_COLLECT_MIN_WAVES
#define TD_COLLECT_MIN_WAVES 6
#endif
constexpr int COLLECT_LDS_WORDS = (K_THREADS / 64) * DD_BUF;
__device__ __forceinline__ void collect_misses_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb, uint32_t* const lds) {
    uint32_t (*const s_m)[DD_BUF] = reinterpret_cast<uint32_t (*)[DD_BUF]>(lds);  // [K_THREADS / 64][DD_BUF]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = (int)bid * (K_THREADS / 64) + wv, nw = (int)nb * (K_THREADS / 64);
    const int n_flagged = (int)*a.flagged_count;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t sub = (uint32_t)gw % (uint32_t)COLL_SUBS;
    uint32_t* const mb = s_m[wv];
    // (a repeat = tile << (25 + seat bits) | slot << (12 + seat bits) | tile position << seat bits | the table seat that names the piece whose
    // ids it gets; seat bits = a.dd_seat_bits: what the tile number leaves of 39 bits)
    const uint32_t sb = a.dd_seat_bits;

    // no room on a list: the pieces are marked for the scan behind td_merge_pieces' rows (lists are sized for a few times the density of
    // real text, not for a miss every two bytes)
    auto to_scan = [&](const unsigned long long rec) {
        const uint32_t tile = (uint32_t)(rec >> 32), k = ((uint32_t)rec >> 19) & 0x1FFFu;
        a.stage[(size_t)tile * K_STAGE + k] = TOK_MISS | TOK_OVF | ((uint32_t)rec & 0x7FFFFu);
        atomicOr(&a.tile_count[tile], 1u << (TILE_OVF_SHIFT + mq_class((uint32_t)rec & 127u)));
    };
    // the dense part: entries mb[head .. head + cnt) of `tile`, a lane each
    auto process = [&](const uint32_t tile, const uint32_t head, const uint32_t cnt) {
        const bool act = (uint32_t)lane < cnt;
        const uint32_t e = act ? mb[head + (uint32_t)lane] : 0u;  // slot << 19 | tile position << 7 | length
        const uint32_t len = e & 127u, pos = (e >> 7) & 0xFFFu;
        const unsigned long long rec = ((unsigned long long)tile << 32) | e;
        unsigned long long other = 0ull;
        uint32_t seat = 0;
        if (act && a.dedupe && len >= a.dd_minlen) {  // (shorter pieces are merged in fewer rounds than the detour through the table costs)
            const int64_t g = (int64_t)tile * K_TILE + pos;
            // hash over the piece's dwords; its first 16 bytes stay in registers for the comparison
            uint32_t w0[4];
            dd_chunk(a.text, a.n, g, len, w0);
            uint32_t h = 0x9747B28Cu ^ len;
#pragma unroll
            for (int i = 0; i < 4; ++i) h = dd_mix(h, w0[i]);
            for (uint32_t c = 16u; c < len; c += 16u) {
                uint32_t w[4];
                dd_chunk(a.text, a.n, g + c, len - c, w);
#pragma unroll
                for (int i = 0; i < 4; ++i) h = dd_mix(h, w[i]);
            }
            h = dd_fmix(h);
            const unsigned long long mine = rec | ((unsigned long long)(h >> 24) << 56);
            // (a.dd_replicas seats per piece, one per group of workgroups: the seats — and the text — of a handful of very frequent
            // pieces are otherwise a few cache lines that every CU of the chip asks one L2 channel for)
            // Measured on 256 MiB (collect + merge + copy): chat markup with all specials 0.89 ms with one seat per piece, 0.40 with four, 0.29
            // with sixteen; mixed-script text 0.64 / 0.72 / 0.78 (more pieces merged); the code file set 0.98 / 0.95 / 1.04.  Four.
            uint32_t i = (h + (bid & (a.dd_replicas - 1u)) * 0x61C88647u) & a.dd_mask;
            for (int probe = 0; probe < 2; ++probe, i ^= 1u) {
                // (a plain load first: the seats of frequent pieces are taken early and then only ever read, out of the CU's own cache;
                // atomics on one address are served one after the other.  A stale zero only costs the compare-and-swap it leads to.)
                unsigned long long cur = a.dd_table[i];
                if (cur == 0ull) {
                    // (the zero may be this CU's stale cache line, or the kernel has only just begun and every wavefront finds the seats
                    // of the few most frequent pieces empty at once.  A load that bypasses the vector cache first: it is not serialised.)
                    cur = __hip_atomic_load(&a.dd_table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == 0ull) cur = atomicCAS(&a.dd_table[i], 0ull, mine);
                    if (cur == 0ull) break;  // the first piece with these bytes (as far as the table knows): merged, and named to the others
                }
                if ((cur >> 56) == (mine >> 56) && ((uint32_t)cur & 127u) == len) {
                    const int64_t g2 = (int64_t)(uint32_t)((cur & DD_REC_MASK) >> 32) * K_TILE + (((uint32_t)cur >> 7) & 0xFFFu);
                    uint32_t v0[4];
                    dd_chunk(a.text, a.n, g2, len, v0);
                    if (((w0[0] ^ v0[0]) | (w0[1] ^ v0[1]) | (w0[2] ^ v0[2]) | (w0[3] ^ v0[3])) == 0u && dd_same_from(a.text, a.n, g, g2, len, 16u)) {
                        other = cur & DD_REC_MASK;
                        seat = i;
                        break;
                    }
                }
            }
        }
        // Onto the lists: ONE atomic instruction per 64 pieces — lane c reserves the places of class c on this wavefront's list of that class,
        // lane K_MISS_CLASSES those of the repeats (six different addresses) — then every lane writes its own record.  (Round 5's first form
        // held records back in LDS until 64 of a kind were there: 26 KB of LDS per workgroup kept the kernel at 4 wavefronts per SIMD, and
        // it waits for memory 72 % of the time.)
        const bool dup = other != 0ull, uq = act && !dup;
        const uint32_t cls = mq_class(len);
        uint64_t bc[K_MISS_CLASSES + 1];
        uint32_t myn = 0;
#pragma unroll
        for (int c = 0; c < K_MISS_CLASSES; ++c) {
            bc[c] = __ballot(uq && cls == (uint32_t)c);
            if (lane == c) myn = (uint32_t)__popcll((unsigned long long)bc[c]);
        }
        bc[K_MISS_CLASSES] = __ballot(dup);
        if (lane == K_MISS_CLASSES) myn = (uint32_t)__popcll((unsigned long long)bc[K_MISS_CLASSES]);
        uint32_t at = 0;
        if (myn) at = atomicAdd(&a.coll_count[((uint32_t)lane < (uint32_t)K_MISS_CLASSES ? (uint32_t)lane * COLL_SUBS + sub : (uint32_t)DD_CTR + sub) * COLL_STRIDE], myn);
        uint32_t ats[K_MISS_CLASSES + 1];  // (every lane learns all six answers: uniform)
#pragma unroll
        for (int c = 0; c <= K_MISS_CLASSES; ++c) ats[c] = (uint32_t)__builtin_amdgcn_readlane((int)at, c);
        bool lost = false;  // my piece found no room
        if (uq) {
            uint32_t at_c = 0, cap = 0, rank = 0;
            unsigned long long base = 0;
#pragma unroll
            for (int c = 0; c < K_MISS_CLASSES; ++c)
                if (cls == (uint32_t)c) { at_c = ats[c]; cap = a.coll_cap[c]; base = a.coll_base[c]; rank = (uint32_t)__popcll((unsigned long long)(bc[c] & lt)); }
            const uint32_t idx = at_c + rank;
            if (idx < cap && idx >= at_c) a.miss_list[base + (size_t)sub * cap + idx] = rec;
            else { to_scan(rec); lost = true; }
        }
        if (dup) {
            const uint32_t at_d = ats[K_MISS_CLASSES];
            const uint32_t idx = at_d + (uint32_t)__popcll((unsigned long long)(bc[K_MISS_CLASSES] & lt));
            if (idx < a.dup_cap && idx >= at_d) a.dup_list[(size_t)sub * a.dup_cap + idx] = ((((unsigned long long)tile << 13 | (e >> 19)) << 12 | pos) << sb) | seat;
            else { to_scan(rec); lost = true; }  // (merged by the scan after all)
        }
        if (__ballot(lost) && lane == 0) atomicAdd(a.ovf_count, 1u);
    };

    for (int f = gw; f < n_flagged; f += nw) {
        const uint32_t tile = a.flagged_list[f];
        const uint32_t cnt = a.tile_count[tile] & TILE_COUNT_MASK;
        const uint32_t* slots = a.stage + (size_t)tile * K_STAGE;
        const uint32_t rows = (cnt + 63u) >> 6;
        uint32_t tail = 0;
        for (uint32_t row = 0; row < rows; row += 8) {
            uint32_t v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t k = (row + r) * 64u + lane;
                v[r] = k < cnt ? slots[k] : 0u;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool miss = (v[r] & 0xC0000000u) == TOK_MISS;
                const uint64_t b = __ballot(miss);
                if (miss) mb[tail + (uint32_t)__popcll((unsigned long long)(b & lt))] = ((((row + r) * 64u + lane) & 0x1FFFu) << 19) | (v[r] & 0x7FFFFu);
                tail += (uint32_t)__popcll((unsigned long long)b);
            }
            wave_sync_lds();
            uint32_t head = 0;
            for (; tail - head >= 64u; head += 64u) process(tile, head, 64u);
            if (head) {
                const uint32_t rem = tail - head;
                const uint32_t x = (uint32_t)lane < rem ? mb[head + (uint32_t)lane] : 0u;
                wave_sync_lds();
                if ((uint32_t)lane < rem) mb[lane] = x;
                tail = rem;
                wave_sync_lds();
            }
        }
        if (tail) process(tile, 0u, tail);
        wave_sync_lds();
    }
}
__global__ __launch_bounds__(K_THREADS, TD_COLLECT_MIN_WAVES) void td_collect_misses(const EncodeArgs a) {
    __shared__ uint32_t s_lds[COLLECT_LDS_WORDS];
    collect_misses_body(a, blockIdx.x, gridDim.x, s_lds);
}

// td_copy_dups (behind td_merge_pieces): a lane per repeat — the other piece's finished slot (its id count), this piece's slot (TOK_DUPREF:
// the seat that names the other piece + the count; the ids themselves are NOT copied, the pack kernels read them where they are), and the tile's
// extra ids with one atomic per tile of the wavefront's 64 repeats (they come from two or three tiles).
__device__ __forceinline__ void copy_dups_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb) {
    const int lane = threadIdx.x & 63;
    const uint32_t gw = (bid * (uint32_t)K_THREADS + threadIdx.x) >> 6, nw = nb * (uint32_t)(K_THREADS / 64);
    // With allowed special tokens (td_special_ids puts a literal's id where its first piece's ids were) or a generic pattern (markers of its
    // own behind this kernel) a piece's place in merge_out does not keep its ids until the pack kernels run: the ids are copied here then.
    const bool by_ref = a.sp.n == 0u && !(a.pat_flags & PV_GENERIC);
    if (bid == 0 && threadIdx.x < (unsigned)(K_MISS_CLASSES + 1) * COLL_SUBS) {  // (statistics: TD_INFO_REPEATS, TD_INFO_LISTED_PIECES; a thread per list)
        const uint32_t q = threadIdx.x, c = a.coll_count[q * COLL_STRIDE];
        uint32_t cap = a.dup_cap;
#pragma unroll
        for (int k = 0; k < K_MISS_CLASSES; ++k)
            if (q / COLL_SUBS == (uint32_t)k) cap = a.coll_cap[k];
        if (c) atomicAdd(&a.dd_stats[q >= (uint32_t)DD_CTR ? 0 : 1], c < cap ? c : cap);
    }
    for (uint32_t sub = 0; sub < (uint32_t)COLL_SUBS; ++sub) {
        const uint32_t c0 = a.coll_count[((uint32_t)DD_CTR + sub) * COLL_STRIDE];
        const uint32_t cnt = c0 < a.dup_cap ? c0 : a.dup_cap;
        for (uint32_t base = gw * 64u; base < cnt; base += nw * 64u) {
            const uint32_t idx = base + (uint32_t)lane;
            uint32_t tile = 0xFFFFFFFFu, extra = 0;
            if (idx < cnt) {
                const unsigned long long ent = a.dup_list[(size_t)sub * a.dup_cap + idx];
                const uint32_t sb = a.dd_seat_bits;
                tile = (uint32_t)(ent >> (25u + sb));
                const uint32_t k = (uint32_t)(ent >> (12u + sb)) & 0x1FFFu, pos = (uint32_t)(ent >> sb) & 0xFFFu;
                const unsigned long long other = a.dd_table[(uint32_t)ent & ((1u << sb) - 1u)] & DD_REC_MASK;
                const uint32_t r_tile = (uint32_t)(other >> 32), r_k = ((uint32_t)other >> 19) & 0x1FFFu;
                const uint32_t rs = a.stage[(size_t)r_tile * K_STAGE + r_k];
                uint32_t nt = rs & 127u;
                if ((rs & (0xC0000000u | TOK_MERGED)) != (TOK_MISS | TOK_MERGED) || nt == 0u) {
                    raise(a, TD_E_HIP, (int64_t)tile * K_TILE + pos);  // (cannot happen: the other piece was on a list or marked for the scan)
                    nt = 1u;
                }
                if (by_ref) {
                    // (the ids stay where they are: the slot names the seat, the pack kernels read them there — marker_ids)
                    a.stage[(size_t)tile * K_STAGE + k] = TOK_MISS | TOK_MERGED | TOK_DUPREF | (((uint32_t)ent & ((1u << sb) - 1u)) << 7) | nt;
                } else {
                    const uint32_t* const src = a.merge_out + (size_t)r_tile * K_STAGE + (((uint32_t)other >> 7) & 0xFFFu);
                    uint32_t* const mo = a.merge_out + (size_t)tile * K_STAGE + pos;
                    for (uint32_t i = 0; i < nt; i += 4u) {
                        const uint32_t x0 = src[i], x1 = i + 1u < nt ? src[i + 1u] : 0u, x2 = i + 2u < nt ? src[i + 2u] : 0u, x3 = i + 3u < nt ? src[i + 3u] : 0u;
                        mo[i] = x0;
                        if (i + 1u < nt) mo[i + 1u] = x1;
                        if (i + 2u < nt) mo[i + 2u] = x2;
                        if (i + 3u < nt) mo[i + 3u] = x3;
                    }
                    a.stage[(size_t)tile * K_STAGE + k] = TOK_MISS | TOK_MERGED | (pos << 7) | nt;
                }
                extra = nt - 1u;
            }
            for (uint64_t pend = __ballot(extra != 0); pend;) {
                const int l = td_ctz64(pend);
                const uint32_t tl = (uint32_t)__shfl((int)tile, l);
                const bool same = extra != 0 && tile == tl;
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(same ? extra : 0u, lane), 63);
                if (lane == l) atomicAdd(&a.tile_extra[tl], tot);
                pend &= ~__ballot(same);
            }
        }
    }
}
__global__ __launch_bounds__(K_THREADS) void td_copy_dups(const EncodeArgs a) { copy_dups_body(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------ td_merge_pieces ---------
// Every wavefront works alone (no workgroup barrier).  Its work comes as ROWS of the miss lists — K_MISS_CLASSES lists the tile
// loops fill (tiles with a few missed pieces) and K_MISS_CLASSES x COLL_SUBS lists td_collect_misses fills (the flagged tiles) —
// a row = one full batch of one length class (64 / units pieces; <= 8, 16, 32, 48, 64 bytes = 1, 1, 2, 3, 4 units), dealt
// round-robin over all wavefronts of the grid.  One piece per owner lane, the rounds of a batch run until its longest chain is
// done — pieces of one class need about the same number of rounds.
// A merged piece's ids go to its own bytes' slots of the result buffer (a.merge_out[tile * K_STAGE + tile position + i]:
// pieces do not overlap and a piece has at most as many ids as bytes), its slot becomes TOK_MISS | position << 7 | ids and
// the tile's extra ids are added to tile_extra; td_pack_tokens expands the markers.
// Behind the rows, only when td_collect_misses found a list full (*ovf_count): the tiles it marked are scanned for the classes it
// marked, 16 pieces of 4 units at a time — slow, and only there so that no input is refused.
constexpr int MG_LISTS = K_MISS_CLASSES * (1 + COLL_SUBS);
static_assert(MG_LISTS <= 64, "a lane per list");

#ifndef TD_MERGE_MIN_WAVES
#define TD_MERGE_MIN_WAVES 4
#endif
#ifndef TD_MERGE_THREADS
#define TD_MERGE_THREADS 256
#endif
constexpr int MG_THREADS = TD_MERGE_THREADS;  // (a wavefront's key + id arrays are 8 KB)
constexpr int MERGE_LDS_WORDS = 2 * (MG_THREADS / 64) * 64 * MG_UNIT;
__device__ __forceinline__ void merge_pieces_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb, uint32_t* const lds) {  // lds: 16-byte aligned
    constexpr int NW = MG_THREADS / 64;
    uint32_t (*const s_keys)[64 * MG_UNIT] = reinterpret_cast<uint32_t (*)[64 * MG_UNIT]>(lds);
    uint32_t (*const s_ids)[64 * MG_UNIT] = reinterpret_cast<uint32_t (*)[64 * MG_UNIT]>(lds + NW * 64 * MG_UNIT);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const Tables T = uniform_tables(a.Tp);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t* const keys = s_keys[wv];
    uint32_t* const ids = s_ids[wv];

#ifdef TD_MERGE_TIMING
    unsigned long long t_init = 0, t_rounds = 0, t_out = 0, t_total0 = __builtin_readcyclecounter(), n_batches = 0, n_rounds = 0;
#define TD_TICK(var) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_now = __builtin_readcyclecounter(); var += t_now - t_last; t_last = t_now; }
#else
#define TD_TICK(var)
#endif
    // one batch: lane's piece = rec (0: none), its units start at unit `t0`; wide: part masks of 64 bits (pieces above 32 bytes)
    auto run_batch = [&](const unsigned long long rec, const uint32_t t0, const bool wide) {
#ifdef TD_MERGE_TIMING
        unsigned long long t_last = __builtin_readcyclecounter();
        ++n_batches;
#endif
        MergeState st;
        st.t = t0;
        st.len = (uint32_t)rec & 127u;
        st.alive = st.len >= 64u ? ~0ull : ((1ull << st.len) - 1ull);
        const uint32_t tile = (uint32_t)(rec >> 32), pos = ((uint32_t)rec >> 7) & 0xFFFu;
        const int64_t gpos = (int64_t)tile * K_TILE + pos;
        if (st.len) mg_init_piece(a.text, a.n, T, keys, ids, st, gpos);
        TD_TICK(t_init)
        for (;;) {  // (pieces of at most 32 bytes: the part mask is one register)
            if (TD_STOP(41)) break;
#ifdef TD_MERGE_TIMING
            ++n_rounds;
#endif
            const bool more = wide ? mg_round_t<uint64_t>(T, keys, ids, st) : mg_round_t<uint32_t>(T, keys, ids, st);  // (uniform)
            if (!__any(more)) break;
        }
        TD_TICK(t_rounds)
        // A merged piece's ids go to its own bytes' slots of the tile's region of merge_out; the tile's extra ids are added with
        // ONE atomic per tile of the batch (a batch's pieces come from a few tiles, and 64 atomics on one address are served
        // one after the other).  (Filling the region densely instead needs the atomic's answer before the ids can be
        // written: +0.2 ms on mixed-script text, and td_pack_tokens was no faster for it.)
        uint32_t extra = 0;
        if (st.len) {
            uint32_t* out = a.merge_out + (size_t)tile * K_STAGE + pos;
            uint32_t nt = 0;
            for (uint64_t al = st.alive; al; al &= al - 1ull) {
                const uint32_t j = (uint32_t)td_ctz64(al);
                const uint32_t id = ids[mg_slot(st.t, j)];
                if ((int32_t)id >= T.pseudo_base) raise(a, TD_E_UNKNOWN_BYTE, gpos + j);
                out[nt++] = id;
            }
            // (a piece of >= 2 bytes always keeps a part: nt >= 1.  td_pack_plain shifts by ids - 1 per marker and the count here adds
            // nt - 1: the two sides agree only for nt >= 1, so a marker of td_merge_pieces with no id is an error, not a shift — ADVICE r4)
            if (nt == 0u) raise(a, TD_E_HIP, gpos);
            a.stage[(size_t)tile * K_STAGE + (((uint32_t)rec >> 19) & 0x1FFFu)] = TOK_MISS | TOK_MERGED | (pos << 7) | nt;
            extra = nt > 1 ? nt - 1 : 0;
        }
        for (uint64_t pend = __ballot(extra != 0); pend;) {
            const int l = td_ctz64(pend);
            const uint32_t tl = (uint32_t)__shfl((int)tile, l);
            const bool same = extra != 0 && tile == tl;
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(same ? extra : 0u, lane), 63);
            if (lane == l) atomicAdd(&a.tile_extra[tl], tot);
            pend &= ~__ballot(same);
        }
        wave_sync();  // (the batch's LDS reads are done before the next batch's writes)
        TD_TICK(t_out)
    };

    // the lists, one per lane: 0..4 the tile loops' (class = lane), then td_collect_misses' (class-major)
    uint32_t l_cls = 0, l_cnt = 0;
    unsigned long long l_base = 0;
    if (lane < K_MISS_CLASSES) {
        l_cls = (uint32_t)lane;
        const uint32_t c0 = a.miss_count[lane];
        l_cnt = c0 < a.miss_cap ? c0 : a.miss_cap;
        l_base = (unsigned long long)lane * a.miss_cap;
    } else if (lane < MG_LISTS) {
        const uint32_t q = (uint32_t)lane - K_MISS_CLASSES;
        l_cls = q / COLL_SUBS;
        uint32_t cap = 0;
        unsigned long long base = 0;
#pragma unroll
        for (int c = 0; c < K_MISS_CLASSES; ++c)
            if (l_cls == (uint32_t)c) { cap = a.coll_cap[c]; base = a.coll_base[c]; }
        const uint32_t c0 = a.coll_count[q * COLL_STRIDE];
        l_cnt = c0 < cap ? c0 : cap;
        l_base = base + (unsigned long long)(q % COLL_SUBS) * cap;
    }
    const uint32_t l_per = 64u / mq_units(l_cls);
    const uint32_t l_rows = (l_cnt + l_per - 1u) / l_per;
    const uint32_t l_incl = wave_incl_scan(l_rows, lane), l_excl = l_incl - l_rows;
    const uint32_t total_rows = (uint32_t)__builtin_amdgcn_readlane((int)l_incl, 63);
    const uint32_t gw = bid * NW + (uint32_t)wv, nwaves_all = nb * NW;

    // One call site of run_batch (the batch code is large: the kernel must stay inside the instruction cache): first the rows, then —
    // normally never — the overflow scan.
    const uint32_t n_ovf = *a.ovf_count;
    const int n_flagged = n_ovf ? (int)*a.flagged_count : 0;
    uint32_t row = gw;
    int of = (int)gw - (int)nwaves_all;  // overflow scan: position on the flagged list, the tile's state
    uint32_t o_tile = 0, o_mask = 0, o_cnt = 0, o_row = 0, o_v = 0;
    uint64_t o_pend = 0;
    for (;;) {
        unsigned long long rec = 0;
        uint32_t t0 = (uint32_t)lane;
        bool wide = false;
        if (row < total_rows) {
            const int L = (int)td_ctz64(__ballot(l_rows != 0u && row >= l_excl && row < l_incl));
            const uint32_t c = (uint32_t)__shfl((int)l_cls, L), r_in = row - (uint32_t)__shfl((int)l_excl, L), cntL = (uint32_t)__shfl((int)l_cnt, L);
            const unsigned long long base = ((unsigned long long)(uint32_t)__shfl((int)(l_base >> 32), L) << 32) | (uint32_t)__shfl((int)(uint32_t)l_base, L);
            const uint32_t u = mq_units(c), per = 64u / u;
            const uint32_t i = (uint32_t)lane / u, idx = r_in * per + i;
            if ((uint32_t)lane == i * u && i < per && idx < cntL) rec = a.miss_list[base + idx];
            wide = c > 2u;
            row += nwaves_all;
        } else if (n_ovf) {
            if (!o_pend) {  // the next row of slots with a piece of a marked class
                ++o_row;
                if (o_row * 64u >= o_cnt) {
                    of += (int)nwaves_all;
                    if (of >= n_flagged) break;
                    o_tile = a.flagged_list[of];
                    const uint32_t tc = a.tile_count[o_tile];
                    o_mask = (tc >> TILE_OVF_SHIFT) & ((1u << K_MISS_CLASSES) - 1u);
                    o_cnt = o_mask ? (tc & TILE_COUNT_MASK) : 0u;
                    o_row = 0xFFFFFFFFu;
                    continue;
                }
                const uint32_t k = o_row * 64u + (uint32_t)lane;
                o_v = k < o_cnt ? a.stage[(size_t)o_tile * K_STAGE + k] : 0u;
                o_pend = __ballot((o_v & (0xC0000000u | TOK_MERGED | TOK_OVF)) == (TOK_MISS | TOK_OVF));  // (the slots whose records found no room)
                continue;
            }
            const uint32_t rank = (uint32_t)__popcll((unsigned long long)(o_pend & lt));
            const bool sel = ((o_pend >> lane) & 1ull) && rank < 16u;
            if (sel) rec = ((unsigned long long)o_tile << 32) | (((o_row * 64u + (uint32_t)lane) & 0x1FFFu) << 19) | (o_v & 0x7FFFFu);
            t0 = rank * 4u;
            wide = true;
            o_pend &= ~__ballot(sel);
        } else {
            break;
        }
        run_batch(rec, t0, wide);
    }
#ifdef TD_MERGE_TIMING
    if (lane == 0 && (bid % 97) == 0 && wv == 0)
        printf("merge wave b%d: total %llu init %llu rounds %llu out %llu cycles, %llu batches %llu rounds\n", (int)bid,
               (unsigned long long)(__builtin_readcyclecounter() - t_total0), t_init, t_rounds, t_out, n_batches, n_rounds);
#endif
}
__global__ __launch_bounds__(MG_THREADS, TD_MERGE_MIN_WAVES) void td_merge_pieces(const EncodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[MERGE_LDS_WORDS];
    merge_pieces_body(a, blockIdx.x, gridDim.x, s_lds);
}

// ------------------------------------------------------------------ td_long_pieces ----------
// Pieces longer than K_MAXSHORT bytes (runs of one character, long identifiers, CJK sentences ...).
// The merge is inherently sequential per piece (one lowest-rank pair per round, tiktoken.cpp:322-343), so
// the kernel buys throughput with pieces in flight: parts live in LDS, kept dense (a merge shifts the tail
// left by one), and a piece gets a 16-lane group (<= 256 B, four pieces per wavefront) or a whole wavefront
// (<= 1024 B).  Per round: strided min over the rank array, shuffle min-reduce (key = rank<<10 | position,
// so ties go left), shift, two pair-table probes.  Anything longer falls back to parts in an HBM pool.
constexpr int LP_SMALL = 256;    // bytes handled by a 16-lane group
constexpr int LP_MEDIUM = 1024;  // bytes handled by a wavefront
constexpr int LP_SLOTS = 16;     // ceil(LP_SMALL/16) == ceil(LP_MEDIUM/64)

template <int G>
__device__ __forceinline__ uint32_t lp_merge_lds(const Tables& T, volatile uint32_t* id, volatile uint32_t* rk, uint32_t m, int gl) {
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t q = gl; q + 1 < m; q += G) {
            const uint32_t r = rk[q];
            if (r != (uint32_t)NO_RANK) {
                const uint32_t key = (r << 10) | q;
                best = key < best ? key : best;
            }
        }
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(best, d, G);
            best = o < best ? o : best;
        }
        if (best == 0xFFFFFFFFu) break;
        const uint32_t w = best & 1023u, r = best >> 10;
        // close the gap left by the absorbed part w+1: read everything first, then write (lanes overlap)
        uint32_t ti[LP_SLOTS], tr[LP_SLOTS];
#pragma unroll
        for (int u = 0; u < LP_SLOTS; ++u) {
            const uint32_t q = w + 2 + gl + u * G;
            ti[u] = q < m ? id[q] : 0u;
            tr[u] = q < m ? rk[q] : 0u;
        }
#pragma unroll
        for (int u = 0; u < LP_SLOTS; ++u) {
            const uint32_t q = w + 2 + gl + u * G;
            if (q < m) { id[q - 1] = ti[u]; rk[q - 1] = tr[u]; }
        }
        --m;
        if (gl == 0) {
            id[w] = r;
            rk[w] = (w + 1 < m) ? (uint32_t)pair_lookup(T, r, id[w + 1]) : (uint32_t)NO_RANK;
        } else if (gl == 1 && w > 0) {
            rk[w - 1] = (uint32_t)pair_lookup(T, id[w - 1], r);
        }
    }
    return m;
}

// Round 6: EVERY pair of the lowest rank per round, as far as that is what the sequential loop does (one wavefront, parts dense in LDS,
// m <= 1024: td_small_encode's pieces above 64 bytes — a lone run of one letter, of blanks, of dashes was a chain of one merge per ~3 us:
// 'a' * 1000 2.9 ms where the reference's quadratic loop takes 0.37).  r = the lowest rank present; its pairs are taken greedily from the
// left (in a run of consecutive ones every second one: the others lose a part to their left neighbour's merge).  The sequential loop
// (tiktoken.cpp:322-343: lowest rank first, leftmost on ties) merges exactly these, in this order, as long as no pair the merges create —
// (part in front, merged part), where the part in front is itself merged when the pair two positions to the left was taken, and (merged
// part, the still unmerged part behind it) — ranks at or below r: such a pair lies LEFT of every rank-r pair still to come, so the loop
// would take it first.  A round applies the selected merges up to and including the first one that creates such a pair; the next round
// starts from the lowest rank again.  At least the plain sequential step per round, hundreds of merges on repetitive pieces.  The rule as
// a CPU model against the reference's loop, vocabularies with ranks out of merge order included: tools/sim_rank_batches.py,
// tests/test_rank_batches_model.py; this function against the compiled reference: tests/test_gpu_rank_batches.py.
// A lane owns sixteen consecutive positions; everything a round needs is read into registers first, the compacted arrays written last.
__device__ __forceinline__ uint32_t lp_merge_batched(const Tables& T, volatile uint32_t* vid, volatile uint32_t* vrk, uint32_t m, const int lane) {
    uint32_t* const id = const_cast<uint32_t*>(vid);
    uint32_t* const rk = const_cast<uint32_t*>(vrk);
    constexpr uint32_t NR = (uint32_t)NO_RANK;
    typedef const uint64_t __attribute__((address_space(1)))* gpair_t;
    gpair_t const ps = (gpair_t)(uintptr_t)T.pair_slots;
    const uint32_t q0 = 16u * (uint32_t)lane;
    auto wave_min = [](uint32_t v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = o < v ? o : v; }
        return v;
    };
    while (m >= 2u) {
        uint32_t iv[16], rv[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint4 a4 = reinterpret_cast<const uint4*>(id + q0)[t], b4 = reinterpret_cast<const uint4*>(rk + q0)[t];
            iv[4 * t] = a4.x; iv[4 * t + 1] = a4.y; iv[4 * t + 2] = a4.z; iv[4 * t + 3] = a4.w;
            rv[4 * t] = b4.x; rv[4 * t + 1] = b4.y; rv[4 * t + 2] = b4.z; rv[4 * t + 3] = b4.w;
        }
        uint32_t best = NR;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (q0 + (uint32_t)j + 1u >= m) rv[j] = NR;  // (no pair starts at the last part or behind it)
            best = rv[j] < best ? rv[j] : best;
        }
        const uint32_t r = wave_min(best);
        if (r >= NR) break;
        uint32_t cand = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) cand |= (rv[j] == r ? 1u : 0u) << j;
        // the run of rank-r pairs that reaches my first position from the left: odd or even?  (a lane that is all candidates hands the
        // question on: sixteen is even)
        const uint32_t top = cand == 0xFFFFu ? 16u : (uint32_t)__clz(~(cand << 16));  // candidates at my upper end
        const uint64_t full = __ballot(cand == 0xFFFFu), odd = __ballot((top & 1u) != 0u);
        const uint64_t below = ~full & ((1ull << lane) - 1ull);
        uint32_t par = below ? (uint32_t)((odd >> (63 - __clzll((long long)below))) & 1ull) : 0u;
        uint32_t sel = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = (cand >> j) & 1u;
            sel |= (c & (par ^ 1u)) << j;
            par = c ? (par ^ 1u) : 0u;
        }
        // what the neighbours hold
        uint32_t sel_prev = (uint32_t)__shfl_up((int)sel, 1), iv15_prev = (uint32_t)__shfl_up((int)iv[15], 1);
        if (lane == 0) { sel_prev = 0; iv15_prev = 0; }
        const uint32_t iv0_next = (uint32_t)__shfl_down((int)iv[0], 1), iv1_next = (uint32_t)__shfl_down((int)iv[1], 1);
        // pass A: the two pairs every selected merge creates; positions 2 s and 2 s + 1 cannot both be selected: a slot per pair of positions
        uint32_t L8[8], R8[8], vpos = 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint64_t eL1[4], eL2[4], eR1[4], eR2[4];
            uint32_t pv[4], nx[4];
            bool hp[4], hn[4], has[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s8 = 4 * h + u, ja = 2 * s8, jb = ja + 1;
                const bool ta = (sel >> ja) & 1u, tb = (sel >> jb) & 1u;
                has[u] = ta || tb;
                const uint32_t q = q0 + (uint32_t)(tb ? jb : ja);
                const bool ps_a = s8 >= 1 ? ((sel >> (ja - 2 >= 0 ? ja - 2 : 0)) & 1u) : ((sel_prev >> 14) & 1u);
                const bool ps_b = s8 >= 1 ? ((sel >> (jb - 2)) & 1u) : ((sel_prev >> 15) & 1u);
                const uint32_t pp_a = s8 >= 1 ? iv[ja - 1 >= 0 ? ja - 1 : 0] : iv15_prev, pp_b = iv[ja];
                const uint32_t nn_a = ja + 2 <= 15 ? iv[ja + 2 <= 15 ? ja + 2 : 15] : iv0_next;
                const uint32_t nn_b = jb + 2 <= 15 ? iv[jb + 2 <= 15 ? jb + 2 : 15] : (jb + 2 == 16 ? iv0_next : iv1_next);
                pv[u] = (tb ? ps_b : ps_a) ? r : (tb ? pp_b : pp_a);
                nx[u] = tb ? nn_b : nn_a;
                hp[u] = has[u] && q > 0u;
                hn[u] = has[u] && q + 2u < m;
                eL1[u] = eL2[u] = eR1[u] = eR2[u] = PAIR_EMPTY;
                if (hp[u]) { eL1[u] = ps[hash_pair(pv[u], r) & T.pair_mask]; eL2[u] = ps[hash_pair2(pv[u], r) & T.pair_mask]; }
                if (hn[u]) { eR1[u] = ps[hash_pair(r, nx[u]) & T.pair_mask]; eR2[u] = ps[hash_pair2(r, nx[u]) & T.pair_mask]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s8 = 4 * h + u;
                L8[s8] = hp[u] ? (uint32_t)pair_match(eL1[u], eL2[u], pv[u], r) : NR;
                R8[s8] = hn[u] ? (uint32_t)pair_match(eR1[u], eR2[u], r, nx[u]) : NR;
                if (has[u] && (L8[s8] <= r || R8[s8] <= r)) {
                    const uint32_t q = q0 + (uint32_t)(((sel >> (2 * s8 + 1)) & 1u) ? 2 * s8 + 1 : 2 * s8);
                    vpos = q < vpos ? q : vpos;
                }
            }
        }
        const uint32_t cut = wave_min(vpos);  // the first merge that turns the sequential order elsewhere: applied, and the last one of the round
        uint32_t app = sel;
        if (cut < q0) app = 0;
        else if (cut - q0 < 15u) app &= (2u << (cut - q0)) - 1u;
        // pass B: the compacted arrays
        uint32_t app_prev = (uint32_t)__shfl_up((int)app, 1), app_next = (uint32_t)__shfl_down((int)app, 1);
        const uint32_t L_next0 = (uint32_t)__shfl_down((int)L8[0], 1);
        if (lane == 0) app_prev = 0;
        if (lane == 63) app_next = 0;
        const uint32_t absorbed = ((app << 1) | (app_prev >> 15)) & 0xFFFFu;
        const uint32_t napp = (uint32_t)__popc(app);
        const uint32_t incl = wave_incl_scan(napp, lane);
        const uint32_t before = incl - napp, total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        wave_sync_lds();  // (every lane has read its parts: the arrays may change now)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t q = q0 + (uint32_t)j;
            if (q < m && !((absorbed >> j) & 1u)) {
                const bool mine = (app >> j) & 1u;
                uint32_t nid, nrk;
                if (mine) {  // the pair that starts at a merged part: with the next merged part (that one's pair in front) or with the unmerged part behind
                    nid = r;
                    const bool next_merged = j + 2 <= 15 ? ((app >> (j + 2 <= 15 ? j + 2 : 15)) & 1u) : ((app_next >> (j + 2 - 16)) & 1u);
                    nrk = next_merged ? (j + 2 <= 15 ? L8[(j + 2 <= 15 ? j + 2 : 15) >> 1] : L_next0) : R8[j >> 1];
                } else {
                    nid = iv[j];
                    const bool next_merged = j + 1 <= 15 ? ((app >> (j + 1 <= 15 ? j + 1 : 15)) & 1u) : (app_next & 1u);
                    nrk = next_merged ? (j + 1 <= 15 ? L8[(j + 1 <= 15 ? j + 1 : 15) >> 1] : L_next0) : rv[j];
                }
                const uint32_t nq = q - (before + (uint32_t)__popc(app & ((1u << j) - 1u)));
                id[nq] = nid;
                rk[nq] = nrk;
            }
        }
        m -= total;
        wave_sync_lds();
    }
    return m;
}

// one group of G lanes handles entry j entirely in LDS; returns through the entry + tile_extra
template <int G>
__device__ __forceinline__ void lp_do_piece(const EncodeArgs& a, const Tables& T, uint32_t j, volatile uint32_t* id,
                                            volatile uint32_t* rk, int gl) {
    const int64_t gs = a.long_list[j].gs;
    const uint32_t len = a.long_list[j].len;
    const uint8_t* p = a.text + gs;
    int32_t whole = NO_RANK;
    if (a.use_fastpath && len <= T.max_token_len) {  // whole-piece table first (CoreBPE::encode, tiktoken.cpp:209-215)
        if (gl == 0) {
            auto get = [p](uint32_t k) { return (uint32_t)p[k]; };
            whole = piece_lookup(T, hash_bytes(get, len), len, get);
        }
        whole = __shfl(whole, 0, G);
    }
    uint32_t m;
    if (whole != NO_RANK) {
        m = 1;
        if (gl == 0) id[0] = (uint32_t)whole;
    } else {
        for (uint32_t q = gl; q < len; q += G) {
            const uint32_t b = p[q];
            id[q] = (uint32_t)T.byte_id[b];
            rk[q] = (q + 1 < len) ? (uint32_t)T.byte_pair[(b << 8) | p[q + 1]] : (uint32_t)NO_RANK;
        }
        m = lp_merge_lds<G>(T, id, rk, len, gl);
    }
    unsigned long long off = 0;
    if (gl == 0) off = atomicAdd(a.pool_used, (unsigned long long)m);
    off = __shfl(off, 0, G);
    if (off + m > a.pool_cap) {
        if (gl == 0) raise(a, TD_E_SCRATCH, gs);
        return;
    }
    for (uint32_t q = gl; q < m; q += G) {
        const uint32_t v = id[q];
        if ((int32_t)v >= T.pseudo_base) raise(a, TD_E_UNKNOWN_BYTE, gs);
        a.pool[off + q] = v;
    }
    if (gl == 0) {
        a.long_list[j].ntok = m;
        a.long_list[j].pool_off = off;
        if (m > 1) atomicAdd(&a.tile_extra[gs / K_TILE], m - 1);
    }
}

// Set-up of a linked piece with character seeds (td_common.h): a lane takes 16 CONSECUTIVE bytes and every step is a batch of
// independent loads, because a first form — every position asking cseed_part_at about itself, the part in front of it and the part
// behind it, one dependent load after the other — cost more than the rounds the seeds save (td_long_pieces 0.94 -> 1.13 ms per 256 MiB
// of mixed-script text).  Pass 1: is a seeded character starting at each of my bytes?  (the table entries of all sixteen, then their two
// neighbour-byte words, then the verdicts: to rk[], with the byte's own id in id[]).  Pass 2, behind a fence: what stands in front of
// and behind each of my part starts, read from the verdicts (all reads into registers, another fence), then the pair ranks of all my
// parts as one batch of lookups, then the writes.  Returns the number of parts.  (= cseed_part_at at every byte; tests/test_char_seeds.py
// holds that function to the reference's loop, the GPU parity tests hold this one to the reference.)
#ifndef TD_LP_SEED_INLINE
#define TD_LP_SEED_INLINE __forceinline__  // (__noinline__: 71 -> 105 spilled VGPRs and 448 B of scratch in td_long_pieces — measured at the compiler, round 6)
#endif
constexpr int LP_TINY = 128;      // eight lanes per piece up to here
constexpr int LP_LINKED = 255;    // sixteen lanes per piece up to here (links are bytes)
constexpr uint32_t LP_END = 255;  // "no neighbour" in the link arrays
constexpr uint32_t LPS_SEEDED = 0x80000000u;  // verdict: a seeded character starts here | its length << 21 | its id
template <int G>
__device__ TD_LP_SEED_INLINE uint32_t lp_seed_setup(const EncodeArgs& a, const Tables& T, int64_t gs, uint32_t len, uint32_t* id, uint32_t* rk,
                                                  uint8_t* nx, uint8_t* pv, int gl) {
    const uint32_t c = (uint32_t)gl * 16u;
    uint32_t w[5];
    load_piece_window(a.text, a.n, gs + (int64_t)c - 1, w);  // bytes c - 1 .. c + 18 of the piece (what lies outside the piece is never looked at)
#define LPS_W(i) ((w[(i) >> 2] >> (8 * ((i) & 3))) & 0xFFu)  /* byte c - 1 + i */
    uint32_t dec[16], bid[16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // (eight bytes at a time: the batches' registers)
        uint64_t e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = 8 * h + u;
            const uint32_t q = c + jj, b0 = LPS_W(jj + 1), b1 = LPS_W(jj + 2), b2 = LPS_W(jj + 3);
            const uint32_t k = b0 < 0xE0u ? 2u : 3u;
            uint32_t cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
            bool ok = q < len && b0 >= 0xC2u && b0 < 0xF0u && q + k <= len && (b1 & 0xC0u) == 0x80u;
            if (k == 3u) {
                cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
                ok = ok && (b2 & 0xC0u) == 0x80u && cp >= 0x800u;
            }
            e[u] = ok ? T.cseed[cp] : 0ull;
            bid[jj] = q < len ? (uint32_t)T.byte_id[b0] : 0u;
        }
        uint32_t pmw[8], nmw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = 8 * h + u;
            const uint32_t q = c + jj, k = LPS_W(jj + 1) < 0xE0u ? 2u : 3u;
            const uint32_t pb = LPS_W(jj), nb = k == 2u ? LPS_W(jj + 3) : LPS_W(jj + 4);
            const bool v = (e[u] & CS_VALID) != 0ull;
            pmw[u] = (v && q > 0u) ? T.cseed_pm[(uint32_t)((e[u] >> 21) & 0xFFu) * 8u + (pb >> 5)] : 0u;
            nmw[u] = (v && q + k < len) ? T.cseed_nm[(uint32_t)((e[u] >> 29) & 0xFFu) * 8u + (nb >> 5)] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jj = 8 * h + u;
            const uint32_t q = c + jj, k = LPS_W(jj + 1) < 0xE0u ? 2u : 3u;
            const uint32_t pb = LPS_W(jj), nb = k == 2u ? LPS_W(jj + 3) : LPS_W(jj + 4);
            const bool v = (e[u] & CS_VALID) != 0ull && !(((pmw[u] >> (pb & 31u)) | (nmw[u] >> (nb & 31u))) & 1u);
            dec[jj] = v ? (LPS_SEEDED | (k << 21) | ((uint32_t)e[u] & 0x1FFFFFu)) : 0u;
            if (q < len) { rk[q] = dec[jj]; id[q] = bid[jj]; }
        }
    }
    wave_sync_lds();
    // pass 2: my part starts, the parts behind and in front of them
    uint32_t my_id[16], nid[16], meta[16];  // meta: next part start | previous part start << 8 | 1 << 16 (a part starts here) | 1 << 17 (both parts single bytes)
    uint32_t starts = 0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const uint32_t q = c + jj;
        my_id[jj] = 0; nid[jj] = 0; meta[jj] = 0;
        if (q < len) {
            const uint32_t dm1 = jj >= 1 ? dec[jj >= 1 ? jj - 1 : 0] : (q >= 1u ? rk[q - 1u] : 0u);
            const uint32_t dm2 = jj >= 2 ? dec[jj >= 2 ? jj - 2 : 0] : (q >= 2u ? rk[q - 2u] : 0u);
            const uint32_t dm3 = jj >= 3 ? dec[jj >= 3 ? jj - 3 : 0] : (q >= 3u ? rk[q - 3u] : 0u);
            const bool inside = (dm1 & LPS_SEEDED) || ((dm2 & LPS_SEEDED) && ((dm2 >> 21) & 3u) == 3u);
            if (!inside) {
                const uint32_t d0 = dec[jj];
                const uint32_t k0 = (d0 & LPS_SEEDED) ? ((d0 >> 21) & 3u) : 1u;
                const uint32_t nq = q + k0;
                uint32_t dn = 0, nb_id = 0;
                if (nq < len) { dn = rk[nq]; nb_id = id[nq]; }
                my_id[jj] = (d0 & LPS_SEEDED) ? (d0 & 0x1FFFFFu) : bid[jj];
                nid[jj] = (dn & LPS_SEEDED) ? (dn & 0x1FFFFFu) : nb_id;
                // the part in front: the byte in front, or the seeded character it is the inside of
                const uint32_t pq = q == 0u ? LP_END : (dm2 & LPS_SEEDED) ? q - 2u : ((dm3 & LPS_SEEDED) && ((dm3 >> 21) & 3u) == 3u) ? q - 3u : q - 1u;
                meta[jj] = (nq < len ? nq : LP_END) | (pq << 8) | (1u << 16) | ((!(d0 & LPS_SEEDED) && !(dn & LPS_SEEDED)) ? (1u << 17) : 0u);
                ++starts;
            }
        }
    }
    wave_sync_lds();  // (every lane has read the verdicts: rk[] and id[] take their final contents now)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        uint64_t e1[4], e2[4];
        uint32_t bp[4];
        typedef const uint64_t __attribute__((address_space(1)))* gpair_t;
        gpair_t const ps = (gpair_t)(uintptr_t)T.pair_slots;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = 4 * h + u;
            const bool st = (meta[jj] >> 16) & 1u, has_next = (meta[jj] & 0xFFu) != LP_END, plain = (meta[jj] >> 17) & 1u;
            e1[u] = PAIR_EMPTY; e2[u] = PAIR_EMPTY; bp[u] = (uint32_t)NO_RANK;
            if (st && has_next) {
                if (plain) bp[u] = (uint32_t)T.byte_pair[(LPS_W(jj + 1) << 8) | LPS_W(jj + 2)];
                else { e1[u] = ps[hash_pair(my_id[jj], nid[jj]) & T.pair_mask]; e2[u] = ps[hash_pair2(my_id[jj], nid[jj]) & T.pair_mask]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = 4 * h + u;
            const uint32_t q = c + jj;
            if (q < len) {
                const bool st = (meta[jj] >> 16) & 1u, plain = (meta[jj] >> 17) & 1u;
                if (st) {
                    id[q] = my_id[jj];
                    rk[q] = plain ? bp[u] : (uint32_t)pair_match(e1[u], e2[u], my_id[jj], nid[jj]);
                    nx[q] = (uint8_t)(meta[jj] & 0xFFu);
                    pv[q] = (uint8_t)((meta[jj] >> 8) & 0xFFu);
                } else {
                    id[q] = TOK_NONE;
                    rk[q] = (uint32_t)NO_RANK;
                    nx[q] = (uint8_t)LP_END;
                    pv[q] = (uint8_t)LP_END;
                }
            }
        }
    }
#undef LPS_W
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) starts += (uint32_t)__shfl_xor((int)starts, d, G);
    return starts;
}

// Pieces of 65..LP_TINY bytes (the bulk of the long pieces: comment rulers, CJK sentences): eight lanes per piece,
// eight pieces per wavefront, parts as a doubly linked list in LDS — a merge touches a handful of entries instead of
// shifting the tail, so a round is: strided min over the rank array (dead parts carry NO_RANK), 3-step shuffle reduce,
// relink, two pair-table probes (one lane each).  The rounds are bound by the probe latency; pieces in flight per
// wavefront are what buys throughput.
template <int G>
__device__ __forceinline__ void lp_do_piece_linked(const EncodeArgs& a, const Tables& T, uint32_t j, volatile uint32_t* vid,
                                                   volatile uint32_t* vrk, volatile uint8_t* vnx, volatile uint8_t* vpv, int grp, int gl) {
    // (plain pointers + a wavefront-scope fence at the end of every round: as volatile arrays every one of the sixteen reads of
    // a lane's stripe was waited for on its own)
    uint32_t* const id = const_cast<uint32_t*>(vid);
    uint32_t* const rk = const_cast<uint32_t*>(vrk);
    uint8_t* const nx = const_cast<uint8_t*>(vnx);
    uint8_t* const pv = const_cast<uint8_t*>(vpv);
    const int64_t gs = a.long_list[j].gs;
    const uint32_t len = a.long_list[j].len;
    const uint8_t* p = a.text + gs;
    int32_t whole = NO_RANK;
    if (a.use_fastpath && len <= T.max_token_len) {  // whole-piece table first (CoreBPE::encode, tiktoken.cpp:209-215)
        if (gl == 0) {
            auto get = [p](uint32_t k) { return (uint32_t)p[k]; };
            whole = piece_lookup(T, hash_bytes(get, len), len, get);
        }
        whole = __shfl(whole, 0, G);
    }
    uint32_t m = len;
    if (whole != NO_RANK) {
        m = 1;
        if (gl == 0) { id[0] = (uint32_t)whole; nx[0] = (uint8_t)LP_END; }
    } else {
        // Round 5: characters that may be entered whole (td_common.h: character seeds) are ONE part from the start — a CJK sentence
        // of 27 characters begins with ~30 parts instead of 81.  Every position settles by itself what stands there (a byte, a seeded
        // character, the inside of one), what part stands in front of it and what part follows: the answers depend on the piece's
        // bytes only, so the lanes agree without talking.  Pieces without a byte above 0xC1 take the plain set-up.
        bool multi = false;
        if (T.cseed) {
            bool mine = false;
            for (uint32_t q = gl; q < len; q += G) mine = mine || p[q] >= 0xC2u;
            multi = (((uint32_t)(__ballot(mine) >> (grp * G))) & ((1u << G) - 1u)) != 0u;
        }
        if (multi) {
            m = lp_seed_setup<G>(a, T, gs, len, id, rk, nx, pv, gl);
        } else
        for (uint32_t q = gl; q < len; q += G) {
            const uint32_t b = p[q];
            id[q] = (uint32_t)T.byte_id[b];
            rk[q] = (q + 1 < len) ? (uint32_t)T.byte_pair[(b << 8) | p[q + 1]] : (uint32_t)NO_RANK;
            nx[q] = (uint8_t)((q + 1 < len) ? q + 1 : LP_END);
            pv[q] = (uint8_t)(q ? q - 1 : LP_END);
        }
        // A round is one dependent round trip to the pair table (about a microsecond), and a piece of a hundred bytes takes a
        // hundred of them when every round applies ONE merge.  The sequential rule (lowest rank first, leftmost on ties;
        // tiktoken.cpp:334-342) is kept while applying up to LP_K merges per round: the candidates are the LP_K lowest keys
        // in order; all their new pairs are looked up together (two lookups each, a lane each); candidate c is applied only
        // if it touches neither the parts nor the neighbours of the ones applied before it this round (so its lookups,
        // made on the state in front of the round, still hold) and its key is lower than every key those created — then it
        // is exactly what the sequential loop would merge next.  The first candidate that fails ends the round.
#ifndef TD_LP_K
#define TD_LP_K 2
#endif
        constexpr int LP_K = TD_LP_K;
        wave_sync_lds();  // (the parts set up above are visible to the whole group)
        static_assert(2 * LP_K <= G, "a lane per lookup");
        constexpr uint32_t INF = 0xFFFFFFFFu;
        auto gmin = [&](uint32_t v) {
#pragma unroll
            for (int d = G / 2; d >= 1; d >>= 1) {
                const uint32_t o = __shfl_xor(v, d, G);
                v = o < v ? o : v;
            }
            return v;
        };
        for (;;) {
            // my stripe's two lowest keys (rank << 8 | position: ties go left, like the reference's strict '<' scan)
            // (the groups of a wavefront start their stripes G x group slots apart: their arrays lie a multiple of 64 words apart, so with
            // every group at slot gl + k G the lanes of ALL groups asked the same G banks — 49 % of this kernel's LDS cycles were conflicts)
            uint32_t b0 = INF, b1 = INF;
            const uint32_t nk = (len + G - 1) / G;  // (> the group's number: len > 8 G)
            for (uint32_t kk = 0; kk < nk; ++kk) {
                uint32_t k = kk + (uint32_t)grp;
                if (k >= nk) k -= nk;
                const uint32_t q = (uint32_t)gl + k * G;
                const uint32_t r = q < len ? rk[q] : (uint32_t)NO_RANK;
                if (r != (uint32_t)NO_RANK) {
                    const uint32_t key = (r << 8) | q;
                    if (key < b0) { b1 = b0; b0 = key; } else if (key < b1) b1 = key;
                }
            }
            // the group's lowest keys in order, until a lane has given both of its own (its third is not known)
            uint32_t ck[LP_K];
            int nc = 0;
            uint32_t used = 0;
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                ck[c] = INF;
                if (nc == c) {
                    const uint32_t cur = used == 0 ? b0 : used == 1 ? b1 : INF;
                    const uint32_t g = gmin(cur);
                    if (g != INF) {
                        const bool mine = cur == g;
                        if (mine) ++used;
                        ck[c] = g;
                        nc = c + 1;
                        // (a lane that has given both of its keys may hold the next lowest one too: no further candidates)
                        const uint32_t stop = gmin((mine && used == 2) ? 0u : 1u);
                        if (stop == 0u && c + 1 < LP_K) { nc = -(c + 1); }
                    }
                }
            }
            if (nc < 0) nc = -nc;
            if (nc == 0) break;
            // the candidates' parts and neighbours (state in front of the round; the same values in every lane of the group)
            uint32_t cw[LP_K], cr[LP_K], cnx[LP_K], cnn[LP_K], cpw[LP_K], cidn[LP_K], cidp[LP_K];
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                cw[c] = ck[c] & 255u; cr[c] = ck[c] >> 8;
                cnx[c] = LP_END; cnn[c] = LP_END; cpw[c] = LP_END; cidn[c] = 0; cidp[c] = 0;
                if (c < nc) {
                    cnx[c] = nx[cw[c]];
                    cnn[c] = nx[cnx[c]];
                    cpw[c] = pv[cw[c]];
                    cidn[c] = cnn[c] != LP_END ? id[cnn[c]] : 0u;
                    cidp[c] = cpw[c] != LP_END ? id[cpw[c]] : 0u;
                }
            }
            // the lookups: lane 2c the pair (merged part, part behind it), lane 2c + 1 the pair (part in front of it, merged part)
            uint32_t mine_res = (uint32_t)NO_RANK;
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                if (c < nc && gl == 2 * c && cnn[c] != LP_END) mine_res = (uint32_t)pair_lookup(T, cr[c], cidn[c]);
                if (c < nc && gl == 2 * c + 1 && cpw[c] != LP_END) mine_res = (uint32_t)pair_lookup(T, cidp[c], cr[c]);
            }
            uint32_t rw[LP_K], rp[LP_K];
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                rw[c] = __shfl(mine_res, 2 * c, G);
                rp[c] = __shfl(mine_res, 2 * c + 1, G);
            }
            // which candidates are applied
            uint32_t minnew = INF;
            int napply = 0;
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                if (c < nc && napply == c) {
                    bool ok = c == 0 || ck[c] < minnew;
#pragma unroll
                    for (int i = 0; i < LP_K; ++i) {
                        if (i < c) {
                            // my parts against the earlier one's parts and neighbours; my neighbours against its parts
                            const uint32_t a0 = cw[i], a1 = cnx[i], n0 = cpw[i], n1 = cnn[i];
                            const bool hit = cw[c] == a0 || cw[c] == a1 || cw[c] == n0 || cw[c] == n1 || cnx[c] == a0 || cnx[c] == a1 ||
                                             cnx[c] == n0 || cnx[c] == n1 || (cpw[c] != LP_END && (cpw[c] == a0 || cpw[c] == a1)) ||
                                             (cnn[c] != LP_END && (cnn[c] == a0 || cnn[c] == a1));
                            ok = ok && !hit;
                        }
                    }
                    if (ok) {
                        napply = c + 1;
                        const uint32_t kw = (cnn[c] != LP_END && rw[c] != (uint32_t)NO_RANK) ? ((rw[c] << 8) | cw[c]) : INF;
                        const uint32_t kp = (cpw[c] != LP_END && rp[c] != (uint32_t)NO_RANK) ? ((rp[c] << 8) | cpw[c]) : INF;
                        minnew = kw < minnew ? kw : minnew;
                        minnew = kp < minnew ? kp : minnew;
                    }
                }
            }
            // apply: lane c does candidate c (they touch different entries)
#pragma unroll
            for (int c = 0; c < LP_K; ++c) {
                if (c < napply && gl == c) {
                    const uint32_t w = cw[c], nxt = cnx[c], nn = cnn[c], pw = cpw[c];
                    id[w] = cr[c];
                    rk[nxt] = (uint32_t)NO_RANK;
                    id[nxt] = TOK_NONE;
                    nx[w] = (uint8_t)nn;
                    if (nn != LP_END) pv[nn] = (uint8_t)w;
                    rk[w] = (nn != LP_END) ? rw[c] : (uint32_t)NO_RANK;
                    if (pw != LP_END) rk[pw] = rp[c];
                }
            }
            m -= (uint32_t)napply;
            wave_sync_lds();
        }
    }
    wave_sync_lds();
    unsigned long long off = 0;
    if (gl == 0) off = atomicAdd(a.pool_used, (unsigned long long)m);
    off = __shfl(off, 0, G);
    if (off + m > a.pool_cap) {
        if (gl == 0) raise(a, TD_E_SCRATCH, gs);
        return;
    }
    // surviving parts in position order: eight positions at a time, rank inside the group's byte of the ballot
    uint32_t run = 0;
    const uint32_t span = (m == 1 && whole != NO_RANK) ? 1u : len;
    for (uint32_t q0 = 0; q0 < span; q0 += G) {
        const uint32_t q = q0 + gl;
        const uint32_t v = (q < span) ? id[q] : TOK_NONE;
        const bool alive = v != TOK_NONE;
        const uint32_t bits = (uint32_t)(__ballot(alive) >> (grp * G)) & ((1u << G) - 1u);
        if (alive) {
            if ((int32_t)v >= T.pseudo_base) raise(a, TD_E_UNKNOWN_BYTE, gs);
            a.pool[off + run + __popc(bits & ((1u << gl) - 1u))] = v;
        }
        run += __popc(bits);
    }
    if (gl == 0) {
        a.long_list[j].ntok = m;
        a.long_list[j].pool_off = off;
        if (m > 1) atomicAdd(&a.tile_extra[gs / K_TILE], m - 1);
    }
}

constexpr int LP_WAVE_WORDS = 8 * (2 * LP_TINY + 2 * LP_TINY / 4);  // LDS words per wavefront: 8 pieces x (ids, ranks, links)
static_assert(LP_WAVE_WORDS >= 2 * LP_MEDIUM, "the wavefront-per-piece pass reuses the same LDS");
// ascending bitonic sort of one key per lane across the wavefront
__device__ __forceinline__ uint32_t wave_sort_asc(uint32_t key, const int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)key, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            const uint32_t mn = o < key ? o : key, mx = o < key ? key : o;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}
constexpr int LONG_LDS_WORDS = 4 * LP_WAVE_WORDS;
__device__ __forceinline__ void long_pieces_body(const EncodeArgs& a, const uint32_t bid, const uint32_t nb, uint32_t* const lds) {
    uint32_t (*const s_parts)[LP_WAVE_WORDS] = reinterpret_cast<uint32_t (*)[LP_WAVE_WORDS]>(lds);  // per wavefront: ids | ranks (| links)
    const Tables T = uniform_tables(a.Tp);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nlong = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    const uint32_t wave_global = bid * (blockDim.x >> 6) + wv;
    const uint32_t nwaves = nb * (blockDim.x >> 6);

#ifndef TD_LP_CHUNKED
    // pass 0: pieces <= 128 B, an 8-lane group each (linked parts)
    {
        const int grp = lane >> 3, gl = lane & 7;
        volatile uint32_t* id = &s_parts[wv][grp * (LP_WAVE_WORDS / 8)];
        volatile uint32_t* rk = id + LP_TINY;
        volatile uint8_t* nx = reinterpret_cast<volatile uint8_t*>(rk + LP_TINY);
        volatile uint8_t* pv = nx + LP_TINY;
        // (rows drawn from a counter instead of dealt, as the fused loop draws its tiles: 0.94 -> 0.99 ms on mixed-script text)
        for (uint32_t j = wave_global * 8 + grp; j < nlong; j += nwaves * 8)
            if (a.long_list[j].len <= LP_TINY) lp_do_piece_linked<8>(a, T, j, id, rk, nx, pv, grp, gl);
    }
    // pass 1: pieces <= 255 B, a 16-lane group each (linked parts); 256 B: the dense variant
    {
        const int grp = lane >> 4, gl = lane & 15;
        volatile uint32_t* id = &s_parts[wv][grp * (LP_WAVE_WORDS / 4)];
        volatile uint32_t* rk = id + LP_SMALL;
        volatile uint8_t* nx = reinterpret_cast<volatile uint8_t*>(rk + LP_SMALL);
        volatile uint8_t* pv = nx + LP_SMALL;
        for (uint32_t j = wave_global * 4 + grp; j < nlong; j += nwaves * 4) {
            const uint32_t len = a.long_list[j].len;
            if (len > LP_TINY && len <= LP_LINKED) lp_do_piece_linked<16>(a, T, j, id, rk, nx, pv, grp, gl);
            else if (len == LP_SMALL) lp_do_piece<16>(a, T, j, id, rk, gl);
        }
    }
#else
    // (-DTD_LP_CHUNKED, measured and NOT in: mixed-script text 0.81 -> 0.93 ms per 256 MiB, the code file set 0.45 -> 0.50 — a chunk's pieces are one
    // wavefront's, and the sort's gain in lockstep does not make up for the coarser deal; both forms in one kernel: 72 -> 239 spilled VGPRs)
    // The list is taken 64 entries (a CHUNK) at a time: a lane reads an entry's length, the wavefront SORTS its chunk by length (bitonic, six
    // stages of shuffles) and deals the pieces to its lane groups in that order — the eight pieces a wavefront merges side by side advance
    // in lockstep (a row takes as long as its slowest piece), and pieces of one length need about the same number of rounds; dealt in list
    // order a row held lengths from 65 to 128 bytes and 25 of 64 lanes were active on average.  Chunks are DRAWN from a counter (the draw
    // for the next chunk is issued in front of this chunk's work, its answer read behind it): dealt by stride the wavefronts were done at
    // very different times — 1.95 of 4 wavefronts per SIMD resident on average over the kernel (round 5's counters).
    // (a chunk is 8 .. 64 entries: at least four chunks per wavefront of the grid — with 64 entries a chunk, 256 MiB of mixed-script text
    // (132 000 long pieces, 4096 wavefronts) kept half of the wavefronts without any: 0.81 -> 1.19 ms)
    uint32_t csh = 6;
    while (csh > 3 && (nlong >> csh) < 4u * nwaves) --csh;
    const uint32_t C = 1u << csh;
    const uint32_t nchunks = (nlong + C - 1u) >> csh;
    uint32_t chunk = wave_global;
    while (chunk < nchunks) {
        uint32_t drawn = 0;
        if (lane == 0) drawn = atomicAdd(a.lp_next, 1u);
        const uint32_t j0 = (chunk << csh) + (uint32_t)lane;
        const uint32_t len_l = ((uint32_t)lane < C && j0 < nlong) ? a.long_list[j0].len : 0u;
        constexpr uint32_t INF = 0xFFFFFFFFu;
        // pass 0: pieces <= 128 B, an 8-lane group each (linked parts)
        {
            const int grp = lane >> 3, gl = lane & 7;
            volatile uint32_t* id = &s_parts[wv][grp * (LP_WAVE_WORDS / 8)];
            volatile uint32_t* rk = id + LP_TINY;
            volatile uint8_t* nx = reinterpret_cast<volatile uint8_t*>(rk + LP_TINY);
            volatile uint8_t* pv = nx + LP_TINY;
            const uint32_t key = wave_sort_asc((len_l != 0u && len_l <= (uint32_t)LP_TINY) ? ((len_l << 6) | (uint32_t)lane) : INF, lane);
            const uint32_t cnt = (uint32_t)__popcll((unsigned long long)__ballot(key != INF));
            for (uint32_t r = 0; r * 8u < cnt; ++r) {
                const uint32_t k = (uint32_t)__shfl((int)key, (int)(r * 8u) + grp);
                if (k != INF) lp_do_piece_linked<8>(a, T, (chunk << csh) + (k & 63u), id, rk, nx, pv, grp, gl);
            }
        }
        wave_sync_lds();
        // pass 1: pieces <= 255 B, a 16-lane group each (linked parts); 256 B: the dense variant
        {
            const int grp = lane >> 4, gl = lane & 15;
            volatile uint32_t* id = &s_parts[wv][grp * (LP_WAVE_WORDS / 4)];
            volatile uint32_t* rk = id + LP_SMALL;
            volatile uint8_t* nx = reinterpret_cast<volatile uint8_t*>(rk + LP_SMALL);
            volatile uint8_t* pv = nx + LP_SMALL;
            const uint32_t key = wave_sort_asc((len_l > (uint32_t)LP_TINY && len_l <= (uint32_t)LP_SMALL) ? ((len_l << 6) | (uint32_t)lane) : INF, lane);
            const uint32_t cnt = (uint32_t)__popcll((unsigned long long)__ballot(key != INF));
            for (uint32_t r = 0; r * 4u < cnt; ++r) {
                const uint32_t k = (uint32_t)__shfl((int)key, (int)(r * 4u) + grp);
                if (k != INF) {
                    const uint32_t len = k >> 6, j = (chunk << csh) + (k & 63u);
                    if (len <= (uint32_t)LP_LINKED) lp_do_piece_linked<16>(a, T, j, id, rk, nx, pv, grp, gl);
                    else lp_do_piece<16>(a, T, j, id, rk, gl);
                }
            }
        }
        wave_sync_lds();
        chunk = nwaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
    }
#endif
    // pass 2: pieces <= 1024 B, a wavefront each — dealt by stride over the whole list (they come in clusters — a file of emoji
    // sequences — and a chunk's wavefront would take its cluster alone)
    {
        volatile uint32_t* id = &s_parts[wv][0];
        volatile uint32_t* rk = id + LP_MEDIUM;
        for (uint32_t j = wave_global; j < nlong; j += nwaves) {
            const uint32_t len = a.long_list[j].len;
            if (len > LP_SMALL && len <= LP_MEDIUM) lp_do_piece<64>(a, T, j, id, rk, lane);
        }
    }
    // (pieces above LP_MEDIUM bytes: td_giant_pieces)
}
__global__ __launch_bounds__(256, 4) void td_long_pieces(const EncodeArgs a) {  // (four wavefronts per SIMD is what the LDS allows: the registers must not allow less)
    __shared__ uint32_t s_lds[LONG_LDS_WORDS];
    long_pieces_body(a, blockIdx.x, gridDim.x, s_lds);
}

// ------------------------------------------------------------------ td_giant_pieces ---------
// Pieces above 1 KiB (a megabyte of one letter, of blanks, of DNA ...): the reference's merge loop is O(len^2) there
// (tiktoken.cpp:322-343: erase + full rescan per merge; 20 KB take 0.3 s, a megabyte hours), and so was round 1's path (one
// wavefront, lane 0 applying one merge per HBM round trip).  Here: ROUNDS over the whole piece, a 1024-thread workgroup per
// piece, parts dense in HBM (ids + pair ranks, two generations).  A round takes g = the lowest rank present and merges
// EVERY pair of rank g at once — in a run of consecutive g-pairs every second one, from the left, which is what the
// sequential leftmost-first rule does to the run — compacts, and looks the changed pairs up again.  That equals the
// sequential order as long as no pair that the sequential execution would see while it works through the g-pairs (the new
// neighbours of a merged part, including the transient neighbour that is itself about to be merged) ranks below g; every
// such pair is looked up and checked, and a round that fails the check is redone with only the leftmost g-pair merged (the
// plain sequential step, always exact).  'a' * 1 000 000: 16 rounds of sweeps over a shrinking array instead of 10^6 steps.
// Cost: (distinct ranks that occur) x (len / 1024) sweep steps — far from quadratic for the repetitive inputs that produce
// such pieces, still slow for a megabyte of random letters.
constexpr int GP_THREADS = 1024;
template <bool IS_MAX>
__device__ __forceinline__ int gp_wave_scan(int x) {  // inclusive add- or max-scan across the wavefront (DPP)
    constexpr int ROW_SHR1 = 0x111, ROW_SHR2 = 0x112, ROW_SHR4 = 0x114, ROW_SHR8 = 0x118, ROW_BCAST15 = 0x142, ROW_BCAST31 = 0x143;
    constexpr int ID = IS_MAX ? (int)0x80000000 : 0;
    auto op = [](int p, int q) { return IS_MAX ? (p > q ? p : q) : p + q; };
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_SHR1, 0xF, 0xF, false));
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_SHR2, 0xF, 0xF, false));
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_SHR4, 0xF, 0xF, false));
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_SHR8, 0xF, 0xF, false));
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_BCAST15, 0xA, 0xF, false));
    x = op(x, __builtin_amdgcn_update_dpp(ID, x, ROW_BCAST31, 0xC, 0xF, false));
    return x;
}
// inclusive scan over the 1024 threads of the workgroup; `total` = the last thread's value (two barriers)
template <bool IS_MAX>
__device__ __forceinline__ int gp_block_scan(int x, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int ID = IS_MAX ? (int)0x80000000 : 0;
    int y = gp_wave_scan<IS_MAX>(x);
    if (lane == 63) s_w[wv] = y;
    __syncthreads();
    int pre = ID, tot = ID;
#pragma unroll
    for (int w = 0; w < GP_THREADS / 64; ++w) {
        const int sw = s_w[w];
        if (w < wv) pre = IS_MAX ? (pre > sw ? pre : sw) : pre + sw;
        tot = IS_MAX ? (tot > sw ? tot : sw) : tot + sw;
    }
    __syncthreads();
    total = tot;
    return IS_MAX ? (y > pre ? y : pre) : y + pre;
}
__device__ __forceinline__ uint32_t gp_block_min(uint32_t x, uint32_t* s_m) {  // minimum over the workgroup (two barriers)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(x, d);
        x = o < x ? o : x;
    }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < GP_THREADS / 64; ++w) m = s_m[w] < m ? s_m[w] : m;
    __syncthreads();
    return m;
}

// Round 4: BANDS of ranks instead of one rank per round.  A round's candidates are the pairs that rank strictly below their
// left neighbour pair and not above their right one (in a run of equal ranks that starts so: every second one) — no two of
// them overlap.  The ones below a bound B merge together; B is lowered until (1) every pair below B that is left over is
// overlapped by a merging one (so the sequential loop, tiktoken.cpp:322-343, has nothing below B to merge but the selected
// pairs) and (2) every pair the merges create — and every transient pair the sequential order would see between two merges
// one part apart — ranks at or above B (so nothing new can come in between).  Then the selected merges do not interact
// and the result is the sequential one whatever their order.  A bound that ends up at the lowest rank present falls back to
// the plain sequential step (leftmost pair of the lowest rank).  Checked against the heap form of the reference's loop on
// the CPU first (1 600 random pieces over the Llama-4 vocabulary and over toy vocabularies whose ranks are NOT in merge order):
// a megabyte of random letters takes ~45 rounds instead of one per distinct rank (tens of thousands).
constexpr uint32_t GP_INF = 0x00FFFFFFu;   // rank word: low 24 bits = rank of the pair that starts here (GP_INF: none)
constexpr uint32_t GP_BASE = 0x40000000u;  //            the pair is a candidate of this round
constexpr int GP_E = 4;                    // parts per thread and step
constexpr uint32_t GP_STEP = GP_THREADS * GP_E;
// (the scope of the loads of what other lanes wrote: workgroup — a piece is one workgroup's, and a CU's waves share its L1;
// agent scope sends every one of these loads to the memory side of the fabric: 554 -> ? ms for a megabyte of random letters)
#ifndef GP_LD_SCOPE
#define GP_LD_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
// COOP (round 5, below): what the workgroups of a launch exchange is written and read at agent scope (write-through stores, loads
// that pass the L1 and the XCD's L2): no cache write-back or invalidation at the grid barriers — with release / acquire fences
// around them (buffer_wbl2 / buffer_inv by 4096 wavefronts) a barrier cost ~260 us, a megabyte of random letters 94 ms
template <bool COOP>
__device__ __forceinline__ uint32_t gp_ld(const uint32_t* q) {
    if constexpr (COOP) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(q, __ATOMIC_RELAXED, GP_LD_SCOPE);  // (written by other lanes of this workgroup)
}
template <bool COOP>
__device__ __forceinline__ void gp_st(uint32_t* q, uint32_t v) {
    if constexpr (COOP) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
}
__device__ __forceinline__ uint32_t gp_rank(const Tables& T, uint32_t l, uint32_t r) {
    const int32_t v = pair_lookup(T, l, r);
    return v == NO_RANK ? GP_INF : (uint32_t)v;
}

// Round 5: ONE piece over ALL workgroups of the launch (a megabyte of random letters was 0.55 s on one CU).
// The rule and its three kinds of sweeps are the ones above; what changes is who sweeps what.  The parts of a generation are cut
// into one stretch per workgroup (a multiple of GP_STEP positions; positions stay global, so a stretch reads its neighbours'
// border parts like its own); a phase ends in a grid barrier (one counter in the control block; the arrays are written and
// read at agent scope, see gp_ld / gp_st); the three workgroup-wide combinations become grid-wide ones through a slot per workgroup in HBM:
//   run start in front of a stretch   each workgroup's last run start -> barrier -> maximum over the workgroups in front
//   the bound's vmin / smax           a pair of slots per workgroup -> barrier -> every wavefront reduces the slots itself
//   compaction                        survivors counted per stretch -> barrier -> sum over the workgroups in front = where
//                                     the stretch's survivors go (a pass more than the single workgroup, which compacts in place)
// and the ranks of the pairs the merges create go to an array of their own (in place they would land in a stretch another
// workgroup has not read yet): 5 x len words of the pool instead of 4.  The slots are double-buffered by the parity of a
// sequence number every thread counts alike: between two uses of one parity lies the barrier of the other.
// A launch's workgroups are resident together (launch_encode: one per CU on half of the CUs; its first barrier finds out whether
// they really are, gp_grid_meet) — a later barrier that is not passed within GP_BAR_TIMEOUT (a fault) raises TD_E_HIP and lets
// every workgroup go.
constexpr int GP_MAX_BLOCKS = 256;
constexpr uint32_t GP_LIST_CAP = 1024;  // pieces for all workgroups together, per call (further ones: a workgroup each)
constexpr uint32_t GP_SCRATCH_WORDS = GP_LIST_CAP + 2u * GP_MAX_BLOCKS * 2u;  // the list | 2 parities x workgroups x 2 slots
static_assert(GP_SCRATCH_WORDS * 4u == (uint32_t)TD_GP_SCRATCH_BYTES, "td_kernels.h");
constexpr unsigned long long GP_BAR_TIMEOUT = 400000000ull;  // wall_clock64 ticks (100 MHz): 4 s
struct GpCoop {
    uint32_t* bar;     // a.gp_ctl[0]: arrivals (zero at launch; counts up)
    uint32_t* stop;    // a.gp_ctl[2]: a barrier timed out
    uint32_t* slots;
    uint32_t nblk, seq;
    int* s_flag;
};
__device__ __forceinline__ bool gp_grid_barrier(const GpCoop& c) {
    __builtin_amdgcn_s_waitcnt(0);  // (this wavefront's write-through stores have arrived)
    __syncthreads();
    if (threadIdx.x == 0) {
        // (arrival = release, leaving = acquire, agent scope — ADVICE r5: with relaxed orders the barrier held on this hardware, the
        // s_waitcnt + workgroup barrier in front and the agent-scope accesses of gp_ld / gp_st saw to that, but the HIP memory model did
        // not say so; -DTD_GP_BARRIER_RELAXED = the old orders, for the A/B in profiles/r6_10_…)
#ifndef TD_GP_BARRIER_RELAXED
        constexpr int GP_ARRIVE = __ATOMIC_RELEASE, GP_LEAVE = __ATOMIC_ACQUIRE;
#else
        constexpr int GP_ARRIVE = __ATOMIC_RELAXED, GP_LEAVE = __ATOMIC_RELAXED;
#endif
        const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, GP_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = (old / c.nblk + 1u) * c.nblk;
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(c.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if (wall_clock64() - t0 > GP_BAR_TIMEOUT) {
                __hip_atomic_store(c.stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (GP_LEAVE != __ATOMIC_RELAXED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (once, behind the polls)
        *c.s_flag = ok;
    }
    __syncthreads();
    return *c.s_flag != 0;
}
// The launch's FIRST grid barrier asks what all later ones rely on: are its workgroups resident at the same time?  Next to this
// step's own kernels they are (those end and make room).  Next to ANOTHER launch of this kernel they need not be — a second handle
// (td_clone) or process on the device with a listed piece of its own: each launch holds a part of the CUs and waits for the rest
// of its workgroups, which wait for a CU.  So the first barrier gives up after GP_MEET_TIMEOUT, for ALL workgroups alike: the one
// that runs out of patience sets GP_BAR_DEAD in the counter with a compare-and-swap against a value that has not reached the
// target (it has reached it meanwhile: the barrier stands, carry on), and whoever reads or increments the counter afterwards finds
// the bit.  -> false in every workgroup: each merges the pieces it had listed itself, alone (td_giant_pieces, below).  Once the
// first barrier stands every workgroup is resident until the kernel ends and the later barriers cannot starve.
constexpr unsigned long long GP_MEET_TIMEOUT = 5000000ull;  // 50 ms
constexpr uint32_t GP_BAR_DEAD = 0x80000000u;
__device__ __forceinline__ bool gp_grid_meet(const GpCoop& c) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
#ifndef TD_GP_BARRIER_RELAXED
        const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // (orders as in gp_grid_barrier)
#else
        const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        if (old & GP_BAR_DEAD) {
            ok = 0;
        } else {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                uint32_t v = __hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & GP_BAR_DEAD) { ok = 0; break; }
                if (v >= c.nblk) break;  // (the counter is zero at launch)
                if (wall_clock64() - t0 > GP_MEET_TIMEOUT) {
                    if (__hip_atomic_compare_exchange_strong(c.bar, &v, v | GP_BAR_DEAD, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                    continue;  // (somebody arrived meanwhile: look again)
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
#ifndef TD_GP_BARRIER_RELAXED
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        *c.s_flag = ok;
    }
    __syncthreads();
    return *c.s_flag != 0;
}
__device__ __forceinline__ uint32_t* gp_slot(const GpCoop& c, uint32_t blk, uint32_t k) {
    return c.slots + (((c.seq & 1u) * GP_MAX_BLOCKS + blk) * 2u + k);
}
__device__ __forceinline__ uint32_t gp_slot_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// minimum of x0 and of x1 over the whole grid (x0, x1: this workgroup's values, the same in all its threads)
__device__ __forceinline__ bool gp_grid_min2(GpCoop& c, uint32_t& x0, uint32_t& x1) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(gp_slot(c, blockIdx.x, 0), x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gp_slot(c, blockIdx.x, 1), x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!gp_grid_barrier(c)) return false;
    uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
    for (uint32_t k = threadIdx.x & 63u; k < c.nblk; k += 64u) {
        const uint32_t v0 = gp_slot_ld(gp_slot(c, k, 0)), v1 = gp_slot_ld(gp_slot(c, k, 1));
        m0 = v0 < m0 ? v0 : m0;
        m1 = v1 < m1 ? v1 : m1;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o0 = __shfl_xor(m0, d), o1 = __shfl_xor(m1, d);
        m0 = o0 < m0 ? o0 : m0;
        m1 = o1 < m1 ? o1 : m1;
    }
    x0 = m0; x1 = m1;
    ++c.seq;
    return true;
}
// x = this workgroup's value -> before = the maximum (IS_MAX) or the sum of the workgroups' in front of it (identity 0), total = of all
template <bool IS_MAX>
__device__ __forceinline__ bool gp_grid_prefix(GpCoop& c, uint32_t x, uint32_t& before, uint32_t& total) {
    if (threadIdx.x == 0) __hip_atomic_store(gp_slot(c, blockIdx.x, 0), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!gp_grid_barrier(c)) return false;
    uint32_t bf = 0, tt = 0;
    for (uint32_t k = threadIdx.x & 63u; k < c.nblk; k += 64u) {
        const uint32_t v = gp_slot_ld(gp_slot(c, k, 0));
        if (IS_MAX) { tt = v > tt ? v : tt; if (k < blockIdx.x) bf = v > bf ? v : bf; }
        else { tt += v; if (k < blockIdx.x) bf += v; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t ob = __shfl_xor(bf, d), ot = __shfl_xor(tt, d);
        if (IS_MAX) { bf = ob > bf ? ob : bf; tt = ot > tt ? ot : tt; }
        else { bf += ob; tt += ot; }
    }
    before = bf; total = tt;
    ++c.seq;
    return true;
}

struct GpShared {
    int* s_w;
    uint32_t* s_m;
};

// One piece: the `len` bytes at p, arrays at `base` in the pool (4 x len words; COOP: 5 x len).  COOP = false: this workgroup alone
// (what rounds 2-4 had); COOP = true: every workgroup of the launch, all in the same control flow (m, g, B are grid-wide values).
// -> false: a grid barrier gave up (c.stop): leave the kernel.
template <bool COOP>
__device__ bool gp_piece(const EncodeArgs& a, const Tables& T, const uint32_t j, uint32_t* base, const GpShared sh, GpCoop& co) {
    int* const s_w = sh.s_w;
    uint32_t* const s_m = sh.s_m;
    const int tid = threadIdx.x;
    const uint32_t len = a.long_list[j].len;
    const int64_t gs = a.long_list[j].gs;
    const uint8_t* p = a.text + gs;
    const uint32_t blk = COOP ? blockIdx.x : 0u, nblk = COOP ? co.nblk : 1u;
    uint32_t* id_cur = base;               // two generations of (ids, rank words): a round reads one, writes the other
    uint32_t* rk_cur = id_cur + len;
    uint32_t* id_nxt = rk_cur + len;
    uint32_t* rk_nxt = id_nxt + len;
    uint32_t* const rk_own = rk_nxt + len;  // (COOP) the ranks of the pairs the merges create
    // (one workgroup: they go to rk_nxt, where the compaction takes them from, in place)
    // generation 0: one part per byte, rank of every byte pair
    uint32_t m = len, lmin = GP_INF;
    for (uint32_t i = blk * GP_THREADS + tid; i < len; i += nblk * GP_THREADS) {
        const uint32_t b = p[i];
        const int32_t r0 = (i + 1 < len) ? T.byte_pair[(b << 8) | p[i + 1]] : NO_RANK;
        const uint32_t r = r0 == NO_RANK ? GP_INF : (uint32_t)r0;
        gp_st<COOP>(id_cur + i, (uint32_t)T.byte_id[b]);
        gp_st<COOP>(rk_cur + i, r);
        lmin = r < lmin ? r : lmin;
    }
    if (!COOP) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t g = gp_block_min(lmin, s_m);
    if constexpr (COOP) {
        uint32_t dummy = 0;
        if (!gp_grid_min2(co, g, dummy)) return false;
    }
    while (g != GP_INF) {
        uint32_t* const rk_new = COOP ? rk_own : rk_nxt;
        // this workgroup's stretch of the generation
        uint32_t lo = 0, hi = m;
        if constexpr (COOP) {
            const uint32_t per = ((m + nblk - 1u) / nblk + GP_STEP - 1u) / GP_STEP * GP_STEP;
            const unsigned long long l64 = (unsigned long long)blk * per;
            lo = l64 < m ? (uint32_t)l64 : m;
            hi = m - lo < per ? m : lo + per;
        }
        // ---- sweep 1: the round's candidates.  rank word i gets GP_BASE when pair i ranks below its left neighbour pair's
        //      run start ... (see above): r_i <= r_{i+1}, the run of equal ranks it lies in starts at rs with r_rs < r_{rs-1},
        //      and i - rs is even ----
        int carry = -1;  // last position so far where the rank differs from the one in front of it (run start)
        if constexpr (COOP) {  // ... in front of the stretch: the last run start of the workgroups in front of this one
            uint32_t last1 = 0;  // (position + 1; 0: none)
            for (uint32_t q = lo + (uint32_t)tid; q < hi; q += GP_THREADS) {
                const uint32_t r = gp_ld<COOP>(rk_cur + q) & GP_INF;
                if (q == 0u || r != (gp_ld<COOP>(rk_cur + q - 1) & GP_INF)) last1 = q + 1u;  // (q grows: the last one stays)
            }
            last1 = ~gp_block_min(~last1, s_m);
            uint32_t before, total;
            if (!gp_grid_prefix<true>(co, last1, before, total)) return false;
            carry = (int)before - 1;
        }
        for (uint32_t i0 = lo; i0 < hi; i0 += GP_STEP) {
            const uint32_t i = i0 + (uint32_t)tid * GP_E;
            uint32_t r[GP_E + 2];  // r[0] = rank at i - 1 ... r[GP_E + 1] = rank at i + GP_E
#pragma unroll
            for (int e = 0; e < GP_E + 2; ++e) {
                const uint32_t q = i + (uint32_t)e - 1u;
                r[e] = (q < m) ? (gp_ld<COOP>(rk_cur + q) & GP_INF) : GP_INF;  // (q = i - 1 wraps for i = 0: >= m)
            }
            int last = -1;  // my last run start
#pragma unroll
            for (int e = 0; e < GP_E; ++e)
                if (i + (uint32_t)e < m && (i + (uint32_t)e == 0u || r[e + 1] != r[e])) last = (int)(i + (uint32_t)e);
            int tot;
            const int incl = gp_block_scan<true>(last, s_w, tot);  // (max-scan: identity 0x80000000 < -1)
            int before = __shfl_up(incl, 1);                       // run start in front of my first part: the thread before me ...
            if ((tid & 63) == 0) before = -1;
            {   // ... (across wavefronts: the scan's own prefix) — recomputed from the totals
                int pre = carry;
                for (int w = 0; w < (tid >> 6); ++w) pre = s_w[w] > pre ? s_w[w] : pre;
                before = before > pre ? before : pre;
            }
            int rs = before;
#pragma unroll
            for (int e = 0; e < GP_E; ++e) {
                const uint32_t q = i + (uint32_t)e;
                if (q < m) {
                    if (q == 0u || r[e + 1] != r[e]) rs = (int)q;
                    const uint32_t rq = r[e + 1];
                    bool cand = rq != GP_INF && rq <= r[e + 2] && !(((int)q - rs) & 1);
                    if (cand) {  // the run's start ranks strictly below the pair in front of it
                        const uint32_t rl = rs > 0 ? (gp_ld<COOP>(rk_cur + rs - 1) & GP_INF) : GP_INF;
                        cand = rq < rl;
                    }
                    gp_st<COOP>(rk_cur + q, rq | (cand ? GP_BASE : 0u));  // (others read the low 24 bits of this word meanwhile: they do not change)
                }
            }
            carry = tot > carry ? tot : carry;
            __syncthreads();  // (s_w is read above after the scan's last barrier)
        }
        if constexpr (COOP) {
            if (!gp_grid_barrier(co)) return false;
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
        }
        // ---- sweeps 2: the bound.  sel_i = GP_BASE and rank < B; lowered until the two conditions hold ----
        uint32_t B = GP_INF;       // merge the candidates that rank below B
        uint32_t first_sel = 0;    // B == 0: the plain sequential step — only this pair merges
        for (;;) {
            auto sel_of = [&](uint32_t q, uint32_t w) { return B ? ((w & GP_BASE) && (w & GP_INF) < B) : (q == first_sel); };
            uint32_t vmin = GP_INF, smax = 0;  // lowest rank that has to stay above the merged ones; highest merged rank
            for (uint32_t q = lo + (uint32_t)tid; q < hi; q += GP_THREADS) {
                const uint32_t w0 = gp_ld<COOP>(rk_cur + q), r0 = w0 & GP_INF;
                const uint32_t wm1 = q >= 1 ? gp_ld<COOP>(rk_cur + q - 1) : GP_INF, wp1 = q + 1 < m ? gp_ld<COOP>(rk_cur + q + 1) : GP_INF;
                const bool s0 = sel_of(q, w0), sm1 = q >= 1 && sel_of(q - 1, wm1), sp1 = q + 1 < m && sel_of(q + 1, wp1);
                if (!s0) {
                    if (!sm1 && !sp1 && r0 < vmin) vmin = r0;  // a pair that is left over and not overlapped by a merging one
                    continue;
                }
                smax = r0 > smax ? r0 : smax;
                const uint32_t wp2 = q + 2 < m ? gp_ld<COOP>(rk_cur + q + 2) : GP_INF, wm2 = q >= 2 ? gp_ld<COOP>(rk_cur + q - 2) : GP_INF;
                const bool sp2 = q + 2 < m && sel_of(q + 2, wp2), sm2 = q >= 2 && sel_of(q - 2, wm2);
                if (q + 2 < m) {  // the pair my merged part makes with what follows it
                    const uint32_t idn = gp_ld<COOP>(id_cur + q + 2);
                    const uint32_t nr = gp_rank(T, r0, sp2 ? (wp2 & GP_INF) : idn);
                    gp_st<COOP>(rk_new + q, nr);
                    vmin = nr < vmin ? nr : vmin;
                    if (sp2) {  // two merges one part apart: the pair the sequential order sees in between
                        const uint32_t tr = r0 <= (wp2 & GP_INF) ? gp_rank(T, r0, idn) : gp_rank(T, gp_ld<COOP>(id_cur + q + 1), wp2 & GP_INF);
                        vmin = tr < vmin ? tr : vmin;
                    }
                } else {
                    gp_st<COOP>(rk_new + q, GP_INF);
                }
                if (q >= 1 && !sm2) {  // ... and with the (unchanged) part in front of it
                    const uint32_t nl = gp_rank(T, gp_ld<COOP>(id_cur + q - 1), r0);
                    gp_st<COOP>(rk_new + q - 1, nl);
                    vmin = nl < vmin ? nl : vmin;
                }
            }
            vmin = gp_block_min(vmin, s_m);
            smax = ~gp_block_min(~smax, s_m);
            if constexpr (COOP) {
                uint32_t nsmax = ~smax;
                if (!gp_grid_min2(co, vmin, nsmax)) return false;
                smax = ~nsmax;
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
            }
            if (B == 0u || smax < vmin) break;       // (the sequential step needs no check)
            B = vmin;
            if (B <= g) {  // nothing but the lowest rank is left below the bound: its leftmost pair alone
                B = 0u;
                uint32_t f = 0xFFFFFFFFu;
                for (uint32_t q = lo + (uint32_t)tid; q < hi; q += GP_THREADS)
                    if ((gp_ld<COOP>(rk_cur + q) & GP_INF) == g && q < f) f = q;
                first_sel = gp_block_min(f, s_m);
                if constexpr (COOP) {
                    uint32_t dummy = 0;
                    if (!gp_grid_min2(co, first_sel, dummy)) return false;
                }
            }
        }
        // ---- sweep 3: next generation — surviving parts compacted, the new ranks taken from rk_new ----
        {
            auto sel_of = [&](uint32_t q, uint32_t w) { return B ? ((w & GP_BASE) && (w & GP_INF) < B) : (q == first_sel); };
            uint32_t out_base = 0, m_next = 0;
            if constexpr (COOP) {  // where this stretch's survivors go: the survivors of the stretches in front of it
                uint32_t cnt = 0;
                for (uint32_t q = lo + (uint32_t)tid; q < hi; q += GP_THREADS)
                    cnt += (q >= 1 && sel_of(q - 1, gp_ld<COOP>(rk_cur + q - 1))) ? 0u : 1u;
                int tot;
                (void)gp_block_scan<false>((int)cnt, s_w, tot);
                if (!gp_grid_prefix<false>(co, (uint32_t)tot, out_base, m_next)) return false;
            }
            lmin = GP_INF;
            for (uint32_t i0 = lo; i0 < hi; i0 += GP_STEP) {
                const uint32_t i = i0 + (uint32_t)tid * GP_E;
                uint32_t A[GP_E], nr[GP_E];
                bool surv[GP_E];
                uint32_t cnt = 0;
                uint32_t wprev = i >= 1 && i - 1 < m ? gp_ld<COOP>(rk_cur + i - 1) : GP_INF;
                bool sprev = i >= 1 && i - 1 < m && sel_of(i - 1, wprev);
#pragma unroll
                for (int e = 0; e < GP_E; ++e) {
                    const uint32_t q = i + (uint32_t)e;
                    surv[e] = false; A[e] = 0; nr[e] = GP_INF;
                    if (q < m) {
                        const uint32_t w0 = gp_ld<COOP>(rk_cur + q);
                        const bool s0 = sel_of(q, w0);
                        surv[e] = !sprev;
                        if (surv[e]) {
                            const bool sn = q + 1 < m && sel_of(q + 1, gp_ld<COOP>(rk_cur + q + 1));
                            A[e] = s0 ? (w0 & GP_INF) : gp_ld<COOP>(id_cur + q);
                            nr[e] = (s0 || sn) ? gp_ld<COOP>(rk_new + q) : (w0 & GP_INF);  // (a merged part's pairs: looked up by the sweeps above)
                            if (s0 && q + 2 >= m) nr[e] = GP_INF;
                            if (!s0 && q + 1 >= m) nr[e] = GP_INF;
                            ++cnt;
                        }
                        sprev = s0;
                    }
                }
                int tot;
                const int incl = gp_block_scan<false>((int)cnt, s_w, tot);
                uint32_t o = out_base + (uint32_t)incl - cnt;
#pragma unroll
                for (int e = 0; e < GP_E; ++e)
                    if (surv[e]) {
                        gp_st<COOP>(id_nxt + o, A[e]);
                        gp_st<COOP>(rk_nxt + o, nr[e]);
                        lmin = nr[e] < lmin ? nr[e] : lmin;
                        ++o;
                    }
                out_base += (uint32_t)tot;
            }
            if (!COOP) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            g = gp_block_min(lmin, s_m);
            if constexpr (COOP) {
                uint32_t dummy = 0;
                if (!gp_grid_min2(co, g, dummy)) return false;
                m = m_next;
            } else {
                m = out_base;
            }
            uint32_t* t0 = id_cur; id_cur = id_nxt; id_nxt = t0;
            uint32_t* t1 = rk_cur; rk_cur = rk_nxt; rk_nxt = t1;
            if (!COOP) __syncthreads();
        }
    }
    // the ids of the piece: generation `cur`
    for (uint32_t i = blk * GP_THREADS + tid; i < m; i += nblk * GP_THREADS) {
        const uint32_t v = gp_ld<COOP>(id_cur + i);
        if ((int32_t)v >= T.pseudo_base) raise(a, TD_E_UNKNOWN_BYTE, gs);
    }
    if (tid == 0 && blk == 0u) {
        a.long_list[j].ntok = m;
        a.long_list[j].pool_off = (unsigned long long)(id_cur - a.pool);
        if (m > 1) atomicAdd(&a.tile_extra[gs / K_TILE], m - 1);
    }
    if (!COOP) __syncthreads();
    return true;
}

__device__ __forceinline__ void giant_pieces_body(const EncodeArgs& a) {  // (its workgroups are the launch's: blockIdx.x / gridDim.x)
    __shared__ int s_w[GP_THREADS / 64];
    __shared__ uint32_t s_m[GP_THREADS / 64];
    __shared__ unsigned long long s_off;
    __shared__ int s_flag;
    const Tables T = uniform_tables(a.Tp);
    const int tid = threadIdx.x;
    static_assert(K_GIANT_MIN == LP_MEDIUM, "what td_long_pieces leaves");
    // (without a piece above LP_MEDIUM bytes there is nothing to look for: walking the list of 132 000 long pieces of 256 MiB
    // of mixed-script text, 128 workgroups with a dependent load per entry, was 0.27 ms for nothing)
    if (*a.giant_count == 0u) return;
    const uint32_t nlong = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    const GpShared sh{s_w, s_m};
    GpCoop co{a.gp_ctl, a.gp_ctl + 2, a.gp_scratch + GP_LIST_CAP, gridDim.x, 0u, &s_flag};
    const bool grid_ok = gridDim.x > 1u && gridDim.x <= (uint32_t)GP_MAX_BLOCKS && a.gp_ctl[3] != 0u;  // (a.gp_ctl[3]: pieces above gp_coop_min, counted where they were listed)
    const uint32_t coop_min = grid_ok ? (a.gp_coop_min > (uint32_t)LP_MEDIUM ? a.gp_coop_min : (uint32_t)LP_MEDIUM) : 0xFFFFFFFFu;
    // ---- the pieces up to coop_min bytes: a workgroup each; the longer ones are listed for all workgroups together ----
    for (uint32_t j = blockIdx.x; j < nlong; j += gridDim.x) {
        const uint32_t len = a.long_list[j].len;
        if (len <= (uint32_t)LP_MEDIUM) continue;  // (uniform: td_long_pieces')
        if (len > coop_min) {
            if (tid == 0) s_off = atomicAdd(a.gp_ctl + 1, 1u);
            __syncthreads();
            const uint32_t k = (uint32_t)s_off;
            __syncthreads();
            if (k < GP_LIST_CAP) {
                if (tid == 0) __hip_atomic_store(a.gp_scratch + k, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }  // (the list is full: this workgroup alone)
        }
        if (tid == 0) s_off = atomicAdd(a.pool_used, 4ull * len);
        __syncthreads();
        const unsigned long long off = s_off;
        __syncthreads();
        if (off + 4ull * len > a.pool_cap) {
            if (tid == 0) raise(a, TD_E_SCRATCH, a.long_list[j].gs);
            continue;
        }
        (void)gp_piece<false>(a, T, j, a.pool + off, sh, co);
    }
    if (!grid_ok) return;
    // ---- the listed pieces: all workgroups on one after the other (every workgroup passes the same barriers) ----
    if (!gp_grid_meet(co)) {  // this launch's workgroups are not resident together: every one merges what it had listed, alone
        for (uint32_t j = blockIdx.x; j < nlong; j += gridDim.x) {
            const uint32_t len = a.long_list[j].len;
            if (len <= coop_min) continue;
            if (__hip_atomic_load(&a.long_list[j].ntok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;  // (the list was full: merged above)
            if (tid == 0) s_off = atomicAdd(a.pool_used, 4ull * len);
            __syncthreads();
            const unsigned long long off1 = s_off;
            __syncthreads();
            if (off1 + 4ull * len > a.pool_cap) {
                if (tid == 0) raise(a, TD_E_SCRATCH, a.long_list[j].gs);
                continue;
            }
            (void)gp_piece<false>(a, T, j, a.pool + off1, sh, co);
        }
        return;
    }
    uint32_t ncoop = gp_slot_ld(a.gp_ctl + 1);
    ncoop = ncoop < GP_LIST_CAP ? ncoop : GP_LIST_CAP;
    // (nothing else takes from the pool from here on: every workgroup computes the same offsets)
    const unsigned long long used0 = __hip_atomic_load(a.pool_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long off = used0;
    for (uint32_t k = 0; k < ncoop; ++k) {
        const uint32_t j = gp_slot_ld(a.gp_scratch + k);
        const uint32_t len = a.long_list[j].len;
        if (off + 5ull * len > a.pool_cap) {
            if (tid == 0 && blockIdx.x == 0u) raise(a, TD_E_SCRATCH, a.long_list[j].gs);
            continue;
        }
        if (!gp_piece<true>(a, T, j, a.pool + off, sh, co)) {
            if (tid == 0) raise(a, TD_E_HIP, a.long_list[j].gs);
            return;
        }
        off += 5ull * len;
    }
    // (every workgroup has read pool_used before the first listed piece's first barrier, which workgroup 0 is behind by now)
    if (ncoop && tid == 0 && blockIdx.x == 0u) atomicAdd(a.pool_used, off - used0);
}
__global__ __launch_bounds__(GP_THREADS) void td_giant_pieces(const EncodeArgs a) { giant_pieces_body(a); }

// ------------------------------------------------------------------ td_scan_tiles -----------
// Device-wide exclusive scan of the per-tile token counts (n_tiles = N/4096: 65 536 entries for 256 MiB), two levels
// in one launch: every 1024-thread workgroup scans one chunk of 4096 tiles (16 B per lane, wavefront shuffle scans + one
// LDS hop) and publishes the chunk total; the workgroup that finishes last scans the chunk totals (chunk_pref).  A
// tile's token base is tile_base[tile] + chunk_pref[tile / 4096].
constexpr int K_SCAN_CHUNK = 4096;
// td_pack_plain's tiles (the pair td_pack_plain / td_pack_rest, a.pack_split): no long piece, at most a few merged ones (the tiles
// the tile loops flag for their many missed pieces are pack_body's: measured, its row groups are faster on them), not placed by
// the fused loop, at most 1024 slots, all of its ids inside the output
__device__ __forceinline__ bool pk_rest_tile(uint32_t tc) {  // (what td_scan_tiles notes in rest_mask: the part that needs no base)
    return (tc & (TILE_HAS_LONG | TILE_HAS_MISS | TILE_DIRECT)) != 0u || (tc & TILE_COUNT_MASK) > 1024u;
}
__device__ __forceinline__ bool pk_simple(uint32_t tc, int64_t base, uint32_t extra, int64_t out_cap) {
    return !pk_rest_tile(tc) && base + (int64_t)(int32_t)((tc & TILE_COUNT_MASK) + extra) <= out_cap;
}
// td_pack_dense's tiles (round 6, the dense launch sequence): every tile that is not td_pack_plain's — many merged pieces, long pieces, more than
// 1024 slots — unless the fused loop placed it or its ids do not fit the output: those are what is left to td_pack_rest.
__device__ __forceinline__ bool pk_dense(uint32_t tc, int64_t base, uint32_t extra, int64_t out_cap) {
    return ((tc & (TILE_HAS_LONG | TILE_HAS_MISS)) != 0u || (tc & TILE_COUNT_MASK) > 1024u) && !(tc & TILE_DIRECT) &&
           base + (int64_t)(int32_t)((tc & TILE_COUNT_MASK) + extra) <= out_cap;
}
__device__ __forceinline__ bool pk_mask_bit(uint32_t tc, int dense) {  // what td_scan_tiles notes for td_pack_rest
    return dense ? (tc & TILE_DIRECT) != 0u : pk_rest_tile(tc);
}
// one chunk of 4096 tiles by the whole (1024-thread) workgroup: tile_base inside the chunk, rest_mask, the chunk's total -> chunk_pref[chunk]
__device__ __forceinline__ void scan_chunk(const EncodeArgs& a, const int chunk, unsigned long long* const s_wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const int c0 = chunk * K_SCAN_CHUNK;
        const int e0 = c0 + tid * 4;
        uint32_t v[4] = {0, 0, 0, 0};
        uint32_t nib = 0;  // bit k: tile e0 + k is td_pack_rest's whatever its base (pk_rest_tile)
        if (e0 + 4 <= a.n_tiles) {
            const uint4 cnt = *reinterpret_cast<const uint4*>(a.tile_count + e0);
            const uint4 ext = *reinterpret_cast<const uint4*>(a.tile_extra + e0);
            v[0] = (cnt.x & TILE_COUNT_MASK) + ext.x; v[1] = (cnt.y & TILE_COUNT_MASK) + ext.y;
            v[2] = (cnt.z & TILE_COUNT_MASK) + ext.z; v[3] = (cnt.w & TILE_COUNT_MASK) + ext.w;
            nib = (pk_mask_bit(cnt.x, a.pack_dense) ? 1u : 0u) | (pk_mask_bit(cnt.y, a.pack_dense) ? 2u : 0u) | (pk_mask_bit(cnt.z, a.pack_dense) ? 4u : 0u) | (pk_mask_bit(cnt.w, a.pack_dense) ? 8u : 0u);
        } else {
            for (int k = 0; k < 4; ++k)
                if (e0 + k < a.n_tiles) {
                    const uint32_t tc = a.tile_count[e0 + k];
                    v[k] = (tc & TILE_COUNT_MASK) + a.tile_extra[e0 + k];
                    nib |= (pk_mask_bit(tc, a.pack_dense) ? 1u : 0u) << k;
                }
        }
        {   // ... sixteen tiles a word: td_pack_rest walks these instead of every tile's count word
            const uint32_t m16 = nib | ((uint32_t)__shfl_down((int)nib, 1) << 4) | ((uint32_t)__shfl_down((int)nib, 2) << 8) | ((uint32_t)__shfl_down((int)nib, 3) << 12);
            if ((lane & 3) == 0 && e0 < a.n_tiles) a.rest_mask[e0 >> 4] = (uint16_t)m16;
        }
        const unsigned long long mine = (unsigned long long)v[0] + v[1] + v[2] + v[3];
        unsigned long long x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(x, d);
            if (lane >= d) x += t;
        }
        if (lane == 63) s_wsum[wv] = x;
        __syncthreads();
        unsigned long long woff = 0;
        for (int w = 0; w < wv; ++w) woff += s_wsum[w];
        unsigned long long run = woff + x - mine;  // exclusive prefix of my first element inside the chunk
        for (int k = 0; k < 4; ++k) {
            if (e0 + k < a.n_tiles) a.tile_base[e0 + k] = (int64_t)run;
            run += v[k];
        }
        if (tid == 1023) a.chunk_pref[chunk] = (int64_t)run;  // chunk total for now; scan_totals turns it into a prefix
        __syncthreads();  // (s_wsum is free again)
    }
}
// exclusive scan of the chunk totals (at most a few hundred), one wavefront
__device__ __forceinline__ void scan_totals(const EncodeArgs& a, const int nchunks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        unsigned long long carry = 0;
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
            const int c = c0 + lane;
            const unsigned long long tot = (c < nchunks) ? (unsigned long long)a.chunk_pref[c] : 0ull;
            unsigned long long x = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long t = __shfl_up(x, d);
                if (lane >= d) x += t;
            }
            if (c < nchunks) a.chunk_pref[c] = (int64_t)(carry + x - tot);
            carry += __shfl(x, 63);
        }
        if (lane == 0) {
            const int64_t total = (int64_t)carry;
            a.chunk_pref[nchunks] = total;
            a.tile_base[a.n_tiles] = total;
            if (total > a.out_cap) raise(a, TD_E_CAPACITY, total);
        }
    }
}
// every workgroup its chunks; the one that finishes the last chunk scans the totals
__device__ __forceinline__ void scan_tiles_parallel(const EncodeArgs& a, unsigned long long* const s_wsum, uint32_t* const s_last) {
    const int nchunks = (a.n_tiles + K_SCAN_CHUNK - 1) / K_SCAN_CHUNK;
    uint32_t last = 0;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        scan_chunk(a, c, s_wsum);
        if (threadIdx.x == 1023) {
            __threadfence();
            *s_last = (atomicAdd(a.scan_done, 1u) == (uint32_t)nchunks - 1u) ? 1u : 0u;
        }
        __syncthreads();
        last |= *s_last;
        __syncthreads();
    }
    if (!last) return;
    __threadfence();
    scan_totals(a, nchunks);
}
__global__ __launch_bounds__(1024) void td_scan_tiles(const EncodeArgs a) {
    __shared__ unsigned long long s_wsum[16];
    __shared__ uint32_t s_last;
    scan_tiles_parallel(a, s_wsum, &s_last);
}

// ------------------------------------------------------------------ several phases in one launch (round 6) ----
// A launch costs a step ~5 us whatever it finds to do (the queue's barrier between two dependent kernels), and on plain text eight of a
// step's fifteen kernels found nothing: at 128 MiB per call — one rank's share of the 1024 MiB corpus on eight GPUs — that was 40 of
// 415 us.  Kernels whose work is usually absent are now PHASES of one launch: every workgroup reads the counters the kernels in front
// left (final: a kernel boundary lies in between), phases without work are skipped by all workgroups alike without any
// synchronisation, and only where a phase with work feeds another one do the workgroups meet at a grid barrier.  The first such
// barrier asks whether the launch's workgroups are resident together (ph_meet: as gp_grid_meet of td_giant_pieces — several handles
// or processes with such launches at once could each hold part of the CUs and wait for the rest); if they are not, workgroup 0 walks
// the phases alone, the others leave: slower, never wrong, never stuck.  Data crosses a phase boundary through agent-scope release /
// acquire fences (each XCD has an L2 of its own).
//   td_far_probe   (the dense sequence)   td_split_far_pieces | td_split_far_tiles | td_probe_tiles over the deferred tiles
//   td_tail        (the sparse sequence)  the same three | td_collect_misses + td_long_pieces | td_merge_pieces | td_copy_dups
//   td_giant_scan  (the sparse sequence)  td_giant_pieces | td_scan_tiles
struct PhaseSync {
    uint32_t* bar;   // arrivals (zero at launch)
    uint32_t nb;     // workgroups that meet
    int* s_flag;
    bool solo;       // this workgroup walks the phases alone (the others have left)
};
constexpr unsigned long long PH_MEET_TIMEOUT = 5000000ull;    // wall_clock64 ticks (100 MHz): 50 ms
constexpr unsigned long long PH_BAR_TIMEOUT = 400000000ull;   // 4 s: a fault
constexpr uint32_t PH_DEAD = 0x80000000u;
__device__ __forceinline__ bool ph_meet(const PhaseSync& c) {
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old & PH_DEAD) {
            ok = 0;
        } else {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                uint32_t v = __hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & PH_DEAD) { ok = 0; break; }
                if (v >= c.nb) break;
                if (wall_clock64() - t0 > PH_MEET_TIMEOUT) {
                    if (__hip_atomic_compare_exchange_strong(c.bar, &v, v | PH_DEAD, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                    continue;  // (somebody arrived meanwhile: look again)
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        *c.s_flag = ok;
    }
    __syncthreads();
    return *c.s_flag != 0;
}
// end of a phase: what this workgroup wrote is visible to the others behind it (and the other way round).  false: a fault (TD_E_HIP raised).
__device__ __forceinline__ bool ph_sync(const EncodeArgs& a, const PhaseSync& c) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (!c.solo) {
        if (threadIdx.x == 0) {
            const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = ((old & ~PH_DEAD) / c.nb + 1u) * c.nb;
            const unsigned long long t0 = wall_clock64();
            int ok = (old & PH_DEAD) ? 0 : 1;
            while (ok) {
                const uint32_t v = __hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & PH_DEAD) { ok = 0; break; }
                if (v >= target) break;
                if (wall_clock64() - t0 > PH_BAR_TIMEOUT) {
                    __hip_atomic_fetch_or(c.bar, PH_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    raise(a, TD_E_HIP, 0);
                    ok = 0;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            *c.s_flag = ok;
        }
        __syncthreads();
        if (!*c.s_flag) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
// ... between phases whose only traffic is agent-scope atomics and agent-scope loads / stores (the far phases: START bits by atomicOr,
// tile_carry, the error word): no cache write-back or invalidation, just "my accesses have arrived" and the counter (the fences above
// cost the code file set 40 us per step: two barriers of 512 workgroups each writing back and invalidating its XCD's L2)
__device__ __forceinline__ bool ph_sync_light(const EncodeArgs& a, const PhaseSync& c) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (!c.solo) {
        if (threadIdx.x == 0) {
            const uint32_t old = __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = ((old & ~PH_DEAD) / c.nb + 1u) * c.nb;
            const unsigned long long t0 = wall_clock64();
            int ok = (old & PH_DEAD) ? 0 : 1;
            while (ok) {
                const uint32_t v = __hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & PH_DEAD) { ok = 0; break; }
                if (v >= target) break;
                if (wall_clock64() - t0 > PH_BAR_TIMEOUT) {
                    __hip_atomic_fetch_or(c.bar, PH_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    raise(a, TD_E_HIP, 0);
                    ok = 0;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            *c.s_flag = ok;
        }
        __syncthreads();
        if (!*c.s_flag) return false;
    }
    return true;
}
__device__ __forceinline__ uint32_t ph_count(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// what td_split_tiles left for the window it could not see through: far pieces -> chains of flagged tiles -> the deferred token tiles.
// -> false: this workgroup is done (it is not the one that walks alone, or a barrier failed)
__device__ __forceinline__ bool far_probe_phases(const EncodeArgs& a, PhaseSync& c, uint32_t& bid, uint32_t& nb, const uint32_t nslow,
                                                 const uint32_t nfar, const uint32_t ndef, bool& met) {
    if (!(nslow | nfar | ndef)) return true;
    if (!met) {
        met = true;
        if (!ph_meet(c)) {
            if (blockIdx.x != 0u) return false;
            c.solo = true; bid = 0u; nb = 1u;
        }
    }
    if (nslow) { far_pieces_body(a, bid, nb); if (!(a.far_light ? ph_sync_light(a, c) : ph_sync(a, c))) return false; }
    if (nfar) { far_tiles_body(a, bid, nb); if (!(a.far_light ? ph_sync_light(a, c) : ph_sync(a, c))) return false; }
    if (ndef) {
        EncodeArgs ad = a;
        ad.probe_deferred = 1;
        probe_tiles_body(ad, bid, nb);
    }
    return true;
}
__global__ __launch_bounds__(K_THREADS) void td_far_probe(const EncodeArgs a) {
    __shared__ int s_flag;
    const uint32_t nslow = *a.slow_count, nfar = *a.far_count, ndef = a.fused ? *a.deferred_count : 0u;
    if (!(nslow | nfar | ndef)) return;
    PhaseSync c{a.ph_bar, gridDim.x, &s_flag, false};
    uint32_t bid = blockIdx.x, nb = gridDim.x;
    bool met = false;
    (void)far_probe_phases(a, c, bid, nb, nslow, nfar, ndef, met);
}

// The sparse sequence's one kernel between the tile loop and the scan (launch_encode takes it when the handle's last counters say the
// text has next to nothing but listed pieces: plain prose).  Correct for ANY text — a call that brings many flagged tiles or long
// pieces after all is only slower here than in the kernels of their own, which run at two to three times the occupancy.
constexpr int TAIL_LDS_WORDS = LONG_LDS_WORDS > MERGE_LDS_WORDS ? (LONG_LDS_WORDS > COLLECT_LDS_WORDS ? LONG_LDS_WORDS : COLLECT_LDS_WORDS)
                                                                : (MERGE_LDS_WORDS > COLLECT_LDS_WORDS ? MERGE_LDS_WORDS : COLLECT_LDS_WORDS);
static_assert(MG_THREADS == K_THREADS, "td_tail runs td_merge_pieces' body with its own workgroup size");
__global__ __launch_bounds__(K_THREADS, 2) void td_tail(const EncodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_arena[TAIL_LDS_WORDS];  // collect | merge | long, one after the other
    __shared__ int s_flag;
    const uint32_t nslow = *a.slow_count, nfar = *a.far_count, ndef = *a.deferred_count;
    uint32_t nflag = *a.flagged_count, nlong = *a.long_count;
    uint32_t bid = blockIdx.x, nb = gridDim.x;
    if (!(nslow | nfar | ndef | nflag)) {  // the usual case: the lists the tile loop filled, a long piece here and there — nothing waits for anything
        merge_pieces_body(a, bid, nb, s_arena);
        if (nlong) {
            __syncthreads();
            long_pieces_body(a, bid, nb, s_arena);
        }
        return;
    }
    PhaseSync c{a.ph_bar, gridDim.x, &s_flag, false};
    bool met = false;
    if (!far_probe_phases(a, c, bid, nb, nslow, nfar, ndef, met)) return;
    if (nslow | nfar | ndef) {  // (the deferred tiles' lookups may have listed pieces, flagged tiles, found long pieces)
        if (!ph_sync(a, c)) return;
        nflag = ph_count(a.flagged_count);
        nlong = ph_count(a.long_count);
    }
    // (the long pieces wait for nobody and nobody in here waits for them: every workgroup does its share BEFORE it asks whether the
    // launch meets — a workgroup that learns it does not and leaves has done that share, whenever it started)
    if (nlong) { long_pieces_body(a, bid, nb, s_arena); __syncthreads(); }
    if (nflag) {
        if (!met) {
            met = true;
            if (!ph_meet(c)) {
                if (blockIdx.x != 0u) return;
                c.solo = true; bid = 0u; nb = 1u;
            }
        }
        collect_misses_body(a, bid, nb, s_arena);
        if (!ph_sync(a, c)) return;
    }
    merge_pieces_body(a, bid, nb, s_arena);
    if (nflag && a.dedupe) {
        if (!ph_sync(a, c)) return;
        copy_dups_body(a, bid, nb);
    }
}

// td_giant_pieces and td_scan_tiles in one launch (the sparse sequence): without a piece above 1 KiB — the usual case — every workgroup
// scans its chunks at once; with one, the workgroup that leaves the giant pieces last scans all chunks alone (no workgroup waits for
// another one here: correct whatever is resident; a call with giant pieces pays milliseconds for them anyway).
__global__ __launch_bounds__(GP_THREADS) void td_giant_scan(const EncodeArgs a) {
    static_assert(GP_THREADS == 1024, "td_scan_tiles' workgroup");
    __shared__ unsigned long long s_wsum[16];
    __shared__ uint32_t s_last;
    if (*a.giant_count == 0u) {
        scan_tiles_parallel(a, s_wsum, &s_last);
        return;
    }
    giant_pieces_body(a);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(a.gs_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int nchunks = (a.n_tiles + K_SCAN_CHUNK - 1) / K_SCAN_CHUNK;
    for (int ch = 0; ch < nchunks; ++ch) scan_chunk(a, ch, s_wsum);
    __threadfence();
    __syncthreads();
    scan_totals(a, nchunks);
}

// ------------------------------------------------------------------ td_pack_tokens ----------
// per-tile slots -> densely packed ids + per-document token offsets.  One WAVEFRONT per tile (a tile's ids are only a few
// KB: many small independent copies in flight beat few large ones), no workgroup barrier, no LDS.
//   plain tiles (every slot is an id): 16-byte copies;
//   tiles with markers: a sweep over rows of 64 slots with a running id count: a slot's size is 1, the ids of a merged
//   piece (TOK_MISS | position | ids, from a.merge_out) or of a long piece (TOK_LONGREF, from the pool; copied by the whole
//   wavefront); the documents that start in the tile pick their offset out of the row scan their slot falls in.
// (A workgroup-per-tile version of the marker path with the tile's offsets in LDS cost 14-17 us per tile: five barriers
// and six dependent global loads in a row; plain English has a marker in every third tile.)
// Where the ids of a merged marker are: at the piece's own bytes' slots of merge_out, or — a repeat (TOK_DUPREF) — at the bytes' slots of the
// piece it repeats, which the table of distinct pieces names (one more load; the seats of frequent pieces are hot in every cache).
__device__ __forceinline__ const uint32_t* marker_ids(const EncodeArgs& a, uint32_t tile, uint32_t v) {
    if (v & TOK_DUPREF) {
        const unsigned long long other = a.dd_table[(v >> 7) & 0x1FFFFFu];
        return a.merge_out + (size_t)((uint32_t)(other >> 32) & 0xFFFFFFu) * K_STAGE + (((uint32_t)other >> 7) & 0xFFFu);
    }
    return a.merge_out + (size_t)tile * K_STAGE + ((v >> 7) & 0xFFFu);
}

constexpr int PK_G = 8;       // rows of 64 slots per group of the marker path
constexpr int PK_ECAP = 256;  // merged pieces the expansion list holds
// PLAIN_ONLY: only the pipelined path of the plain tiles (every slot an id, at most 1024 of them); SKIP_PLAIN: everything else.
// Launched as a pair (a.pack_split): the first has none of the marker path's registers — 8 wavefronts per SIMD instead of 4 —
// the second finds next to nothing to do on plain text.  <false, false>: one kernel for all tiles (rounds 2-3).
template <bool PLAIN_ONLY, bool SKIP_PLAIN>
__device__ __forceinline__ void pack_body(const EncodeArgs& a, unsigned long long (*s_elist)[PK_ECAP]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef TD_PACK_TIMING
    unsigned long long t_meta = 0, t_rows = 0, t_scan = 0, t_plain = 0, t_list = 0, t_flush = 0, t_docs = 0, t_plainpath = 0, n_mt = 0, n_pt = 0, n_fl = 0;
    unsigned long long t_last = __builtin_readcyclecounter(), t_total0 = t_last;
#define PK_TICK(var) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_now = __builtin_readcyclecounter(); var += t_now - t_last; t_last = t_now; }
#else
#define PK_TICK(var)
#endif
    const int64_t total = a.tile_base[a.n_tiles];
    const int nwaves = gridDim.x * (K_THREADS / 64);
    auto base_of = [&](int tile) { return a.tile_base[tile] + a.chunk_pref[tile / K_SCAN_CHUNK]; };
    // Software pipeline over the wavefront's tiles.  Loads and stores share one in-order counter on this hardware: waiting
    // for a load also waits for every store issued BEFORE it, so "load tile t, store tile t, load tile t+1 ..." exposes a
    // store round trip per tile (the kernel ran at 2.6 TB/s).  Here the loads of tile t+1 are issued in front of the stores
    // of tile t, and what depends on the tile index only (count, base, first document: wave-uniform scalar loads) is
    // requested two tiles ahead.  The pipelined form covers plain tiles of up to 1024 ids (every slot an id: one pass of four
    // 16-byte loads per lane); the others take the general path below.  (Plain variables and macros, no structs handed to
    // lambdas: as a struct by reference the tile's data lived in scratch memory, every load followed by a wait.)
    typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at a dword-aligned address
    (void)base_of;
    const int tile_first = (int)uni32((uint32_t)(blockIdx.x * (K_THREADS / 64) + wv));
    uint32_t tcA = 0, dfA = 0, tcB = 0, dfB = 0, tcC = 0, dfC = 0;          // A = this tile, B = the next one, C = the one after
    int64_t baseA = 0, baseB = 0, baseC = 0;
#define PK_FETCH(t, X) { tc##X = a.tile_count[t]; base##X = a.tile_base[t] + a.chunk_pref[(t) / K_SCAN_CHUNK]; df##X = a.tile_first_doc[t]; }
#define PK_FAST(X) (!(tc##X & (TILE_HAS_LONG | TILE_HAS_MISS | TILE_MISS_LISTED | TILE_DIRECT)) && (tc##X & TILE_COUNT_MASK) <= 1024u && \
                    base##X + (int64_t)(tc##X & TILE_COUNT_MASK) <= a.out_cap)
    uint4 cx0 = make_uint4(0, 0, 0, 0), cx1 = cx0, cx2 = cx0, cx3 = cx0, nx0 = cx0, nx1 = cx0, nx2 = cx0, nx3 = cx0;
    uint32_t ch0 = 0, ctl = 0, cdsl = 0, nh0 = 0, ntl = 0, ndsl = 0;
    int64_t cdpos = 0, ndpos = 0;
    // the tile's ids as 16-byte pieces aligned to the DESTINATION (single ids up to its next 16-byte boundary and behind the
    // last piece), its first 64 documents
#define PK_GEOM(X) const uint32_t g_c = tc##X & TILE_COUNT_MASK;                                                     \
                   uint32_t g_head = (uint32_t)((16u - ((uint32_t)(uintptr_t)(a.out_tokens + base##X) & 15u)) & 15u) >> 2; \
                   if (g_head > g_c) g_head = g_c;                                                                   \
                   const uint32_t g_nv = (g_c - g_head) >> 2, g_done = g_head + 4 * g_nv;
#define PK_LOAD(t, X, P) { PK_GEOM(X)                                                                                 \
        const uint32_t* g_src = a.stage + (size_t)(t) * K_STAGE;                                                      \
        P##h0 = 0; P##tl = 0; P##dsl = 0; P##dpos = a.n;                                                              \
        if ((uint32_t)lane < g_head) P##h0 = g_src[lane];                                                             \
        if (g_done + (uint32_t)lane < g_c) P##tl = g_src[g_done + lane];                                              \
        if ((uint32_t)lane < g_nv) { const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(g_src + g_head + 4 * lane); P##x0 = make_uint4(v.x, v.y, v.z, v.w); } \
        if ((uint32_t)lane + 64 < g_nv) { const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(g_src + g_head + 4 * (lane + 64)); P##x1 = make_uint4(v.x, v.y, v.z, v.w); } \
        if ((uint32_t)lane + 128 < g_nv) { const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(g_src + g_head + 4 * (lane + 128)); P##x2 = make_uint4(v.x, v.y, v.z, v.w); } \
        if ((uint32_t)lane + 192 < g_nv) { const u32x4a4 v = *reinterpret_cast<const u32x4a4*>(g_src + g_head + 4 * (lane + 192)); P##x3 = make_uint4(v.x, v.y, v.z, v.w); } \
        const int64_t g_dm = (int64_t)df##X + lane;                                                                   \
        if (g_dm < a.n_docs) { P##dpos = a.doc_offsets[g_dm]; P##dsl = a.doc_slot[g_dm]; } }
    bool fast_cur = false;
    if constexpr (!SKIP_PLAIN) {
        if (tile_first < a.n_tiles) PK_FETCH(tile_first, A)
        if (tile_first + nwaves < a.n_tiles) PK_FETCH(tile_first + nwaves, B)
        fast_cur = tile_first < a.n_tiles && PK_FAST(A);
        if (fast_cur) PK_LOAD(tile_first, A, c)
    }
    // SKIP_PLAIN (td_pack_rest): the wavefront's tiles are the same as above (its index, then every nwaves-th), FOUR at a time: their bits
    // of td_scan_tiles' masks of the tiles that are not td_pack_plain's first (four independent loads: on plain text next to none
    // is set, and reading every tile's count word to find that out was 48 us per GiB), then what depends on the tile index only
    // for all of the four that are (one more round trip), then the tiles.  When the output is too small (total > out_cap)
    // every tile is looked at: the ones behind the end of the output are nobody's otherwise.
    const bool walk_all = SKIP_PLAIN && total > a.out_cap;  // (uniform)
    // ... and in the sparse launch sequence (plain text: a tile of td_pack_rest's every few thousand) a LANE per mask word: a wavefront
    // looks at 64 words = 1024 tiles per round trip, in a grid of a few dozen workgroups instead of 16 384 that each find nothing
    // (9 us -> ? at 128 MiB, 34 us per GiB)
    const bool lane_walk = SKIP_PLAIN && (a.sparse != 0 || a.pack_dense != 0);  // (uniform; with td_pack_dense too: next to nothing is left for this kernel)
    const int n_mwords = (a.n_tiles + 15) >> 4;
    int lw_base = (int)(blockIdx.x * (K_THREADS / 64) + wv) * 64 - nwaves * 64, lw_index = 0;
    uint32_t lw_mine = 0, lw_cur = 0;
    uint64_t lw_nz = 0;
    int t0 = tile_first - 4 * nwaves;
    uint32_t cmask = 0, c_tc[4] = {0, 0, 0, 0}, c_df[4] = {0, 0, 0, 0}, c_ex[4] = {0, 0, 0, 0};
    int64_t c_base[4] = {0, 0, 0, 0};
    int tile = tile_first - nwaves;
    for (;;) {
        uint32_t tc, ex = 0;
        int64_t base, dfirst;
        bool fast_this = false;
        uint4 sx0 = cx0, sx1 = cx0, sx2 = cx0, sx3 = cx0;
        uint32_t sh0 = 0, stl = 0, sdsl = 0;
        int64_t sdpos = 0;
        if (SKIP_PLAIN && lane_walk) {
            bool done = false;
            while (!lw_cur) {
                if (!lw_nz) {
                    lw_base += nwaves * 64;
                    if (lw_base >= n_mwords) { done = true; break; }
                    lw_mine = lw_base + lane < n_mwords ? (walk_all ? 0xFFFFu : (uint32_t)a.rest_mask[lw_base + lane]) : 0u;
                    lw_nz = __ballot(lw_mine != 0u);
                    if (!lw_nz) continue;
                }
                const int l = td_ctz64(lw_nz);
                lw_nz &= lw_nz - 1ull;
                lw_cur = (uint32_t)__shfl((int)lw_mine, l);
                lw_index = lw_base + l;
            }
            if (done) break;
            const int bq = (int)td_ctz32(lw_cur);
            lw_cur &= lw_cur - 1u;
            tile = lw_index * 16 + bq;
            if (tile >= a.n_tiles) continue;
            tc = a.tile_count[tile]; ex = a.tile_extra[tile]; dfirst = (int64_t)a.tile_first_doc[tile];
            base = a.tile_base[tile] + a.chunk_pref[tile / K_SCAN_CHUNK];
            if (pk_simple(tc, base, ex, a.out_cap) || (a.pack_dense && pk_dense(tc, base, ex, a.out_cap))) continue;
        } else if constexpr (SKIP_PLAIN) {
            if (!cmask) {
                t0 += 4 * nwaves;
                if (t0 >= a.n_tiles) break;
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t0 + q * nwaves;
                    if (tq < a.n_tiles) w[q] = walk_all ? 0xFFFFu : (uint32_t)a.rest_mask[tq >> 4];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t0 + q * nwaves;
                    if (tq < a.n_tiles && ((w[q] >> (tq & 15)) & 1u)) cmask |= 1u << q;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t0 + q * nwaves;
                    if ((cmask >> q) & 1u) {
                        c_tc[q] = a.tile_count[tq]; c_ex[q] = a.tile_extra[tq]; c_df[q] = a.tile_first_doc[tq];
                        c_base[q] = a.tile_base[tq] + a.chunk_pref[tq / K_SCAN_CHUNK];
                    }
                }
                if (!cmask) continue;
            }
            const int q = (int)td_ctz32(cmask);
            cmask &= cmask - 1u;
            tile = t0 + q * nwaves;
            tc = q == 0 ? c_tc[0] : q == 1 ? c_tc[1] : q == 2 ? c_tc[2] : c_tc[3];
            ex = q == 0 ? c_ex[0] : q == 1 ? c_ex[1] : q == 2 ? c_ex[2] : c_ex[3];
            dfirst = (int64_t)(q == 0 ? c_df[0] : q == 1 ? c_df[1] : q == 2 ? c_df[2] : c_df[3]);
            base = q == 0 ? c_base[0] : q == 1 ? c_base[1] : q == 2 ? c_base[2] : c_base[3];
            if (pk_simple(tc, base, ex, a.out_cap) || (a.pack_dense && pk_dense(tc, base, ex, a.out_cap))) continue;  // (the other kernels; the closing offsets with them when this is the last tile)
        } else {
            tile += nwaves;
            if (tile >= a.n_tiles) break;
            if (tile + 2 * nwaves < a.n_tiles) PK_FETCH(tile + 2 * nwaves, C)
            const bool fast_next = tile + nwaves < a.n_tiles && PK_FAST(B);
            if (fast_next) PK_LOAD(tile + nwaves, B, n)  // (in front of this tile's stores)
            tc = tcA;
            base = baseA;
            dfirst = (int64_t)dfA;
            fast_this = fast_cur;
            sx0 = cx0; sx1 = cx1; sx2 = cx2; sx3 = cx3;
            sh0 = ch0; stl = ctl; sdsl = cdsl;
            sdpos = cdpos;
            tcA = tcB; baseA = baseB; dfA = dfB; tcB = tcC; baseB = baseC; dfB = dfC;
            fast_cur = fast_next;
            cx0 = nx0; cx1 = nx1; cx2 = nx2; cx3 = nx3; ch0 = nh0; ctl = ntl; cdsl = ndsl; cdpos = ndpos;
        }
        (void)ex;
        if (PLAIN_ONLY && !fast_this) continue;
        if (tc & TILE_DIRECT) {  // the fused tile loop wrote this tile's ids and document offsets itself
            if (tile == a.n_tiles - 1) {  // empty documents at the very end + the closing offset
                const int64_t d_end = lower_bound_i64(a.doc_offsets, a.n_docs, a.n);
                for (int64_t d = d_end + lane; d <= a.n_docs; d += 64) a.out_offsets[d] = total;
            }
            continue;
        }
        if (fast_this && !TD_STOP(61)) {
            const uint32_t g_c = tc & TILE_COUNT_MASK;
            int32_t* dst = a.out_tokens + base;
            uint32_t g_head = (uint32_t)((16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u) >> 2;
            if (g_head > g_c) g_head = g_c;
            const uint32_t g_nv = (g_c - g_head) >> 2, g_done = g_head + 4 * g_nv;
            if ((uint32_t)lane < g_nv) *reinterpret_cast<uint4*>(dst + g_head + 4 * lane) = sx0;
            if ((uint32_t)lane + 64 < g_nv) *reinterpret_cast<uint4*>(dst + g_head + 4 * (lane + 64)) = sx1;
            if ((uint32_t)lane + 128 < g_nv) *reinterpret_cast<uint4*>(dst + g_head + 4 * (lane + 128)) = sx2;
            if ((uint32_t)lane + 192 < g_nv) *reinterpret_cast<uint4*>(dst + g_head + 4 * (lane + 192)) = sx3;
            if ((uint32_t)lane < g_head) dst[lane] = (int32_t)sh0;
            if (g_done + (uint32_t)lane < g_c) dst[g_done + lane] = (int32_t)stl;
            const int64_t g_lo = (int64_t)tile * K_TILE;
            const int64_t g_hi = (g_lo + K_TILE < a.n) ? g_lo + K_TILE : a.n;
            const int64_t dm = dfirst + lane;
            if (sdpos < g_hi) a.out_offsets[dm] = base + sdsl;
            if (__all(sdpos < g_hi)) {  // more than 64 documents start in this tile
                for (int64_t d = dfirst + 64 + lane; d < a.n_docs; d += 64) {
                    if (a.doc_offsets[d] >= g_hi) break;
                    a.out_offsets[d] = base + a.doc_slot[d];
                }
            }
            if (tile == a.n_tiles - 1) {  // empty documents at the very end + the closing offset
                const int64_t d_end = lower_bound_i64(a.doc_offsets, a.n_docs, a.n);
                for (int64_t d = d_end + lane; d <= a.n_docs; d += 64) a.out_offsets[d] = total;
            }
            continue;
        }
        if constexpr (PLAIN_ONLY) continue;
        const uint32_t cnt = tc & TILE_COUNT_MASK;
        const uint32_t* src = a.stage + (size_t)tile * K_STAGE;
        const int64_t g_lo = (int64_t)tile * K_TILE;
        const int64_t g_hi = (g_lo + K_TILE < a.n) ? g_lo + K_TILE : a.n;
        PK_TICK(t_meta)
        if (!(tc & (TILE_HAS_LONG | TILE_HAS_MISS | TILE_MISS_LISTED)) || TD_STOP(60)) {  // (60: tuning aid, every tile down the plain path)
            // the first 64 documents of the tile (nearly always all of them): offsets and slots are loaded with the ids
            const int64_t dm = dfirst + lane;
            int64_t dpos = a.n;
            uint32_t dsl = 0;
            if (dm < a.n_docs) { dpos = a.doc_offsets[dm]; dsl = a.doc_slot[dm]; }
            if (base + cnt <= a.out_cap) {
                // 16-byte stores to the (arbitrarily placed) destination: single ids up to its next 16-byte boundary, then
                // four ids per lane (the staging side is read with dword-aligned 16-byte loads); all loads of up to 1024 ids
                // first, then the stores
                int32_t* dst = a.out_tokens + base;
                uint32_t head = (uint32_t)((16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u) >> 2;
                if (head > cnt) head = cnt;
                const uint32_t nv = (cnt - head) >> 2;
                const uint32_t done = head + 4 * nv;
                uint32_t h0 = 0, tl = 0;
                if ((uint32_t)lane < head) h0 = src[lane];
                if (done + (uint32_t)lane < cnt) tl = src[done + lane];
                for (uint32_t v0 = 0; v0 < nv; v0 += 256) {
                    uint4 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t v = v0 + q * 64 + lane;
                        if (v < nv) __builtin_memcpy(&x[q], src + head + 4 * v, 16);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t v = v0 + q * 64 + lane;
                        if (v < nv) *reinterpret_cast<uint4*>(dst + head + 4 * v) = x[q];
                    }
                }
                if ((uint32_t)lane < head) dst[lane] = (int32_t)h0;
                if (done + (uint32_t)lane < cnt) dst[done + lane] = (int32_t)tl;
            }
            if (dpos < g_hi) a.out_offsets[dm] = base + dsl;
            if (__all(dpos < g_hi)) {  // more than 64 documents start in this tile
                for (int64_t d = dfirst + 64 + lane; d < a.n_docs; d += 64) {
                    if (a.doc_offsets[d] >= g_hi) break;
                    a.out_offsets[d] = base + a.doc_slot[d];
                }
            }
        } else {
            // Tiles with markers, PK_G rows of 64 slots at a time.  On this hardware loads and stores share one in-order
            // counter, so a wait for a load also waits for every store issued before it: a row-by-row sweep (load the
            // merged ids of the row's pieces, store them, next row) paid a full store round trip per row.  Here a group's
            // loads all come first (its slots, the sizes of its long pieces), then the offsets (scans, no memory), then all
            // its plain ids are stored; the merged pieces only go on a list in LDS (destination, position, ids) that is
            // worked off one piece per lane, four ids per round trip, when it fills up or the tile ends.
            unsigned long long* const elist = s_elist[wv];
            uint32_t ecount = 0;  // (uniform) pieces on the list
            const uint64_t lt = (1ull << lane) - 1ull;
            auto flush = [&]() {
                PK_TICK(t_list)
                wave_sync();
                for (uint32_t c0 = 0; c0 < ecount; c0 += 64) {
                    const unsigned long long e = c0 + lane < ecount ? elist[c0 + lane] : 0ull;
                    const uint32_t n = (uint32_t)e & 127u;
                    const uint32_t* ps = marker_ids(a, (uint32_t)tile, (uint32_t)e);
                    const int64_t o = base + (int64_t)(uint32_t)(e >> 32);
                    for (uint32_t j = 0; __any(j < n); j += 4) {
                        uint32_t t[4];
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) t[q] = j + q < n ? ps[j + q] : 0u;
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q)
                            if (j + q < n && o + j + q < a.out_cap) a.out_tokens[o + j + q] = (int32_t)t[q];
                    }
                }
                ecount = 0;
                wave_sync();
                PK_TICK(t_flush)
            };
            // the documents that start in this tile, 64 at a time, in slot order (they are consecutive from the tile's first one)
            int64_t dnext = dfirst, dmine = 0;
            uint32_t dslot = 0xFFFFFFFFu;
            bool dfull = false;
            auto load_docs = [&]() {
                dmine = dnext + lane;
                const bool ok = dmine < a.n_docs && a.doc_offsets[dmine] < g_hi;
                dslot = ok ? a.doc_slot[dmine] : 0xFFFFFFFFu;
                dfull = __all(ok);
                dnext += 64;
            };
            load_docs();
            uint32_t carry = 0;  // ids of the rows above
            for (uint32_t r0 = 0; r0 * 64u < cnt; r0 += PK_G) {
                uint32_t v[PK_G], sz[PK_G], off[PK_G];
                uint32_t lmask = 0, mmask = 0;  // bit q: my slot of row r0 + q is a long piece / a merged piece
#pragma unroll
                for (int q = 0; q < PK_G; ++q) {
                    const uint32_t k = (r0 + q) * 64u + lane;
                    v[q] = k < cnt ? src[k] : 0u;
                }
#pragma unroll
                for (int q = 0; q < PK_G; ++q) {
                    const uint32_t k = (r0 + q) * 64u + lane;
                    const bool is_long = k < cnt && (v[q] & TOK_LONGREF) != 0, is_miss = k < cnt && !is_long && (v[q] & TOK_MISS);
                    lmask |= (is_long ? 1u : 0u) << q;
                    mmask |= (is_miss ? 1u : 0u) << q;
                    sz[q] = k < cnt ? (is_miss ? (v[q] & 127u) : 1u) : 0u;
                }
                PK_TICK(t_rows)
                const bool anylong = __any(lmask != 0);
                if (anylong) {
#pragma unroll
                    for (int q = 0; q < PK_G; ++q)
                        if ((lmask >> q) & 1u) sz[q] = a.long_list[v[q] & 0x7FFFFFFFu].ntok;
                }
#pragma unroll
                for (int q = 0; q < PK_G; ++q) {
                    const uint32_t r = r0 + q, k = r * 64u + lane;
                    const uint32_t incl = __any(((lmask | mmask) >> q) & 1u) ? wave_incl_scan(sz[q], lane)
                                        : ((r * 64u < cnt && cnt - r * 64u < 64u && k >= cnt) ? cnt - r * 64u : (r * 64u < cnt ? (uint32_t)lane + 1u : 0u));  // a row of plain ids
                    off[q] = carry + incl - sz[q];  // ids of this tile in front of my slot
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                PK_TICK(t_scan)
#pragma unroll
                for (int q = 0; q < PK_G; ++q) {  // plain ids
                    const uint32_t k = (r0 + q) * 64u + lane;
                    const int64_t o = base + off[q];
                    if (k < cnt && !(((lmask | mmask) >> q) & 1u) && o < a.out_cap) a.out_tokens[o] = (int32_t)v[q];
                }
                PK_TICK(t_plain)
#pragma unroll
                for (int q = 0; q < PK_G; ++q) {  // merged pieces: on the list
                    const bool m = (mmask >> q) & 1u;
                    const uint64_t bm = __ballot(m);
                    if (bm) {
                        if (m) elist[ecount + (uint32_t)__popcll((unsigned long long)(bm & lt))] = ((unsigned long long)off[q] << 32) | (v[q] & 0x1FFFFFFFu);  // (ids | position or seat << 7 | TOK_DUPREF)
                        ecount += (uint32_t)__popcll((unsigned long long)bm);
                        if (ecount > (uint32_t)PK_ECAP - 64u) flush();
                    }
                }
                if (anylong) {  // long pieces: the whole wavefront copies
#pragma unroll
                    for (int q = 0; q < PK_G; ++q) {
                        for (uint64_t lb = __ballot((lmask >> q) & 1u); lb; lb &= lb - 1ull) {
                            const int l = (int)td_ctz64(lb);
                            const LongEntry le = a.long_list[(uint32_t)__shfl((int)v[q], l) & 0x7FFFFFFFu];
                            const int64_t lo = base + (int64_t)(uint32_t)__shfl((int)off[q], l);
                            const uint32_t* ps = a.pool + le.pool_off;
                            for (uint32_t j = lane; j < le.ntok; j += 64)
                                if (lo + j < a.out_cap) a.out_tokens[lo + j] = (int32_t)ps[j];
                        }
                    }
                }
                PK_TICK(t_list)
                for (;;) {  // documents whose first slot lies in this group of rows
                    uint32_t pref = 0;
#pragma unroll
                    for (int q = 0; q < PK_G; ++q) {
                        const uint32_t pq = (uint32_t)__shfl((int)off[q], (int)(dslot & 63u));
                        if ((dslot >> 6) == r0 + (uint32_t)q) pref = pq;
                    }
                    if (dslot != 0xFFFFFFFFu && (dslot >> 6) >= r0 && (dslot >> 6) < r0 + (uint32_t)PK_G) a.out_offsets[dmine] = base + pref;
                    const uint32_t last = (uint32_t)__shfl((int)dslot, 63);
                    if (dfull && (last >> 6) < r0 + (uint32_t)PK_G) { load_docs(); continue; }  // all 64 used up: the next ones may start in this group too
                    break;
                }
                PK_TICK(t_docs)
            }
            flush();
#ifdef TD_PACK_TIMING
            ++n_mt;
#endif
        }
#ifdef TD_PACK_TIMING
        if (!(tc & (TILE_HAS_LONG | TILE_HAS_MISS | TILE_MISS_LISTED))) { PK_TICK(t_plainpath) ++n_pt; }
#endif
        if (tile == a.n_tiles - 1) {  // empty documents at the very end + the closing offset
            const int64_t d_end = lower_bound_i64(a.doc_offsets, a.n_docs, a.n);
            for (int64_t d = d_end + lane; d <= a.n_docs; d += 64) a.out_offsets[d] = total;
        }
    }
#ifdef TD_PACK_TIMING
    if (lane == 0 && (blockIdx.x % 257) == 0 && wv == 0)
        printf("pack wave b%d: total %llu meta %llu plainpath %llu (%llu tiles) | marker tiles %llu: rows %llu scan %llu plain %llu list %llu flush %llu docs %llu\n", (int)blockIdx.x,
               (unsigned long long)(__builtin_readcyclecounter() - t_total0), t_meta, t_plainpath, n_pt, n_mt, t_rows, t_scan, t_plain, t_list, t_flush, t_docs);
#endif
}
__global__ __launch_bounds__(K_THREADS, 4) void td_pack_tokens(const EncodeArgs a) {
    __shared__ unsigned long long s_elist[K_THREADS / 64][PK_ECAP];  // dst offset << 32 | tile position << 7 | ids
    pack_body<false, false>(a, s_elist);
}
// The tiles without long pieces (every slot an id or the marker of a merged piece, at most 1024 slots, inside the output:
// pk_simple) nearly as a copy: a WAVEFRONT per tile in a grid of short workgroups (no persistent loop, no pipeline, no
// alignment of the stores to the destination); what depends on the tile index only goes out in ONE round trip, then all
// sixteen dwords per lane and the documents, then the stores.  A row of 64 slots that holds a marker takes one add-scan
// (a marker stands for its piece's ids: 0..64 of them, in merge_out at the piece's own bytes' slots).
// (Round 4: torch's strided copy moves 840 of every 4160 slots to a dense array at 5.3 TB/s on this box —
// tools/gpu_copy_ceiling.py — where the pipelined path of pack_body reaches 3.1; a first form of this kernel with a workgroup
// per tile and three dependent round trips before the stores was latency-bound at 2.7; this one runs the plain tiles of
// 1 GiB of English at 5.5.)
#ifndef TD_PACK_PLAIN_WAVES
#define TD_PACK_PLAIN_WAVES 6  // (measured on 1 GiB of English: 0.392 ms at 4 wavefronts per SIMD, 0.371 at 6, 0.646 at 8: its marker rows spill there)
#endif
__global__ __launch_bounds__(K_THREADS, TD_PACK_PLAIN_WAVES) void td_pack_plain(const EncodeArgs a) {
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * (K_THREADS / 64);
    for (int tile = blockIdx.x * (K_THREADS / 64) + (threadIdx.x >> 6); tile < a.n_tiles; tile += nwaves) {
        const uint32_t tc = a.tile_count[tile], ex = a.tile_extra[tile];
        const int64_t tb = a.tile_base[tile], cp = a.chunk_pref[tile / K_SCAN_CHUNK];
        const int64_t dfirst = (int64_t)a.tile_first_doc[tile];
        const uint32_t cnt = tc & TILE_COUNT_MASK;
        const int64_t base = tb + cp;
        if (!pk_simple(tc, base, ex, a.out_cap)) continue;  // (td_pack_rest's)
        const uint32_t* src = a.stage + (size_t)tile * K_STAGE;
        int32_t* dst = a.out_tokens + base;
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t k = (uint32_t)lane + 64u * q;
            v[q] = k < cnt ? src[k] : 0u;
        }
        const int64_t g_lo = (int64_t)tile * K_TILE;
        const int64_t g_hi = (g_lo + K_TILE < a.n) ? g_lo + K_TILE : a.n;
        const bool marks = (tc & TILE_MISS_LISTED) != 0;  // (uniform)
        if (!marks) {
            for (int64_t d = dfirst + lane; d < a.n_docs; d += 64) {
                if (a.doc_offsets[d] >= g_hi) break;
                a.out_offsets[d] = base + a.doc_slot[d];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t k = (uint32_t)lane + 64u * q;
                if (k < cnt) dst[k] = (int32_t)v[q];
            }
        } else {
            uint32_t sh[16], carry = 0;  // ids of the tile in front of my slot of row q, minus the slot's index (modulo 2^32: a marker may stand for no id)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bool m = (v[q] & 0xC0000000u) == TOK_MISS;
                uint32_t excl = 0, rowsum = 0;
                if (__ballot(m)) {
                    const uint32_t dlt = m ? (v[q] & 127u) - 1u : 0u;
                    const uint32_t incl = wave_incl_scan(dlt, lane);
                    excl = incl - dlt;
                    rowsum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                sh[q] = carry + excl;
                carry += rowsum;
            }
            for (int64_t d = dfirst + lane; __any(d < a.n_docs); d += 64) {  // (every lane stays in: the shuffles below)
                const bool mine = d < a.n_docs && a.doc_offsets[d < a.n_docs ? d : 0] < g_hi;
                if (!__any(mine)) break;
                const uint32_t ks = mine ? a.doc_slot[d] : 0u;
                uint32_t s0 = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t t = (uint32_t)__shfl((int)sh[q], (int)(ks & 63u));
                    if ((ks >> 6) == (uint32_t)q) s0 = t;
                }
                if (mine) a.out_offsets[d] = base + (int64_t)(int32_t)(ks + (ks < cnt ? s0 : carry));  // (a document that starts behind the tile's last slot)
                if (!__all(mine)) break;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t k = (uint32_t)lane + 64u * q;
                if (k < cnt) {
                    int32_t* o = dst + (int64_t)(int32_t)(k + sh[q]);
                    if ((v[q] & 0xC0000000u) == TOK_MISS) {
                        const uint32_t* mo = marker_ids(a, (uint32_t)tile, v[q]);
                        const uint32_t nt = v[q] & 127u;
                        for (uint32_t i = 0; i < nt; ++i) o[i] = (int32_t)mo[i];
                    } else {
                        *o = (int32_t)v[q];
                    }
                }
            }
        }
        if (tile == a.n_tiles - 1) {  // empty documents at the very end + the closing offset
            const int64_t total = a.tile_base[a.n_tiles];
            const int64_t d_end = lower_bound_i64(a.doc_offsets, a.n_docs, a.n);
            for (int64_t d = d_end + lane; d <= a.n_docs; d += 64) a.out_offsets[d] = total;
        }
    }
}
// td_pack_dense (round 6, the dense launch sequence): the tiles of mixed-script text and source code — dozens of merged pieces, a long piece
// or two — in td_pack_plain's shape instead of td_pack_rest's persistent pipeline: a WAVEFRONT per tile in a grid of short workgroups, what
// depends on the tile index in ONE round trip, then all sixteen rows of slots (+ the tile's documents, + the sizes of its long pieces), then
// the offsets (a DPP add-scan per row that holds a marker), then all plain ids, the merged pieces through a list in LDS (a lane per piece, four
// ids a round trip; a repeat's ids are read where the piece it repeats has them), the long pieces by the whole wavefront.  td_pack_rest went
// through a tile in two groups of eight rows, every group's loads behind the stores of the one before (the shared counter), and its wavefronts
// carried state from tile to tile: ~10 dependent round trips per tile where this one has ~6.
// (MULTI: the tile has more than 1024 slots — source code, punctuation — and goes through them sixteen rows at a time; as ONE loop for both
// cases the usual single pass lost a fifth of its speed: 282 -> 343 us per 256 MiB of mixed-script text)
template <bool MULTI>
__device__ __forceinline__ void pack_dense_tile(const EncodeArgs& a, unsigned long long* const elist, const int tile, const uint32_t cnt, const int64_t base,
                                                const int64_t dfirst, const int lane) {
    const uint32_t* src = a.stage + (size_t)tile * K_STAGE;
    const int64_t g_lo = (int64_t)tile * K_TILE;
    const int64_t g_hi = (g_lo + K_TILE < a.n) ? g_lo + K_TILE : a.n;
    uint32_t ecount = 0;  // (uniform) merged pieces on the list
    const uint64_t lt = (1ull << lane) - 1ull;
    auto flush = [&]() {
        wave_sync();
        for (uint32_t c0 = 0; c0 < ecount; c0 += 64) {
            const unsigned long long e = c0 + lane < ecount ? elist[c0 + lane] : 0ull;
            const uint32_t n = (uint32_t)e & 127u;
            const uint32_t* ps = marker_ids(a, (uint32_t)tile, (uint32_t)e);
            const int64_t o = base + (int64_t)(uint32_t)(e >> 32);
            for (uint32_t j = 0; __any(j < n); j += 4) {
                uint32_t t[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) t[u] = j + u < n ? ps[j + u] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u)
                    if (j + u < n) a.out_tokens[o + j + u] = (int32_t)t[u];
            }
        }
        ecount = 0;
        wave_sync();
    };
    uint32_t carry = 0;  // ids of the segments (and rows) in front
    // (a tile of source code or punctuation can hold more than 1024 pieces: sixteen rows of 64 slots at a time — nearly always ONE segment)
    for (uint32_t s0 = 0;; s0 += 1024u) {
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t k = s0 + (uint32_t)lane + 64u * q;
            v[q] = k < cnt ? src[k] : 0u;
        }
        // the first 64 documents of the tile (nearly always all of them), requested with the slots
        int64_t dm = dfirst + lane;
        bool dmine = dm < a.n_docs && a.doc_offsets[dm < a.n_docs ? dm : 0] < g_hi;
        uint32_t dks = dmine ? a.doc_slot[dm] : 0u;
        uint32_t lmask = 0, mmask = 0;  // bit q: my slot of row q is a long piece / a merged piece
        uint32_t off[16];               // first: the slot's size; then: ids of the tile in front of it
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t k = s0 + (uint32_t)lane + 64u * q;
            const bool is_long = k < cnt && (v[q] & TOK_LONGREF) != 0u, is_miss = k < cnt && !is_long && (v[q] & TOK_MISS) != 0u;
            lmask |= (is_long ? 1u : 0u) << q;
            mmask |= (is_miss ? 1u : 0u) << q;
            off[q] = k < cnt ? (is_miss ? (v[q] & 127u) : 1u) : 0u;
        }
        const bool anylong = __any(lmask != 0u);
        if (anylong) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((lmask >> q) & 1u) off[q] = a.long_list[v[q] & 0x7FFFFFFFu].ntok;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t r64 = s0 + 64u * (uint32_t)q, k = r64 + (uint32_t)lane;
            const uint32_t sz = off[q];
            const uint32_t incl = __any(((lmask | mmask) >> q) & 1u) ? wave_incl_scan(sz, lane)
                                : ((r64 < cnt && cnt - r64 < 64u && k >= cnt) ? cnt - r64 : (r64 < cnt ? (uint32_t)lane + 1u : 0u));  // a row of plain ids
            off[q] = carry + incl - sz;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        const bool last_seg = !MULTI || s0 + 1024u >= cnt;
        // the documents that start in the tile: the offset of a document's slot lives in lane (slot mod 64), row (slot / 64) of its segment
        for (;;) {
            const uint32_t rel = dks - s0;
            uint32_t pref = carry;  // (a document that starts behind the tile's last slot: behind all of the tile's ids)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t t = (uint32_t)__shfl((int)off[q], (int)(rel & 63u));
                if ((rel >> 6) == (uint32_t)q) pref = t;
            }
            const bool in_seg = dmine && dks < cnt && dks >= s0 && rel < 1024u, behind = dmine && dks >= cnt && last_seg;
            if (in_seg || behind) a.out_offsets[dm] = base + (int64_t)(behind ? carry : pref);
            if (!__all(dmine)) break;  // more than 64 documents start in this tile: the next 64
            dm += 64;
            dmine = dm < a.n_docs && a.doc_offsets[dm < a.n_docs ? dm : 0] < g_hi;
            dks = dmine ? a.doc_slot[dm] : 0u;
            if (!__any(dmine)) break;
        }
        // plain ids
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t k = s0 + (uint32_t)lane + 64u * q;
            if (k < cnt && !(((lmask | mmask) >> q) & 1u)) a.out_tokens[base + off[q]] = (int32_t)v[q];
        }
        // merged pieces: onto the list in LDS, worked off a lane per piece
        if (__any(mmask != 0u)) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bool m = (mmask >> q) & 1u;
                const uint64_t bm = __ballot(m);
                if (bm) {
                    if (m) elist[ecount + (uint32_t)__popcll((unsigned long long)(bm & lt))] = ((unsigned long long)off[q] << 32) | (v[q] & 0x1FFFFFFFu);
                    ecount += (uint32_t)__popcll((unsigned long long)bm);
                    if (ecount > (uint32_t)PK_ECAP - 64u) flush();
                }
            }
        }
        if (anylong) {  // long pieces: the whole wavefront copies
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                for (uint64_t lb = __ballot((lmask >> q) & 1u); lb; lb &= lb - 1ull) {
                    const int l = (int)td_ctz64(lb);
                    const LongEntry le = a.long_list[(uint32_t)__shfl((int)v[q], l) & 0x7FFFFFFFu];
                    const int64_t lo = base + (int64_t)(uint32_t)__shfl((int)off[q], l);
                    const uint32_t* ps = a.pool + le.pool_off;
                    for (uint32_t j = lane; j < le.ntok; j += 64) a.out_tokens[lo + j] = (int32_t)ps[j];
                }
            }
        }
        if (last_seg) break;
    }
    if (ecount) flush();
    if (tile == a.n_tiles - 1) {  // empty documents at the very end + the closing offset
        const int64_t total = a.tile_base[a.n_tiles];
        const int64_t d_end = lower_bound_i64(a.doc_offsets, a.n_docs, a.n);
        for (int64_t d = d_end + lane; d <= a.n_docs; d += 64) a.out_offsets[d] = total;
    }
}
#ifndef TD_PACK_DENSE_WAVES
#define TD_PACK_DENSE_WAVES 5
#endif
// (two launches: the tiles of at most 1024 slots, then the ones above — both cases in ONE kernel cost the usual one a quarter of its speed: 282 -> 370 us)
template <bool MULTI>
__global__ __launch_bounds__(K_THREADS, MULTI ? 4 : TD_PACK_DENSE_WAVES) void td_pack_dense(const EncodeArgs a) {
    __shared__ unsigned long long s_elist[K_THREADS / 64][PK_ECAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (K_THREADS / 64) + wv;
    if (tile >= a.n_tiles) return;
    const uint32_t tc = a.tile_count[tile], ex = a.tile_extra[tile];
    const int64_t tb = a.tile_base[tile], cp = a.chunk_pref[tile / K_SCAN_CHUNK];
    const int64_t dfirst = (int64_t)a.tile_first_doc[tile];
    const uint32_t cnt = tc & TILE_COUNT_MASK;
    const int64_t base = tb + cp;
    if (!pk_dense(tc, base, ex, a.out_cap)) return;
    if ((cnt > 1024u) != MULTI) return;
    pack_dense_tile<MULTI>(a, s_elist[wv], tile, cnt, base, dfirst, lane);
}
__global__ __launch_bounds__(K_THREADS) void td_pack_rest(const EncodeArgs a) {
    __shared__ unsigned long long s_elist[K_THREADS / 64][PK_ECAP];
    pack_body<false, true>(a, s_elist);
}

// ------------------------------------------------------------------ td_small_encode ---------
// The whole path in ONE launch for inputs of at most 4 KiB (a chat message, a line of code: the calls the reference's
// tests/performance_benchmark.py:239-387 times): one workgroup reads the text and the document offsets straight out of a
// pinned host buffer, classifies, finds the piece boundaries (every lane from the first provable sync point of its 16
// bytes, scan_piece on the class bytes in LDS), looks the pieces up, merges the misses one lane per piece (mg_round), packs
// and writes ids + offsets + status back into pinned host memory, and releases a sequence number the host spins on.
// (The batch pipeline is thirteen launches: ~160 us for "Hello, world!", against microseconds on the reference's CPU path.)
// A piece above 1 KiB (or more than SM_LONG_MAX pieces above 64 bytes) makes the kernel hand the call back (fallback = 1) to the general path.
constexpr int SM_MAXBYTES = K_TILE;
constexpr int SM_LONG_MAX = 64;  // pieces above 64 bytes td_small_encode merges itself (more: the general path)
struct SmallCfAcc {  // scanner accessor over the class bytes in LDS; positions >= n read as end of subject
    using pos_t = int;
    const uint8_t* cfs;
    const uint8_t* txt;
    int lim;
    __device__ __forceinline__ uint32_t cf(int i) const { return cfs[i]; }
    __device__ __forceinline__ uint32_t byte(int i) const { return txt[i]; }
};
__device__ __forceinline__ void small_encode_body(const SmallArgs& a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[SM_MAXBYTES + 80];
    __shared__ __attribute__((aligned(16))) uint8_t s_cf[SM_MAXBYTES + 80];
    __shared__ uint32_t s_doc[SM_MAXBYTES / 32 + 4];
    __shared__ uint32_t s_start[SM_MAXBYTES / 32 + 4];
    __shared__ uint16_t s_plist[SM_MAXBYTES + 8];
    __shared__ __attribute__((aligned(16))) uint32_t s_tok[SM_MAXBYTES + 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[K_THREADS * MG_UNIT];
    __shared__ __attribute__((aligned(16))) uint32_t s_ids[K_THREADS * MG_UNIT];
    __shared__ uint16_t s_miss[SM_MAXBYTES / 2 + 8];
    __shared__ uint32_t s_off[K_THREADS];
    __shared__ uint16_t s_valid[K_THREADS];
    __shared__ int32_t s_byteid[256];
    __shared__ uint32_t s_wave[8];
    __shared__ uint32_t s_nmiss, s_err, s_errpos, s_fallback;
    __shared__ uint32_t s_nlong;
    __shared__ uint16_t s_long[SM_LONG_MAX];  // pieces of 65..1024 bytes (indices into s_plist)
    __shared__ uint16_t s_doff[SM_MAXDOCS + 2];  // the document offsets (<= n <= 4096): read from the pinned host buffer ONCE, together with the text —
                                                 // they were read twice, each time a PCIe round trip of its own behind a barrier (~2 us of a 14 us call)

    const int tid = threadIdx.x;
    const Tables T = uniform_tables(a.Tp);
    const int n = a.n;
    for (int d = tid; d <= a.n_docs && d <= SM_MAXDOCS; d += K_THREADS) {
        const int64_t p = a.doc_offsets[d];
        s_doff[d] = (uint16_t)(p < 0 ? n : p > n ? n : p);
    }
    for (int q = tid; q < 256; q += K_THREADS) s_byteid[q] = T.byte_id[q];
    // ---- text, document bits ----
    for (int q = tid; q < (SM_MAXBYTES + 80) / 16; q += K_THREADS) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q * 16 + 16 <= n) v = reinterpret_cast<const uint4*>(a.text)[q];
        else if (q * 16 < n)
            for (int k = 0; k < 16 && q * 16 + k < n; ++k) reinterpret_cast<uint8_t*>(&v)[k] = a.text[q * 16 + k];
        reinterpret_cast<uint4*>(s_txt)[q] = v;
    }
    for (int q = tid; q < SM_MAXBYTES / 32 + 4; q += K_THREADS) { s_doc[q] = 0; s_start[q] = 0; }
    for (int q = tid; q < (SM_MAXBYTES + 64) / 4; q += K_THREADS) reinterpret_cast<uint4*>(s_tok)[q] = make_uint4(TOK_NONE, TOK_NONE, TOK_NONE, TOK_NONE);
    if (tid == 0) { s_nmiss = 0; s_err = 0; s_errpos = 0; s_fallback = 0; s_nlong = 0; }
    __syncthreads();
    for (int d = tid; d < a.n_docs; d += K_THREADS) {
        const int p = s_doff[d];
        if (p < n) atomicOr(&s_doc[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
    auto fail = [&](int code, int pos) { if (atomicCAS(&s_err, 0u, (uint32_t)code) == 0u) s_errpos = (uint32_t)pos; };
    // ---- class + flags of every byte (positions >= n: end of subject) ----
    {
        LdsSrc src;
        src.txt = s_txt; src.docw = s_doc; src.lo = 0; src.hi = n;
        for (int i = tid; i < SM_MAXBYTES + 80; i += K_THREADS) {
            uint32_t v = F_DOC;
            if (i < n) {
                v = classify_at(T, src, i);
                if (src.doc(i)) v |= F_DOC;
            }
            s_cf[i] = (uint8_t)v;
        }
    }
    __syncthreads();
    // ---- piece boundaries: every lane from the first provable sync point of its 16 bytes ----
    {
        const SmallCfAcc A{s_cf, s_txt, n + 64};
        const int c0 = tid * K_CHUNK, c1 = c0 + K_CHUNK;
        int s = -1;
        if (c0 < n) {
            const int cend = c1 < n ? c1 : n;
            for (int i = c0; i < cend; ++i)
                if (is_sync(i > 0 ? s_cf[i - 1] : 0u, s_cf[i], T.pat_flags)) { s = i; break; }
        }
        for (int p = s; p >= 0 && p < n;) {
            if (p >= c1 && is_sync(s_cf[p - 1], s_cf[p], T.pat_flags)) break;  // the lane owning p starts there
            atomicOr(&s_start[p >> 5], 1u << (p & 31));
            p = scan_piece(A, p, T.pat_flags);
        }
    }
    __syncthreads();
    // ---- dense piece list ----
    uint32_t np_total;
    {
        const int c0 = tid * K_CHUNK;
        uint32_t smask = (c0 < n) ? (s_start[c0 >> 5] >> (c0 & 31)) & 0xFFFFu : 0u;
        if (c0 + K_CHUNK > n && c0 < n) smask &= (1u << (n - c0)) - 1u;
        const uint32_t pbase = block_excl_scan(__popc(smask), s_wave, np_total);
        uint32_t k = pbase;
        while (smask) {
            const int b = __ffs(smask) - 1;
            smask &= smask - 1;
            s_plist[k++] = (uint16_t)(c0 + b);
        }
        if (tid == 0) s_plist[np_total] = (uint16_t)n;
    }
    __syncthreads();
    // ---- lookup: a hit writes its id at the piece's first byte, a miss goes on the list ----
    for (uint32_t k = tid; k < np_total; k += K_THREADS) {
        const int i = s_plist[k];
        const uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)i;
        const uint8_t* pb = s_txt + i;
        if (len == 1) {
            const int32_t id = s_byteid[pb[0]];
            if (id >= T.pseudo_base) fail(TD_E_UNKNOWN_BYTE, i);
            s_tok[i] = (uint32_t)id;
            continue;
        }
        if (len > (uint32_t)K_MAXSHORT) {
            // (round 5: pieces of 65..1024 bytes are merged here too, a wavefront per piece behind the short ones — the general path
            // costs a dozen launches and four copies, 0.37 ms for a line of a hundred blanks; above that, or more than SM_LONG_MAX of them: hand back)
            uint32_t at = (uint32_t)SM_LONG_MAX;
            if (len <= (uint32_t)LP_MEDIUM) at = atomicAdd(&s_nlong, 1u);
            if (at < (uint32_t)SM_LONG_MAX) s_long[at] = (uint16_t)k; else s_fallback = 1;
            continue;
        }
        int32_t r = NO_RANK;
        if (a.use_fastpath) {
            auto get = [pb](uint32_t q) { return (uint32_t)pb[q]; };
            uint64_t key = 0;
            if (len <= 8) for (uint32_t q = 0; q < len; ++q) key |= (uint64_t)pb[q] << (8 * q);
            else key = hash_bytes(get, len);
            r = piece_lookup(T, key, len, get);
        }
        if (r != NO_RANK) s_tok[i] = (uint32_t)r;
        else s_miss[atomicAdd(&s_nmiss, 1u)] = (uint16_t)k;
    }
    __syncthreads();
    // ---- merge the misses, one lane per piece: pieces of at most 16 bytes 256 at a time, longer ones 64 at a time ----
    {
        const uint32_t nmiss = s_nmiss;
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t u = pass == 0 ? 1u : 4u;
            const uint32_t per = (uint32_t)K_THREADS / u;
            // (both passes walk the whole list and take the pieces of their class: the list is short)
            uint32_t done = 0;
            while (done < nmiss) {
                // this batch: the next `per` list entries of the class
                MergeState st;
                st.alive = 0; st.t = (uint32_t)tid; st.len = 0;
                int pos = 0;
                uint32_t taken = 0, scan = done;
                // every lane walks the list the same way (uniform), lane (taken * u) takes the piece
                while (scan < nmiss && taken < per) {
                    const uint32_t k = s_miss[scan];
                    const uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)s_plist[k];
                    if ((len <= 16u) == (pass == 0)) {
                        if ((uint32_t)tid == taken * u) { st.len = len; pos = s_plist[k]; }
                        ++taken;
                    }
                    ++scan;
                }
                done = scan;
                if (st.len) {
                    st.alive = st.len >= 64u ? ~0ull : ((1ull << st.len) - 1ull);
                    for (uint32_t j = 0; j < st.len; ++j) mg_put(T, s_byteid, s_keys, s_ids, st, j, s_txt[pos + j], s_txt[pos + j + 1]);
                    mg_pad(s_keys, st);
                }
                for (;;) {
                    const bool more = mg_round_t<uint64_t>(T, s_keys, s_ids, st);
                    if (!__any(more)) break;
                }
                if (st.len) {
                    for (uint64_t al = st.alive; al; al &= al - 1ull) {
                        const uint32_t j = (uint32_t)td_ctz64(al);
                        const uint32_t id = s_ids[mg_slot(st.t, j)];
                        if ((int32_t)id >= T.pseudo_base) fail(TD_E_UNKNOWN_BYTE, pos + (int)j);
                        s_tok[pos + j] = id;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    // ---- pieces of 65..1024 bytes: a wavefront each, parts dense in LDS (the arrays of the lane-per-piece merge are free now) ----
    if (s_nlong && !s_fallback) {
        const int wv = tid >> 6, lane = tid & 63;
        uint32_t* const base = (wv < 2 ? s_keys : s_ids) + (wv & 1) * 2 * LP_MEDIUM;
        static_assert(K_THREADS * MG_UNIT >= 4 * LP_MEDIUM, "two wavefronts' ids + ranks per array");
        volatile uint32_t* const id = base;
        volatile uint32_t* const rk = base + LP_MEDIUM;
        const uint32_t nlong = s_nlong < (uint32_t)SM_LONG_MAX ? s_nlong : (uint32_t)SM_LONG_MAX;
        for (uint32_t e = (uint32_t)wv; e < nlong; e += K_THREADS / 64) {
            const uint32_t k = s_long[e];
            const int pos = s_plist[k];
            const uint32_t len = (uint32_t)s_plist[k + 1] - (uint32_t)pos;
            const uint8_t* pb = s_txt + pos;
            int32_t whole = NO_RANK;
            if (a.use_fastpath && len <= T.max_token_len) {  // whole-piece table first (CoreBPE::encode, tiktoken.cpp:209-215)
                if (lane == 0) {
                    auto get = [pb](uint32_t q) { return (uint32_t)pb[q]; };
                    whole = piece_lookup(T, hash_bytes(get, len), len, get);
                }
                whole = __shfl(whole, 0);
            }
            uint32_t m = 1;
            if (whole != NO_RANK) {
                if (lane == 0) id[0] = (uint32_t)whole;
            } else {
                for (uint32_t q = (uint32_t)lane; q < len; q += 64u) {
                    const uint32_t b = pb[q];
                    id[q] = (uint32_t)s_byteid[b];
                    rk[q] = (q + 1 < len) ? (uint32_t)T.byte_pair[(b << 8) | pb[q + 1]] : (uint32_t)NO_RANK;
                }
                wave_sync_lds();
                m = lp_merge_batched(T, id, rk, len, lane);
            }
            wave_sync_lds();
            for (uint32_t q = (uint32_t)lane; q < m; q += 64u) {
                const uint32_t v = id[q];
                if ((int32_t)v >= T.pseudo_base) fail(TD_E_UNKNOWN_BYTE, pos);
                s_tok[pos + q] = v;  // (a piece has at most as many ids as bytes: its own bytes' places, in order)
            }
            wave_sync_lds();
        }
    }
    __syncthreads();
    // ---- pack: ids in byte order, document offsets ----
    {
        const int c0 = tid * K_CHUNK;
        uint32_t vmask = 0;
#pragma unroll
        for (int k = 0; k < K_CHUNK; ++k) vmask |= (s_tok[c0 + k] != TOK_NONE) ? (1u << k) : 0u;
        uint32_t total;
        const uint32_t off = block_excl_scan(__popc(vmask), s_wave, total);
        s_off[tid] = off;
        s_valid[tid] = (uint16_t)vmask;
        uint32_t o = off;
        if (!s_fallback && !s_err)
            for (uint32_t m = vmask; m; m &= m - 1) a.out_tokens[o++] = (int32_t)s_tok[c0 + __ffs(m) - 1];
        __syncthreads();
        for (int d = tid; d <= a.n_docs; d += K_THREADS) {
            const int p = s_doff[d];
            a.out_offsets[d] = p >= n ? (int64_t)total : (int64_t)(s_off[p >> 4] + __popc((uint32_t)s_valid[p >> 4] & ((1u << (p & 15)) - 1u)));
        }
        __syncthreads();
        if (tid == 0) {
            a.status->n_tokens = total;
            a.status->err = (int)s_err;
            a.status->err_pos = (long long)s_errpos;
            a.status->fallback = (int)s_fallback;
            __hip_atomic_store(&a.status->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(K_THREADS) void td_small_encode(const SmallArgs a) { small_encode_body(a); }

// The same workgroup, RESIDENT for a while (round 6): calls that follow each other within the idle time — a loop over
// chat messages, the reference's latency benchmark — find a kernel that is already running and pay neither a launch (~5 us on the host, ~5
// on the device) nor the set-up of a fresh one.  The host writes a request into the pinned input buffer (header: what SmallArgs carries,
// then offsets and text) and releases its sequence number; thread 0 polls that word over PCIe; the body answers exactly as the
// one-launch kernel does (ids, offsets, status, the sequence number released last).  The kernel ENDS BY ITSELF when no request has come for
// `idle_ticks` (100 MHz ticks; 200 us by default): an application's hipDeviceSynchronize waits at most that long, and the next small call
// simply launches it again.  Leaving, it writes its generation into the output block: a host that finds the kernel gone while its request
// is still unanswered launches the next generation, which starts from the last sequence number ANSWERED — every request is answered once.
__global__ __launch_bounds__(K_THREADS) void td_small_resident(const Tables* Tp, uint8_t* in, uint8_t* out, unsigned long long gen, unsigned long long idle_ticks) {
    __shared__ unsigned long long s_req;
    __shared__ int s_hdr[4];
    const SmallMailbox* const mb = reinterpret_cast<const SmallMailbox*>(in);
    SmallStatus* const status = reinterpret_cast<SmallStatus*>(out);
    unsigned long long served = 0, t_last = 0;
    if (threadIdx.x == 0) {
        served = __hip_atomic_load(&status->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        t_last = wall_clock64();
    }
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned long long r;
            for (;;) {
                r = __hip_atomic_load(&mb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (r != served && r != 0ull) break;
                if (wall_clock64() - t_last > idle_ticks) { r = ~0ull; break; }
            }
            if (r != ~0ull) {
                s_hdr[0] = __hip_atomic_load(&mb->n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_hdr[1] = __hip_atomic_load(&mb->n_docs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_hdr[2] = __hip_atomic_load(&mb->use_fastpath, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_hdr[3] = __hip_atomic_load(&mb->offs_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            s_req = r;
        }
        __syncthreads();
        const unsigned long long req = s_req;
        if (req == ~0ull) break;
        SmallArgs a;
        a.Tp = Tp;
        a.doc_offsets = reinterpret_cast<const int64_t*>(in + 64);
        a.text = in + 64 + s_hdr[3];
        a.status = status;
        a.out_offsets = reinterpret_cast<int64_t*>(out + 64);
        a.out_tokens = reinterpret_cast<int32_t*>(out + 64 + s_hdr[3]);
        a.seq = req;
        a.n = s_hdr[0];
        a.n_docs = s_hdr[1];
        a.use_fastpath = s_hdr[2];
        __syncthreads();
        small_encode_body(a);
        __syncthreads();
        if (threadIdx.x == 0) { served = req; t_last = wall_clock64(); }
    }
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + 40), gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ td_small_decode ---------
// decode_bytes (CoreBPE::decode_bytes, tiktoken.cpp:236-255) for at most 1024 ids in ONE launch: the ids come straight out of
// a pinned host buffer, the bytes go to one through an LDS window (16-byte stores), a sequence number the host spins on is
// released last.  (The general path is three launches and four copies: ~110 us for a chat message's worth of ids.)
__global__ __launch_bounds__(K_THREADS) void td_small_decode(const SmallDecArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_out[SMALL_DEC_MAX_BYTES];
    __shared__ uint32_t s_wave[8];
    __shared__ uint32_t s_err, s_errpos;
    const Tables T = uniform_tables(a.Tp);
    const int tid = threadIdx.x;
    constexpr int PER = SMALL_DEC_MAX_TOKENS / K_THREADS;  // ids per lane (consecutive: the scan is over lanes)
    if (tid == 0) { s_err = 0; s_errpos = 0xFFFFFFFFu; }
    __syncthreads();
    int32_t ids[PER];
    uint32_t so[PER], ln[PER];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid * PER + k;
        ids[k] = i < a.n ? a.tokens[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid * PER + k;
        so[k] = 0; ln[k] = 0;
        if (i < a.n) {
            if (ids[k] >= 0 && ids[k] <= T.max_id) {
                so[k] = T.tok_off[ids[k]];
                ln[k] = T.tok_off[ids[k] + 1] - so[k];
            }
            if (ln[k] == 0) {  // (no token is empty)  The reference throws on the FIRST invalid id (tiktoken.cpp:249): the lowest index
                s_err = (uint32_t)TD_E_BAD_TOKEN;
                atomicMin(&s_errpos, (uint32_t)i);
            }
        }
        mine += ln[k];
    }
    uint32_t total;
    uint32_t off = block_excl_scan(mine, s_wave, total);
    const bool fits = total <= (uint32_t)SMALL_DEC_MAX_BYTES;
    if (fits && !s_err) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint8_t* src = T.tok_bytes + so[k];
            for (uint32_t q = 0; q < ln[k]; ++q) s_out[off + q] = src[q];
            off += ln[k];
        }
    }
    __syncthreads();
    if (fits && !s_err)
        for (uint32_t v = tid; v * 16u < total; v += K_THREADS) reinterpret_cast<uint4*>(a.out)[v] = reinterpret_cast<const uint4*>(s_out)[v];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        a.status->n_tokens = total;
        a.status->err = (int)s_err;
        a.status->err_pos = (long long)s_errpos;
        a.status->fallback = fits ? 0 : 1;
        __hip_atomic_store(&a.status->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_small_decode(const SmallDecArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(td_small_decode, dim3(1), dim3(K_THREADS), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_small_encode(const SmallArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(td_small_encode, dim3(1), dim3(K_THREADS), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_small_resident(const Tables* Tp, void* in, void* out, unsigned long long gen, unsigned long long idle_ticks, hipStream_t stream) {
    hipLaunchKernelGGL(td_small_resident, dim3(1), dim3(K_THREADS), 0, stream, Tp, (uint8_t*)in, (uint8_t*)out, gen, idle_ticks);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launches ----------------
static int g_blocks_split = 0, g_blocks_encode = 0, g_blocks_merge = 0, g_blocks_long = 0;
static int resident_blocks(const void* fn, int fallback_per_cu, int threads = K_THREADS, int max_per_cu = 0) {
    // persistent grid = exactly the workgroups that are resident at once (a larger grid would run in
    // uneven rounds: tiles are dealt round-robin to blockIdx)
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) == hipSuccess && per_cu > 0) {
        if (getenv("TD_DEBUG_GRID")) fprintf(stderr, "[tokendagger] persistent grid: %d CUs x %d workgroups\n", prop.multiProcessorCount, per_cu);
        return prop.multiProcessorCount * (max_per_cu > 0 && per_cu > max_per_cu ? max_per_cu : per_cu);
    }
    return 256 * fallback_per_cu;
}
int encode_grid_blocks() {
    if (!g_blocks_encode) g_blocks_encode = resident_blocks((const void*)td_probe_tiles, 3);
    const char* e = getenv("TD_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_encode;
}
int merge_grid_blocks() {
    // (four workgroups per CU, measured: the occupancy query says five fit — 5 x 32 KB is all of the LDS — but with a grid of five
    // per CU the kernel is 10-20 % slower, as it is with four workgroups of five wavefronts: the fifth does not run beside the others)
    if (!g_blocks_merge) g_blocks_merge = resident_blocks((const void*)td_merge_pieces, 4, MG_THREADS, 4);
    const char* e = getenv("TD_MERGE_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_merge;
}
static int long_grid_blocks() {  // (work is dealt round-robin to the wavefronts: a grid larger than what is resident runs in uneven rounds)
    if (!g_blocks_long) g_blocks_long = resident_blocks((const void*)td_long_pieces, 3);
    const char* e = getenv("TD_LONG_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_long;
}
static int giant_grid_blocks() {
    // (a piece above gp_coop_min bytes is swept by ALL workgroups with grid barriers in between: the grid has to be resident at once —
    // one 1024-thread workgroup per CU at most, whatever the occupancy query says fits — on half of the CUs, so that two such launches
    // (two handles of a process, two processes) do not starve each other; a third finds out at its first barrier: gp_grid_meet)
    static int blocks = 0;
    if (!blocks) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        blocks = 1;  // (no grid barrier without knowing what is resident)
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)td_giant_pieces, GP_THREADS, 0) == hipSuccess && per_cu > 0)
            blocks = (prop.multiProcessorCount < GP_MAX_BLOCKS ? prop.multiProcessorCount : GP_MAX_BLOCKS) / 2;  // (two launches side by side fit; 128 workgroups: 29 ms for the megabyte, 256: 28)
        if (blocks < 1) blocks = 1;
        const char* e = getenv("TD_GIANT_BLOCKS");
        if (e && atoi(e) > 0 && atoi(e) <= blocks) blocks = atoi(e);
    }
    return blocks;
}
static int g_blocks_fused = 0;
int fused_grid_blocks() {
    if (!g_blocks_fused) g_blocks_fused = resident_blocks((const void*)td_split_tiles<0u, true, false>, 3);
    const char* e = getenv("TD_FUSED_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_fused;
}
static int g_blocks_direct = 0;
int direct_grid_blocks() {  // the fused loop with direct placement (an instantiation of its own: other registers, more LDS)
    if (!g_blocks_direct) g_blocks_direct = resident_blocks((const void*)td_split_tiles<0u, true, true>, 3);
    const char* e = getenv("TD_FUSED_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_direct;
}
static int split_grid_blocks() {
    if (!g_blocks_split) g_blocks_split = resident_blocks((const void*)td_split_tiles<0u, false, false>, 3);
    const char* e = getenv("TD_SPLIT_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return g_blocks_split;
}

static int tail_grid_blocks() {  // td_tail: what is resident (its merge rows are dealt to the wavefronts; its barriers need every workgroup on a CU)
    static int blocks = 0;
    if (!blocks) blocks = resident_blocks((const void*)td_tail, 2, K_THREADS, 2);
    const char* e = getenv("TD_TAIL_BLOCKS_PER_CU");
    if (e && atoi(e) > 0) return 256 * atoi(e);
    return blocks;
}
static int far_probe_grid_blocks() {
    const char* e = getenv("TD_FAR_PROBE_BLOCKS_PER_CU");  // (tests: a grid that cannot be resident at once — workgroup 0 walks the phases alone)
    if (e && atoi(e) > 0) return 256 * atoi(e);
    static int blocks = 0;
    if (!blocks) blocks = resident_blocks((const void*)td_far_probe, 2, K_THREADS, 2);
    return blocks;
}

hipError_t launch_encode(const EncodeArgs& a, hipStream_t stream, hipEvent_t* ev, const LaunchAux* aux) {
    if (a.n_tiles <= 0) return hipSuccess;
    if (ev) (void)hipEventRecord(ev[0], stream);
    static const bool prepare_split = getenv("TD_PREPARE_SPLIT") && atoi(getenv("TD_PREPARE_SPLIT")) > 0;  // (A/B: rounds 1-5's two launches)
    if (!prepare_split) {
        // one launch: the per-call state cleared, the document bitmap and the tiles' first documents written by text range
        const int64_t nranges = (((a.n + 31) / 32 + 2) * 32 + PM_RANGE - 1) / PM_RANGE;
        int64_t pb = (nranges + PM_WAVES - 1) / PM_WAVES, zb = ((a.dedupe ? ((int64_t)a.dd_mask + 1) / 2 : 0) + a.n_tiles + 255) / 256;  // (the table of distinct pieces: 16 bytes a thread)
        if (zb > 4096) zb = 4096;
        if (pb < zb) pb = zb;
        if (pb > (1 << 20)) pb = 1 << 20;
        if (pb < 1) pb = 1;
        hipLaunchKernelGGL(td_prepare_mark, dim3((int)pb), dim3(64 * PM_WAVES), 0, stream, a);
    } else {
        {   // one launch clears what was seven memsets: document bits, per-tile long-piece counts, first-document
            // indices (0xFFFFFFFF = none) and the per-call counters
            const int64_t words = (a.n + 31) / 32 + 2;
            int pb = (int)((words / 4 + 255) / 256);
            if (pb > 2048) pb = 2048;
            if (pb < 1) pb = 1;
            hipLaunchKernelGGL(td_prepare, dim3(pb), dim3(256), 0, stream, a);
        }
        {
            const int64_t nd = a.n_docs;
            int blocks = (int)((nd + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(td_mark_docs, dim3(blocks), dim3(256), 0, stream, a.doc_offsets, nd, a.n, a.docbits, a.tile_first_doc);
        }
    }
    if (a.sp.n) {  // allowed special tokens: their two ends become ends of subject
        const hipError_t se = launch_special_cuts(a, stream);
        if (se != hipSuccess) return se;
    }
    const int sblocks = a.n_stiles < split_grid_blocks() ? a.n_stiles : split_grid_blocks();
    const int pblocks = a.n_tiles < encode_grid_blocks() ? a.n_tiles : encode_grid_blocks();
    if (ev) (void)hipEventRecord(ev[1], stream);
    constexpr uint32_t PV_TEKKEN = PV_NO_CONTRACTION | PV_SINGLE_DIGIT;
    constexpr uint32_t PV_CL100K = PV_NO_CONTRACTION | PV_LEADING_CONTRACTION | PV_PLAIN_LETTERS;
    bool fused = false;
    if (a.pat_flags & PV_GENERIC) {  // not a member of the family: the compiled pattern, document by document (td_generic.hip)
        const hipError_t ge = launch_generic_split(a, stream);
        if (ge != hipSuccess) return ge;
    } else {
        fused = a.fused != 0;
#ifdef TD_ABLATE
        if (a.stop_after) fused = false;  // (the phase numbers are the unfused kernels')
#endif
        const int fblocks = a.n_stiles < fused_grid_blocks() ? a.n_stiles : fused_grid_blocks();
        const int dblocks = a.direct ? (a.n_stiles < direct_grid_blocks() ? a.n_stiles : direct_grid_blocks()) : 0;
#define TD_LAUNCH_SPLIT(PVX)                                                                                          \
    if (fused && a.direct) hipLaunchKernelGGL((td_split_tiles<(PVX), true, true>), dim3(dblocks), dim3(K_THREADS), 0, stream, a); \
    else if (fused) hipLaunchKernelGGL((td_split_tiles<(PVX), true, false>), dim3(fblocks), dim3(K_THREADS), 0, stream, a);      \
    else hipLaunchKernelGGL((td_split_tiles<(PVX), false, false>), dim3(sblocks), dim3(K_THREADS), 0, stream, a)
        switch (a.pat_flags) {
            case PV_GPT2: TD_LAUNCH_SPLIT(PV_GPT2); break;
            case 0u: TD_LAUNCH_SPLIT(0u); break;
            case PV_TEKKEN: TD_LAUNCH_SPLIT(PV_TEKKEN); break;
            case PV_CL100K: TD_LAUNCH_SPLIT(PV_CL100K); break;
            case PV_CL100K | PV_WS_EOS_FIRST: TD_LAUNCH_SPLIT(PV_CL100K | PV_WS_EOS_FIRST); break;
            case PV_CL100K | PV_SINGLE_DIGIT: TD_LAUNCH_SPLIT(PV_CL100K | PV_SINGLE_DIGIT); break;
            default: return hipErrorInvalidValue;
        }
#undef TD_LAUNCH_SPLIT
    }
#ifdef TD_ABLATE
    const bool tokens = a.stop_after != 2 && a.stop_after != 11 && a.stop_after != 12;
    const bool merges = a.stop_after != 3 && a.stop_after != 30 && a.stop_after != 31 && a.stop_after != 32;
#else
    const bool tokens = true, merges = true;
#endif
    // The SPARSE sequence (a.sparse: the handle's last counters say plain text — td_api.cpp): everything between the tile loop and the
    // packing is two launches, td_tail and td_giant_scan (see "several phases in one launch").  Six launches a step instead of fifteen.
    const bool sparse = a.sparse && fused && tokens && merges && !a.sp.n && !a.direct;
    if (sparse) {
        if (ev) { (void)hipEventRecord(ev[2], stream); (void)hipEventRecord(ev[3], stream); }
        const int tb = tail_grid_blocks();
        hipLaunchKernelGGL(td_tail, dim3(tb), dim3(K_THREADS), 0, stream, a);
        if (ev) (void)hipEventRecord(ev[4], stream);
        const int nchunks = (a.n_tiles + K_SCAN_CHUNK - 1) / K_SCAN_CHUNK;
        int gb = giant_grid_blocks();
        if (gb < nchunks) gb = nchunks < GP_MAX_BLOCKS ? nchunks : GP_MAX_BLOCKS;
        hipLaunchKernelGGL(td_giant_scan, dim3(gb), dim3(GP_THREADS), 0, stream, a);
    } else {
    // the DENSE sequence: the kernels of their own, at their own occupancies
    if (fused && tokens) {
        // far pieces, chains of flagged tiles, the token tiles the fused loop deferred (normally none of the three): one launch
        hipLaunchKernelGGL(td_far_probe, dim3(far_probe_grid_blocks()), dim3(K_THREADS), 0, stream, a);
        if (ev) { (void)hipEventRecord(ev[2], stream); (void)hipEventRecord(ev[3], stream); }
    } else {
        hipLaunchKernelGGL(td_split_far_pieces, dim3(64), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(td_split_far_tiles, dim3(256), dim3(256), 0, stream, a);
        if (ev) (void)hipEventRecord(ev[2], stream);
        if (tokens) hipLaunchKernelGGL(td_probe_tiles, dim3(pblocks), dim3(K_THREADS), 0, stream, a);
        if (ev) (void)hipEventRecord(ev[3], stream);
    }
    // the long pieces beside the chain of the short ones (see LaunchAux); with per-segment events (ev) everything stays in line
    // Which of the two runs on the caller's stream: the LONG pieces (few workgroups, the longest kernel of the branch) — they are
    // dispatched the moment td_far_probe ends, and the short pieces' chain, which arrives over the second queue a few microseconds
    // later, fills the chip around them.  The other way round (rounds 3-5) the short chain's 65 000 wavefronts held the slots when
    // the long pieces' workgroups arrived, and as plain launches only half of the long pieces' time was hidden (mixed-script text,
    // 256 MiB: 2.86 ms a step against 2.60 as a graph replay).  TD_FORK_LONG_FIRST=0 = the old assignment.
    static const bool long_first = !(getenv("TD_FORK_LONG_FIRST") && atoi(getenv("TD_FORK_LONG_FIRST")) == 0);
    const bool fork = aux && !ev && tokens;
    hipStream_t s_short = stream, s_long = stream;
    if (fork) {
        if (long_first && merges) s_short = aux->s; else s_long = aux->s;
        hipError_t fe = hipEventRecord(aux->fork, stream);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(aux->s, aux->fork, 0);
        if (fe != hipSuccess) return fe;
        hipLaunchKernelGGL(td_long_pieces, dim3(long_grid_blocks()), dim3(256), 0, s_long, a);
        hipLaunchKernelGGL(td_giant_pieces, dim3(giant_grid_blocks()), dim3(GP_THREADS), 0, s_long, a);
        if (s_long == aux->s && (fe = hipEventRecord(aux->join, aux->s)) != hipSuccess) return fe;
    }
    if (tokens && merges) {
        const int wtiles = (a.n_tiles + K_THREADS / 64 - 1) / (K_THREADS / 64);
        const int mblocks = wtiles < merge_grid_blocks() ? wtiles : merge_grid_blocks();
        // (a wavefront per tile up to 16 384 workgroups: with the 2048 of the first form — a third more than are resident at 6 wavefronts per
        // SIMD — the last third ran alone: collect + merge + copy 0.66 -> 0.58 ms per 256 MiB of the code file set, 0.55 -> 0.50 on mixed-script text)
        static const int collect_max = getenv("TD_COLLECT_BLOCKS") ? atoi(getenv("TD_COLLECT_BLOCKS")) : 16384;
        hipLaunchKernelGGL(td_collect_misses, dim3(wtiles < collect_max ? wtiles : collect_max), dim3(K_THREADS), 0, s_short, a);
        hipLaunchKernelGGL(td_merge_pieces, dim3(mblocks), dim3(MG_THREADS), 0, s_short, a);
        static const int copy_max = getenv("TD_COPY_BLOCKS") ? atoi(getenv("TD_COPY_BLOCKS")) : 256 * 4;
        if (a.dedupe) hipLaunchKernelGGL(td_copy_dups, dim3(wtiles < copy_max ? wtiles : copy_max), dim3(K_THREADS), 0, s_short, a);
        if (fork && s_short == aux->s) {
            const hipError_t fe = hipEventRecord(aux->join, aux->s);
            if (fe != hipSuccess) return fe;
        }
    }
    if (ev) (void)hipEventRecord(ev[4], stream);
    if (tokens) {
        if (fork) {
            const hipError_t je = hipStreamWaitEvent(stream, aux->join, 0);
            if (je != hipSuccess) return je;
        } else {
            hipLaunchKernelGGL(td_long_pieces, dim3(long_grid_blocks()), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(td_giant_pieces, dim3(giant_grid_blocks()), dim3(GP_THREADS), 0, stream, a);
        }
        if (a.pat_flags & PV_GENERIC) {  // text the pattern skips gets no tokens
            const hipError_t ge = launch_generic_gaps(a, stream);
            if (ge != hipSuccess) return ge;
        }
        if (a.sp.n) {  // the bytes of a special token that was cut out get its id
            const hipError_t se = launch_special_ids(a, stream);
            if (se != hipSuccess) return se;
        }
        hipLaunchKernelGGL(td_scan_tiles, dim3((a.n_tiles + K_SCAN_CHUNK - 1) / K_SCAN_CHUNK), dim3(1024), 0, stream, a);
    }
    }
    if (tokens) {
        if (ev) (void)hipEventRecord(ev[5], stream);
        if (a.pack_split) {
            const int pwg = (a.n_tiles + K_THREADS / 64 - 1) / (K_THREADS / 64);
            hipLaunchKernelGGL(td_pack_plain, dim3(pwg < (1 << 20) ? pwg : (1 << 20)), dim3(K_THREADS), 0, stream, a);
            if (a.pack_dense) {
                hipLaunchKernelGGL(td_pack_dense<false>, dim3(pwg), dim3(K_THREADS), 0, stream, a);
                hipLaunchKernelGGL(td_pack_dense<true>, dim3(pwg), dim3(K_THREADS), 0, stream, a);
            }
            // (a grid of 16 384 workgroups, eight times the resident ones: on mixed-script text 0.45 -> 0.36 ms per 256 MiB against
            // the 2048 of rounds 2-4 — a wavefront that walks fewer tiles ends its pipeline sooner — and the 47 us it takes to find
            // nothing to do on English are the walk over the tiles' count words)
            static const int rest_max = getenv("TD_PACK_REST_BLOCKS") ? atoi(getenv("TD_PACK_REST_BLOCKS")) : 16384;
            int rwg = pwg < rest_max ? pwg : rest_max;
            if (sparse || a.pack_dense) {  // (a lane per word of the scan's masks: 1024 tiles a wavefront and round trip)
                rwg = (a.n_tiles + 4095) / 4096;
                if (rwg > 2048) rwg = 2048;
            }
            hipLaunchKernelGGL(td_pack_rest, dim3(rwg), dim3(K_THREADS), 0, stream, a);
        } else {
            hipLaunchKernelGGL(td_pack_tokens, dim3(256 * 8), dim3(K_THREADS), 0, stream, a);
        }
    } else if (ev) {
        (void)hipEventRecord(ev[5], stream);
    }
    if (ev) (void)hipEventRecord(ev[6], stream);
    return hipGetLastError();
}

// ------------------------------------------------------------------ decode ------------------
// ids -> bytes (CoreBPE::decode_bytes, tiktoken.cpp:236-255).  td_decode_len: byte length of every token (rank ->
// bytes store), exclusive scan inside chunks of 4096 tokens; td_decode_chunks: exclusive scan of the chunk totals.
// td_decode_copy: one lane per token copies its bytes to chunk base + local
// offset.  An id outside the vocabulary raises TD_E_BAD_TOKEN with its index (the reference throws "Invalid token for
// decoding", tiktoken.cpp:249).
constexpr int K_DEC_CHUNK = 4096;
__global__ __launch_bounds__(1024) void td_decode_len(const DecodeArgs a) {
    __shared__ unsigned long long s_wsum[16];
    const Tables T = uniform_tables(a.Tp);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * K_DEC_CHUNK + tid * 4;
    uint32_t v[4] = {0, 0, 0, 0};
    int32_t ids[4] = {-1, -1, -1, -1};
    if (e0 + 4 <= a.n && (((uintptr_t)a.tokens) & 15) == 0) {
        const int4 q = *reinterpret_cast<const int4*>(a.tokens + e0);
        ids[0] = q.x; ids[1] = q.y; ids[2] = q.z; ids[3] = q.w;
    } else {
        for (int k = 0; k < 4; ++k)
            if (e0 + k < a.n) ids[k] = a.tokens[e0 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = e0 + k;
        if (i >= a.n) break;
        const int32_t id = ids[k];
        uint32_t len = 0;
        if (id >= 0 && id <= T.max_id) len = T.tok_off[id + 1] - T.tok_off[id];
        if (len == 0) {  // (no token is empty)  The reference throws on the FIRST invalid id (tiktoken.cpp:249): the lowest index wins,
            // kept as the MAXIMUM of (INT64_MAX - index) so that the zeroed control block needs no other initial value; the
            // host turns it back (device_status_locked).  Only while the error on record is this one.
            const int was = atomicCAS(a.err, 0, TD_E_BAD_TOKEN);
            if (was == 0 || was == TD_E_BAD_TOKEN)
                atomicMax(reinterpret_cast<unsigned long long*>(a.err_pos), (unsigned long long)(0x7FFFFFFFFFFFFFFFll - i));
        }
        v[k] = len;
    }
    const unsigned long long mine = (unsigned long long)v[0] + v[1] + v[2] + v[3];
    unsigned long long x = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(x, d);
        if (lane >= d) x += t;
    }
    if (lane == 63) s_wsum[wv] = x;
    __syncthreads();
    unsigned long long woff = 0;
    for (int w = 0; w < wv; ++w) woff += s_wsum[w];
    unsigned long long run = woff + x - mine;
    uint32_t lo4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo4[k] = (uint32_t)run;
        run += v[k];
    }
    if (e0 + 4 <= a.n) *reinterpret_cast<uint4*>(a.local_off + e0) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);  // (16-byte aligned: e0 % 4 == 0)
    else
        for (int k = 0; k < 4; ++k)
            if (e0 + k < a.n) a.local_off[e0 + k] = lo4[k];
    if (tid == 1023) a.chunk_pref[blockIdx.x] = (int64_t)run;  // chunk total; td_decode_chunks turns it into a prefix
}
// exclusive scan of the chunk totals (13 k chunks for 55 M ids: a launch of its own — a "last workgroup" scheme would
// serialise one atomic per chunk on a single address)
__global__ __launch_bounds__(1024) void td_decode_chunks(const DecodeArgs a) {
    __shared__ unsigned long long s_wsum[16];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t nchunks = (a.n + K_DEC_CHUNK - 1) / K_DEC_CHUNK;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < nchunks; c0 += 1024) {
        const int64_t c = c0 + tid;
        const unsigned long long tot = (c < nchunks) ? (unsigned long long)a.chunk_pref[c] : 0ull;
        unsigned long long y = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(y, d);
            if (lane >= d) y += t;
        }
        if (lane == 63) s_wsum[wv] = y;
        __syncthreads();
        unsigned long long woff = s_carry;
        for (int w = 0; w < wv; ++w) woff += s_wsum[w];
        if (c < nchunks) a.chunk_pref[c] = (int64_t)(woff + y - tot);
        __syncthreads();
        if (tid == 1023) s_carry = woff + y;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long total = s_carry;
        a.chunk_pref[nchunks] = (int64_t)total;
        if (a.n_bytes) *a.n_bytes = (int64_t)total;
        if ((int64_t)total > a.out_cap && atomicCAS(a.err, 0, TD_E_CAPACITY) == 0) *a.err_pos = (long long)total;
    }
}
// byte offset of every document start (decode_batch): token index -> chunk base + offset inside the chunk
__global__ void td_decode_doc_offsets(const DecodeArgs a) {
    const int64_t nchunks = (a.n + K_DEC_CHUNK - 1) / K_DEC_CHUNK;
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d <= a.n_docs; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = a.doc_tok_offsets[d];
        a.doc_byte_offsets[d] = (i >= a.n) ? a.chunk_pref[nchunks] : a.chunk_pref[i / K_DEC_CHUNK] + a.local_off[i];
    }
}

// One workgroup per chunk of 4096 tokens; the chunk's bytes are contiguous in the output.  Lanes copy their tokens'
// bytes (byte loads from the 1.4 MB rank -> bytes store, L2-resident) into an LDS window, the workgroup streams the
// window out with 16-byte stores; chunks whose bytes exceed the window take several passes.
constexpr int K_DEC_WIN = 32768;
__global__ __launch_bounds__(1024) void td_decode_copy(const DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_out[K_DEC_WIN];
    const Tables T = uniform_tables(a.Tp);
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    const int64_t cbase = a.chunk_pref[c], cend = a.chunk_pref[c + 1];
    if (cend > a.out_cap) return;  // capacity error already raised by td_decode_len
    const int64_t e0 = c * K_DEC_CHUNK + tid * 4;
    uint32_t so[4], lo[4], ln[4];
    int32_t ids[4] = {-1, -1, -1, -1};
    if (e0 + 4 <= a.n && (((uintptr_t)a.tokens) & 15) == 0) {
        const int4 q = *reinterpret_cast<const int4*>(a.tokens + e0);
        const uint4 l = *reinterpret_cast<const uint4*>(a.local_off + e0);
        ids[0] = q.x; ids[1] = q.y; ids[2] = q.z; ids[3] = q.w;
        lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; lo[3] = l.w;
    } else {
        for (int k = 0; k < 4; ++k) {
            lo[k] = 0;
            if (e0 + k < a.n) { ids[k] = a.tokens[e0 + k]; lo[k] = a.local_off[e0 + k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        so[k] = 0; ln[k] = 0;
        if (ids[k] >= 0 && ids[k] <= T.max_id) {
            so[k] = T.tok_off[ids[k]];
            ln[k] = T.tok_off[ids[k] + 1] - so[k];
        }
    }
    const uint32_t cbytes = (uint32_t)(cend - cbase);
    constexpr uint32_t WIN = K_DEC_WIN - 16;  // the window is shifted so that LDS and global addresses agree mod 16
    for (uint32_t w0 = 0; w0 < cbytes; w0 += WIN) {
        const uint32_t wn = (cbytes - w0 < WIN) ? cbytes - w0 : WIN;
        uint8_t* dst = a.out + cbase + w0;
        const uint32_t shift = (uint32_t)(uintptr_t)dst & 15u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the part of token k inside [w0, w0 + wn)
            const uint32_t b = lo[k] > w0 ? lo[k] : w0;
            const uint32_t e = (lo[k] + ln[k] < w0 + wn) ? lo[k] + ln[k] : w0 + wn;
            const uint8_t* src = T.tok_bytes + so[k];
            for (uint32_t q = b; q < e; ++q) s_out[q - w0 + shift] = src[q - lo[k]];
        }
        __syncthreads();
        uint32_t head = (16u - shift) & 15u;
        if (head > wn) head = wn;
        if ((uint32_t)tid < head) dst[tid] = s_out[shift + tid];
        const uint32_t nv = (wn - head) >> 4;
        for (uint32_t v = tid; v < nv; v += 1024)
            *reinterpret_cast<uint4*>(dst + head + 16 * v) = *reinterpret_cast<const uint4*>(s_out + shift + head + 16 * v);
        const uint32_t done = head + 16 * nv;
        if (done + (uint32_t)tid < wn) dst[done + tid] = s_out[shift + done + tid];
        __syncthreads();
    }
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream, int phases) {
    if (a.n <= 0) return hipSuccess;
    const int64_t nchunks = (a.n + K_DEC_CHUNK - 1) / K_DEC_CHUNK;
    if (nchunks > 0x7FFFFFF0ll) return hipErrorInvalidValue;
    if (phases & 1) {
        hipLaunchKernelGGL(td_decode_len, dim3((unsigned)nchunks), dim3(1024), 0, stream, a);
        hipLaunchKernelGGL(td_decode_chunks, dim3(1), dim3(1024), 0, stream, a);
        if (a.doc_byte_offsets && a.doc_tok_offsets) {
            int blocks = (int)((a.n_docs + 256) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(td_decode_doc_offsets, dim3(blocks), dim3(256), 0, stream, a);
        }
    }
    if (phases & 2) hipLaunchKernelGGL(td_decode_copy, dim3((unsigned)nchunks), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

}  // namespace td
