// Generic split patterns on the device: what td_regex.cpp compiled is matched by the backtracking matcher of
// td_regex.h, ONE LANE PER CHUNK of the text (64 B .. 1 KiB, gx_chunk_for), speculatively inside a document and checked afterwards (below; round 2
// ran one lane per document: as fast as its longest document was long).  The kernels write the same START bitmap the
// family's td_split_tiles writes, so td_probe_tiles and everything behind it run unchanged.
//
// Text the pattern SKIPS (the reference tokenizes only what matches) becomes a piece of its own in the bitmap and is noted in
// a second bitmap; after the merge kernels td_generic_gaps turns such a piece's slot into a marker with zero ids and takes
// the ids it had been given out of the tile's count, so td_pack_tokens (which already expands markers of any size) leaves it
// out.  The members of the family keep their own kernels.
#include <hip/hip_runtime.h>

#include "td_kernels.h"
#include "td_regex.h"

namespace td {

namespace {
struct GlobalDoc {  // the subject: one document of the batch
    const uint8_t* p;
    __device__ __forceinline__ uint32_t byte(int64_t i) const { return p[i]; }
};
// The same with the 16 bytes around the last position read kept in registers: the matcher walks its subject a character at
// a time, and a byte load per character was a cache round trip per character in a dependent chain (64 different cache
// lines per wavefront load).  `text` is the whole batch (16-byte aligned), o0 the document's offset in it, n_text its size.
struct WindowDoc {
    const uint8_t* text;
    int64_t o0, n_text;
    mutable int64_t blk;
    mutable uint32_t w0, w1, w2, w3;
    // ... and the 16 bytes BEHIND them requested as soon as a block is entered: the lanes of a wavefront cross their block
    // boundaries at different times, so with a load at the crossing nearly every piece some lane made the whole wavefront
    // wait a global-memory round trip (a lane needed 3 microseconds per byte; the matcher's arithmetic is a tenth of that).
    mutable uint32_t x0, x1, x2, x3;
    __device__ __forceinline__ void load_block(int64_t b, uint32_t& q0, uint32_t& q1, uint32_t& q2, uint32_t& q3) const {
        if (16 * b + 16 <= n_text) {
            const uint4 v = *reinterpret_cast<const uint4*>(text + 16 * b);
            q0 = v.x; q1 = v.y; q2 = v.z; q3 = v.w;
        } else {  // (the last, partial block of the text, or past it)
            uint32_t t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16 && 16 * b + k < n_text; ++k) t[k >> 2] |= (uint32_t)text[16 * b + k] << (8 * (k & 3));
            q0 = t[0]; q1 = t[1]; q2 = t[2]; q3 = t[3];
        }
    }
    __device__ __forceinline__ uint32_t byte(int64_t i) const {
        const int64_t g = o0 + i, b = g >> 4;
        if (b != blk) {
            if (b == blk + 1) { w0 = x0; w1 = x1; w2 = x2; w3 = x3; }  // (requested when block b - 1 was entered)
            else load_block(b, w0, w1, w2, w3);
            blk = b;
            load_block(b + 1, x0, x1, x2, x3);  // (not waited for here: first looked at when the matcher gets there)
        }
        const uint32_t k = (uint32_t)g & 15u;
        const uint32_t lo = (k & 4u) ? w1 : w0, hi = (k & 4u) ? w3 : w2;
        return (((k & 8u) ? hi : lo) >> (8u * (k & 3u))) & 0xFFu;
    }
};
// the matcher's backtracking state of one lane in LDS: word k of lane t at [k * GX_THREADS + t] — every lane its own bank,
// whatever nodes the lanes are at
constexpr int GX_THREADS = 256;
constexpr int GX_STATE_WORDS = 2 * RX_MAX_SEQ + 1;
struct RxStateLds {
    uint32_t* w;  // &s_state[threadIdx.x]
    __device__ __forceinline__ int32_t get_e(int i) const { return (int32_t)w[i * GX_THREADS]; }
    __device__ __forceinline__ void set_e(int i, int32_t v) { w[i * GX_THREADS] = (uint32_t)v; }
    __device__ __forceinline__ uint32_t get_c(int i) const { return w[(RX_MAX_SEQ + 1 + i) * GX_THREADS]; }
    __device__ __forceinline__ void set_c(int i, uint32_t v) { w[(RX_MAX_SEQ + 1 + i) * GX_THREADS] = v; }
};
__device__ __forceinline__ void raise_g(const EncodeArgs& a, int code, int64_t pos) {
    if (atomicCAS(a.err, 0, code) == 0) *a.err_pos = pos;
}
}  // namespace

// ---- inside a document: speculative chunks ---------------------------------------------------------------------------
// The next piece from a position depends on the position and the subject only (rx_next_piece has no other state), so a
// document can be matched in CHUNKS in parallel: every lane starts at the first character boundary of its chunk as if
// a piece started there, marks the piece starts it finds and notes where it left the chunk (its EXIT: the first piece start
// at or behind the chunk end).  td_generic_commit then checks every chunk against its predecessor: the predecessor's exit
// is this chunk's true ENTRY; if the chunk's own run marked a piece start exactly there, everything it found from there on
// is what a sequential run finds (induction from the document start, which is a true start), and what it marked in front
// of it was speculation and is cleared.  A chunk that lies inside one long piece must have left with the same exit.
// Matchers of tokenizer patterns resynchronise within a piece or two, so nearly every chunk passes; the documents of the
// others are redone from their first failing chunk by one lane (td_generic_redo).  Documents that start inside a chunk
// start with a true piece start and need no check.

struct GxWords {  // the lane's words of the START / gap bitmaps, written in increasing order
    uint32_t* sb;
    uint32_t* gb;
    int64_t cw;       // word being filled
    uint32_t s, g;
    __device__ __forceinline__ void upto(int64_t w) {  // words below w are complete
        while (cw < w) { sb[cw] = s; gb[cw] = g; ++cw; s = 0; g = 0; }
    }
    __device__ __forceinline__ void mark(int64_t p, bool gap) {
        upto(p >> 5);
        s |= 1u << (p & 31);
        if (gap) g |= 1u << (p & 31);
    }
};

// piece starts of text[from, lim) inside document [o0, o1), matching from `from`; returns the first piece start >= lim (or o1)
template <class W>
__device__ __forceinline__ int64_t gx_run(const RxProgram& P, const RxTables& T, const uint8_t* text, int64_t n_text, bool aligned, int64_t o0,
                                          int64_t o1, int64_t from, int64_t lim, W& out) {
    const int64_t n = o1 - o0;
    int64_t pos = from - o0;
    if (!aligned) {  // (a text pointer that is not 16-byte aligned: plain byte loads)
        const GlobalDoc s{text + o0};
        while (pos < n) {
            if (o0 + pos >= lim) return o0 + pos;
            int64_t ms, me;
            rx_next_piece(P, T, s, pos, n, ms, me);
            if (ms > pos) {
                out.mark(o0 + pos, true);
                if (o0 + ms >= lim) return o0 + ms;
            }
            out.mark(o0 + ms, false);
            pos = me;
        }
        return o1;
    }
    const WindowDoc s{text, o0, n_text, -2, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    while (pos < n) {
        if (o0 + pos >= lim) return o0 + pos;
        int64_t ms, me;
        rx_next_piece(P, T, s, pos, n, ms, me);
        if (ms > pos) {  // skipped text: a piece of its own, without tokens
            out.mark(o0 + pos, true);
            if (o0 + ms >= lim) return o0 + ms;
        }
        out.mark(o0 + ms, false);
        pos = me;
    }
    return o1;
}

// One chunk [c0, c1): the documents that have bytes in it, one after the other, from document `lo` (the one that holds c0) on;
// returns the chunk's exit.  ONE loop for all of them: a lane that is done with a document goes on to the next one inside the
// same loop, so the lanes of a wavefront need not be in the same document to share the matcher's call site.  (With a loop over
// the documents around a loop over a document's pieces, the wavefront went through the k-th documents of its 64 chunks
// together and every lane waited for the longest of them: a chunk took as long as its documents' longest stretches added up,
// 2.3 times a chunk inside one document — 64 MiB of paragraphs ran at 9.7 GB/s, the same bytes as one document at 20.5.)
__device__ __forceinline__ void gx_set_doc(GlobalDoc& s, const uint8_t* text, int64_t o0) { s.p = text + o0; }
__device__ __forceinline__ void gx_set_doc(WindowDoc& s, const uint8_t*, int64_t o0) { s.o0 = o0; }
template <class Doc>
__device__ __forceinline__ int64_t gx_chunk(const EncodeArgs& a, const RxProgram& P, const RxTables& T, Doc& s, int64_t lo, int64_t c0, int64_t c1,
                                            GxWords& W, RxStateLds& st) {
    int64_t d = lo - 1, o0 = 0, o1 = 0, pos = 0;  // (pos: offset in the text; pos >= o1: the next document's turn)
    bool inside = false;
    for (;;) {
        if (pos >= o1) {
            if (inside && o1 >= c1) return o1;  // the document ended at or behind the chunk end: nothing of it starts further on
            ++d;
            if (d >= a.n_docs) return c1;
            const int64_t q0 = a.doc_offsets[d], q1 = a.doc_offsets[d + 1];
            if (q0 >= c1) return c1;
            if (q1 <= q0) continue;
            o0 = q0; o1 = q1; inside = true;
            gx_set_doc(s, a.text, o0);
            pos = o0;
            if (o0 < c0) {  // the chunk starts inside this document: speculate from its first character boundary
                pos = c0;
                for (int k = 0; k < 3 && pos < o1 && (a.text[pos] & 0xC0u) == 0x80u; ++k) ++pos;
            }
            if (a.gx_prefix) {  // left context only (the end of the special token this segment stands behind): matched from behind it
                const int64_t pre = a.gx_prefix[d];
                if (pre) {
                    if (o0 >= c0) W.mark(o0, true);  // (its bytes: a stretch without tokens, like text the pattern skips)
                    if (pos < o0 + pre) pos = o0 + pre;
                }
            }
            continue;
        }
        if (pos >= c1) return pos;
        int64_t ms, me;
        rx_next_piece(P, T, s, pos - o0, o1 - o0, ms, me, st);
        if (o0 + ms > pos) {  // skipped text: a piece of its own, without tokens
            W.mark(pos, true);
            if (o0 + ms >= c1) return o0 + ms;
        }
        W.mark(o0 + ms, false);
        pos = o0 + me;
    }
}

#ifndef TD_GX_WAVES
#define TD_GX_WAVES 2
#endif
__global__ __launch_bounds__(GX_THREADS, TD_GX_WAVES) void td_generic_chunks(const EncodeArgs a) {
    const RxTables T{a.rx_stage1, a.rx_stage2};
    __shared__ RxProgram sP;  // (the matcher reads a node, a class or a literal at every step: out of HBM each was a cache round trip)
    __shared__ uint32_t s_state[GX_STATE_WORDS * GX_THREADS];
    RxStateLds st{&s_state[threadIdx.x]};
    static_assert(sizeof(RxProgram) % 4 == 0, "copied as dwords");
    for (uint32_t w = threadIdx.x; w < sizeof(RxProgram) / 4; w += blockDim.x)
        reinterpret_cast<uint32_t*>(&sP)[w] = reinterpret_cast<const uint32_t*>(a.rx)[w];
    __syncthreads();
    const RxProgram& P = sP;
    const int64_t GX_CHUNK = a.gx_chunk, n_chunks = (a.n + GX_CHUNK - 1) / GX_CHUNK;
    for (int64_t ch = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ch < n_chunks; ch += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c0 = ch * GX_CHUNK, c1 = (c0 + GX_CHUNK < a.n) ? c0 + GX_CHUNK : a.n;
        // the document that holds c0: the last one with offset <= c0 (behind empty ones at the same offset)
        int64_t lo = 0, hi = a.n_docs;  // doc_offsets[lo] <= c0 < doc_offsets[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.doc_offsets[mid] <= c0) lo = mid; else hi = mid;
        }
        GxWords W{a.startbits, a.gapbits, c0 >> 5, 0u, 0u};
        int64_t exit_at;
        if (a.text_aligned) {
            WindowDoc s{a.text, 0, a.n, -2, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            exit_at = gx_chunk(a, P, T, s, lo, c0, c1, W, st);
        } else {  // (a text pointer that is not 16-byte aligned: plain byte loads)
            GlobalDoc s{a.text};
            exit_at = gx_chunk(a, P, T, s, lo, c0, c1, W, st);
        }
        W.upto((c1 + 31) >> 5);
        if (c1 == a.n) {  // the words behind the text (td_probe_tiles reads a few of them)
            const int64_t wend = ((a.n + 31) >> 5) + 8;
            for (int64_t w = (c1 + 31) >> 5; w < wend; ++w) { a.startbits[w] = 0; a.gapbits[w] = 0; }
        }
        a.gx_exit[ch] = exit_at;
    }
}

__device__ __forceinline__ void gx_clear_below(uint32_t* bits, int64_t c0, int64_t upto) {  // clear bits [c0, upto), c0 word-aligned
    for (int64_t w = c0 >> 5; w < (upto >> 5); ++w) bits[w] = 0;
    if (upto & 31) bits[upto >> 5] &= ~((1u << (upto & 31)) - 1u);
}

__global__ __launch_bounds__(256) void td_generic_commit(const EncodeArgs a) {
    const int64_t GX_CHUNK = a.gx_chunk, n_chunks = (a.n + GX_CHUNK - 1) / GX_CHUNK;
    for (int64_t ch = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ch < n_chunks; ch += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c0 = ch * GX_CHUNK, c1 = (c0 + GX_CHUNK < a.n) ? c0 + GX_CHUNK : a.n;
        bool valid = true;
        if (!((a.docbits[c0 >> 5] >> (c0 & 31)) & 1u)) {  // the chunk starts inside a document: its head [c0, h) was speculation
            int64_t h = c1;  // first document start inside the chunk
            for (int64_t w = c0 >> 5; w < ((c1 + 31) >> 5) && h == c1; ++w) {
                uint32_t m = a.docbits[w];
                if (w == (c1 >> 5) && (c1 & 31)) m &= (1u << (c1 & 31)) - 1u;
                if (m) h = w * 32 + (__ffs(m) - 1);
            }
            const int64_t t = a.gx_exit[ch - 1];  // the true entry: where the chunk in front left off
            if (t >= h) {  // no piece of that document starts in the head
                gx_clear_below(a.startbits, c0, h);
                gx_clear_below(a.gapbits, c0, h);
                valid = h < c1 || a.gx_exit[ch] == t;  // (inside one long piece: the exit carries over)
            } else {
                valid = (a.startbits[t >> 5] >> (t & 31)) & 1u;
                gx_clear_below(a.startbits, c0, t);
                gx_clear_below(a.gapbits, c0, t);
            }
        }
        a.gx_state[ch] = valid ? 1u : 0u;
        if (!valid) {
            const uint32_t at = atomicAdd(a.gap_count, 1u);
            if (at < a.gap_cap) reinterpret_cast<uint32_t*>(a.gap_list)[at] = (uint32_t)ch;
            else raise_g(a, TD_E_SCRATCH, c0);
        }
    }
}

// A document with chunks that failed the check is put right by ONE lane — the lane of its FIRST failed chunk (a
// later failed chunk's predecessor may have been "validated" against a failed chunk's speculative exit, so its entry is not
// known to be true and two lanes of one document would write the same words).  The lane goes chunk by chunk: a chunk is
// matched anew from its true entry; if it leaves the chunk where the chunk's first run had left it and the chunk behind was
// validated against that exit, everything up to the document's next failed chunk is what a sequential run finds (induction
// as in td_generic_commit) and the lane jumps there; otherwise the chunk behind is matched anew as well.  gx_exit is only read
// here (a chunk that holds a document boundary carries the NEXT document's exit).
__global__ __launch_bounds__(64) void td_generic_redo(const EncodeArgs a) {
    const RxTables T{a.rx_stage1, a.rx_stage2};
    const RxProgram& P = *a.rx;
    const uint32_t nbad = *a.gap_count < a.gap_cap ? *a.gap_count : a.gap_cap;
    const int64_t GX_CHUNK = a.gx_chunk;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nbad; j += gridDim.x * blockDim.x) {
        const int64_t ch = reinterpret_cast<const uint32_t*>(a.gap_list)[j];
        const int64_t c0 = ch * GX_CHUNK;
        int64_t lo = 0, hi = a.n_docs;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.doc_offsets[mid] <= c0) lo = mid; else hi = mid;
        }
        const int64_t o0 = a.doc_offsets[lo], o1 = a.doc_offsets[lo + 1];
        bool first = true;  // no chunk that starts inside this document in front of this one has failed
        for (int64_t k = ch - 1; k * GX_CHUNK > o0; --k)
            if (a.gx_state[k] == 0u) { first = false; break; }
        if (!first) continue;
        int64_t m = ch, t = a.gx_exit[ch - 1];  // chunk being matched anew, its true entry (a failed chunk starts inside a document: ch > 0)
        for (;;) {  // (chunk m starts inside the document: m0 < o1.  A chunk behind the last piece start, t >= o1, is cleared)
            const int64_t m0 = m * GX_CHUNK, m1 = m0 + GX_CHUNK;
            const int64_t lim = m1 < o1 ? m1 : o1;
            int64_t e = t;  // (t >= lim: the chunk lies inside one piece and has no bits)
            GxWords W{a.startbits, a.gapbits, m0 >> 5, 0u, 0u};
            if (t < lim) e = gx_run(P, T, a.text, a.n, false, o0, o1, t, lim, W);
            if (lim == m1) {
                W.upto(m1 >> 5);
            } else {  // the document ends inside this chunk: the word that holds o1 keeps what belongs to the next document
                const int64_t wlast = o1 >> 5;
                const uint32_t keep_s = (o1 & 31) ? a.startbits[wlast] & ~((1u << (o1 & 31)) - 1u) : 0u;
                const uint32_t keep_g = (o1 & 31) ? a.gapbits[wlast] & ~((1u << (o1 & 31)) - 1u) : 0u;
                W.upto(wlast);
                if (o1 & 31) { a.startbits[wlast] = W.s | keep_s; a.gapbits[wlast] = W.g | keep_g; }
                break;
            }
            if (m1 >= o1) break;
            if (a.gx_state[m + 1] != 0u && a.gx_exit[m] == e) {  // back in step with the first runs: on to the next failed chunk of the document
                int64_t k = m + 2;
                while (k * GX_CHUNK < o1 && a.gx_state[k] != 0u) ++k;
                if (k * GX_CHUNK >= o1) break;
                m = k;
                t = a.gx_exit[k - 1];
            } else {
                ++m;
                t = e;
            }
        }
    }
}

// one lane per word of the gap bitmap: the slot of a skipped stretch (= pieces of its tile in front of it) becomes
// TOK_MISS | position | 0 ids
__global__ __launch_bounds__(256) void td_generic_gaps(const EncodeArgs a) {
    const int64_t nw = (a.n + 31) >> 5;
    for (int64_t gw = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gw < nw; gw += (int64_t)gridDim.x * blockDim.x) {
        for (uint32_t m = a.gapbits[gw]; m; m &= m - 1u) {
            const int64_t p = gw * 32 + (__ffs(m) - 1);
            if (p >= a.n) break;
            const int64_t tile = p / K_TILE;
            const int64_t w0 = (tile * K_TILE) >> 5, w1 = p >> 5;
            uint32_t slot = 0;
            for (int64_t w = w0; w < w1; ++w) slot += (uint32_t)__popc(a.startbits[w]);
            slot += (uint32_t)__popc(a.startbits[w1] & ((1u << (p & 31)) - 1u));
            uint32_t* sp = a.stage + (size_t)tile * K_STAGE + slot;
            const uint32_t v = *sp;
            uint32_t had = 1;  // ids the piece was given
            if (v & TOK_LONGREF) had = a.long_list[v & 0x7FFFFFFFu].ntok;
            else if (v & TOK_MISS) had = v & 127u;
            *sp = TOK_MISS | ((uint32_t)(p - tile * K_TILE) << 7);
            atomicAdd(&a.tile_extra[tile], 0u - had);  // (the scan adds counts and extras modulo 2^32)
            atomicOr(&a.tile_count[tile], TILE_MISS_LISTED);
        }
    }
}

hipError_t launch_generic_split(const EncodeArgs& a, hipStream_t stream) {
    const int64_t GX_CHUNK = a.gx_chunk, n_chunks = (a.n + GX_CHUNK - 1) / GX_CHUNK;
    int64_t blocks = (n_chunks + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(td_generic_chunks, dim3((unsigned)blocks), dim3(GX_THREADS), 0, stream, a);
    hipLaunchKernelGGL(td_generic_commit, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(td_generic_redo, dim3(256), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_generic_gaps(const EncodeArgs& a, hipStream_t stream) {
    const int64_t nw = (a.n + 31) >> 5;
    int64_t blocks = (nw + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(td_generic_gaps, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace td
