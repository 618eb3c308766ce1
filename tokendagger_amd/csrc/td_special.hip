// Allowed special tokens on the device (reference: CoreBPE::encode's segmentation, tiktoken.cpp:130-154,187-231,
// with tiktoken's semantics — the reference's own loop has iterator-invalidation UB): scanning forward, the LONGEST allowed
// special literal that starts at the current position is cut out and replaced by its id, the search goes on behind it; the
// text between two cuts is tokenized as a subject of its own.
//
// The text is not moved: a cut is made by marking its two ends as DOCUMENT boundaries in the document bitmap (the scanners
// treat them as ends of subject, which is exactly what a separately encoded segment sees), running the usual kernels, and
// — once the pieces of the literal's own bytes have got their (meaningless) ids — turning the first of them into a marker
// that carries ONE id, the special's, and the others into markers with none.  td_pack_tokens expands markers as ever; the
// per-document token offsets come out for the caller's documents, which the extra boundaries do not touch.
//
//   td_special_scan    a lane per 32 bytes: positions whose first two bytes start an allowed literal (8 KB bitmap in LDS)
//                      go on a list of candidates
//   td_special_match   a lane per candidate: matched against the sorted literal table (binary search for the greatest literal
//                      <= the text, then up its chain of prefixes: the first one that is a prefix of the text is the longest)
//                      -> its literal, HIT bitmap
//   td_special_accept  greedy left-to-right: a hit inside an accepted literal is dropped.  Hits more than a literal's length
//                      apart cannot touch, so every hit without another one in the 44 bytes in front of it is accepted for sure
//                      and walks its cluster -> ACCEPTED bitmap
//   td_special_mark    document bits at both ends of every accepted literal (behind td_mark_docs)
//   td_special_ids     behind the merge kernels: the literal's pieces -> markers (1 id, then 0 ids)
#include <hip/hip_runtime.h>

#include "td_kernels.h"

namespace td {

namespace {

__device__ __forceinline__ void raise_s(const EncodeArgs& a, int code, int64_t pos) {
    if (atomicCAS(a.err, 0, code) == 0) *a.err_pos = pos;
}

// first document start in (p, p + span], or p + span
__device__ __forceinline__ int64_t sp_doc_limit(const EncodeArgs& a, int64_t p, int64_t span) {
    int64_t lim = p + span < a.n ? p + span : a.n;
    for (int64_t q = p + 1; q < lim;) {
        uint32_t m = a.docbits[q >> 5] >> (q & 31);
        if (m) { const int64_t f = q + (__ffs(m) - 1); return f < lim ? f : lim; }
        q = ((q >> 5) + 1) << 5;
    }
    return lim;
}

// index of the longest allowed literal that is a prefix of text[p, lim), or -1.  The text at p is taken into registers once
// (12 dwords: literals are at most SP_MAXLEN bytes), the literals are stored dword-aligned and zero-padded, and a comparison
// goes a dword at a time in byte order (a byte-by-byte walk through the 40-byte common prefix of a thousand reserved tokens
// was ~440 dependent loads per match).
constexpr uint32_t SP_QWORDS = 12;
__device__ __forceinline__ int sp_match(const SpecialTable& S, const uint8_t* text, int64_t p, int64_t lim) {
    const int64_t avail64 = lim - p;
    const uint32_t avail = avail64 > (int64_t)(4 * SP_QWORDS) ? 4 * SP_QWORDS : (uint32_t)avail64;
    uint32_t Q[SP_QWORDS];
#pragma unroll
    for (uint32_t k = 0; k < SP_QWORDS; ++k) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            if (4 * k + j < avail) w |= (uint32_t)text[p + 4 * k + j] << (24 - 8 * j);  // (byte order: a dword compare is a memcmp)
        Q[k] = w;
    }
    // literal i <= Q ?  (prefix: the literal is a prefix of Q)
    auto le = [&](int i, bool& prefix) {
        const uint32_t o = S.off[i], len = S.len[i];
        const uint32_t* lw = reinterpret_cast<const uint32_t*>(S.bytes + o);
        prefix = false;
        bool decided = false, res = false;
#pragma unroll
        for (uint32_t k = 0; k < SP_QWORDS; ++k) {
            if (!decided && 4 * k < len) {
                const uint32_t nb = len - 4 * k < 4u ? len - 4 * k : 4u;           // literal bytes in this dword
                const uint32_t mask = nb == 4u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (8 * nb));
                const uint32_t x = __builtin_bswap32(lw[k]) & mask, y = Q[k] & mask;
                if (x != y) { decided = true; res = x < y; }
                else if (4 * k + nb > avail) { decided = true; res = false; }      // Q ends inside the literal (its bytes read as 0): the literal is greater
            }
        }
        if (!decided) { prefix = len <= avail; return prefix; }
        // equal bytes up to a difference: when the difference lies behind the end of Q the literal is the greater one
        return res;
    };
    int lo = -1, hi = (int)S.n;  // literal[lo] <= Q < literal[hi]
    bool pre = false, lo_pre = false;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (le(mid, pre)) { lo = mid; lo_pre = pre; } else hi = mid;
    }
    int i = lo;
    while (i >= 0 && !lo_pre) {  // up the chain of prefixes: everything in [longest prefix literal, Q] starts with it
        i = S.parent[i];
        if (i >= 0) (void)le(i, lo_pre);
    }
    return i;
}

}  // namespace

// candidates: positions whose first two bytes start an allowed literal, appended to a list (one atomic per wavefront).
// Matching them right here, under the few lanes of a wavefront that hold one, was 5 ms per 256 MiB of chat text: a match is
// a dozen dependent table reads, and the wavefront went through them once per distinct bit position of its candidates.
__global__ __launch_bounds__(256) void td_special_scan(const EncodeArgs a) {
    __shared__ uint32_t s_first2[2048];  // bit (b0 << 8 | b1): some allowed literal starts with these two bytes (1-byte literals: all b1)
    for (int q = threadIdx.x; q < 2048; q += blockDim.x) s_first2[q] = a.sp.first2[q];
    __syncthreads();
    constexpr int SP_LCAP = 2048, SP_LFLUSH = 1024;  // (a pass of the workgroup covers 8 KiB: at most 8192 candidates, 20 in chat text)
    __shared__ int64_t s_cpos[SP_LCAP];
    __shared__ uint32_t s_nc, s_base;
    if (threadIdx.x == 0) s_nc = 0;
    __syncthreads();
    auto flush_list = [&](uint32_t cnt_) {  // (all threads; s_nc is reset)
        if (threadIdx.x == 0) s_base = atomicAdd(a.sp.cand_count, cnt_);
        __syncthreads();
        const uint32_t base = s_base;
        for (uint32_t q = threadIdx.x; q < cnt_; q += blockDim.x) {
            if (base + q < a.sp.cand_cap) a.sp.cand_pos[base + q] = s_cpos[q];
            else raise_s(a, TD_E_SCRATCH, s_cpos[q]);  // (more than one candidate per 32 bytes of the whole batch)
        }
        __syncthreads();
        if (threadIdx.x == 0) s_nc = 0;
        __syncthreads();
    };
    const int64_t nw = (a.n + 31) >> 5;
    const int64_t nw_round = (nw + 255) & ~(int64_t)255;  // (whole workgroups take part in the barriers)
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < nw_round; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = w << 5;
        uint32_t cand = 0;
        if (w < nw) {
            uint32_t tw[9];  // my 32 bytes and the one behind them
            if (a.text_aligned && p0 + 36 <= a.n) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(a.text + p0), v1 = *reinterpret_cast<const uint4*>(a.text + p0 + 16);
                tw[0] = v0.x; tw[1] = v0.y; tw[2] = v0.z; tw[3] = v0.w; tw[4] = v1.x; tw[5] = v1.y; tw[6] = v1.z; tw[7] = v1.w;
                tw[8] = a.text[p0 + 32];
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    uint32_t x = 0;
                    for (int j = 0; j < 4; ++j)
                        if (p0 + 4 * k + j < a.n) x |= (uint32_t)a.text[p0 + 4 * k + j] << (8 * j);
                    tw[k] = x;
                }
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t b0 = (tw[k >> 2] >> (8 * (k & 3))) & 0xFFu, b1 = (tw[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFFu;
                const uint32_t key = (b0 << 8) | b1;
                if (p0 + k < a.n && ((s_first2[key >> 5] >> (key & 31)) & 1u)) cand |= 1u << k;
            }
            a.sp.hitbits[w] = 0;
            a.sp.accbits[w] = 0;
        }
        // the workgroup's candidates collect in LDS and go to the global list a few hundred at a time: one atomic on the list's
        // counter per wavefront with a candidate (nearly every one) was 1.3 ms per 256 MiB — same-address atomics are served
        // one after the other
        const uint32_t cnt = __popc(cand);
        uint32_t at = cnt ? atomicAdd(&s_nc, cnt) : 0u;
        for (uint32_t m = cand; m; m &= m - 1u) {
            const int64_t cp = p0 + (__ffs(m) - 1);
            if (at < (uint32_t)SP_LCAP) {
                s_cpos[at] = cp;
            } else {  // (text that is mostly candidates: what the LDS list cannot take goes to the global list one by one)
                const uint32_t g = atomicAdd(a.sp.cand_count, 1u);
                if (g < a.sp.cand_cap) a.sp.cand_pos[g] = cp;
                else raise_s(a, TD_E_SCRATCH, cp);
            }
            ++at;
        }
        __syncthreads();
        const uint32_t have = s_nc;
        if (have >= (uint32_t)SP_LFLUSH) flush_list(have < (uint32_t)SP_LCAP ? have : (uint32_t)SP_LCAP);
        __syncthreads();
    }
    if (s_nc) flush_list(s_nc < (uint32_t)SP_LCAP ? s_nc : (uint32_t)SP_LCAP);
}

// a lane per candidate: the longest allowed literal there (inside its document), or none
__global__ __launch_bounds__(256) void td_special_match(const EncodeArgs a) {
    const uint32_t nc = *a.sp.cand_count < a.sp.cand_cap ? *a.sp.cand_count : a.sp.cand_cap;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        const int64_t p = a.sp.cand_pos[j];
        const int i = sp_match(a.sp, a.text, p, sp_doc_limit(a, p, a.sp.maxlen));
        a.sp.cand_lit[j] = i;
        if (i >= 0) atomicOr(&a.sp.hitbits[p >> 5], 1u << (p & 31));
    }
}

// greedy left-to-right: a hit inside an accepted literal is dropped.  Hits further apart than a literal is long cannot touch,
// so a hit without another one in the maxlen - 1 bytes in front of it is accepted for sure; its lane walks its cluster
__global__ __launch_bounds__(256) void td_special_accept(const EncodeArgs a) {
    const uint32_t nc = *a.sp.cand_count < a.sp.cand_cap ? *a.sp.cand_count : a.sp.cand_cap;
    const int64_t back = (int64_t)a.sp.maxlen - 1;  // a literal that starts further back ends in front of the hit
    auto next_hit = [&](int64_t q, int64_t lim) {  // first hit in [q, lim), or lim
        while (q < lim) {
            const uint32_t m = a.sp.hitbits[q >> 5] >> (q & 31);
            if (m) { const int64_t f = q + (__ffs(m) - 1); return f < lim ? f : lim; }
            q = ((q >> 5) + 1) << 5;
        }
        return lim;
    };
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        int i = a.sp.cand_lit[j];
        if (i < 0) continue;
        const int64_t p = a.sp.cand_pos[j];
        const int64_t lo = p - back > 0 ? p - back : 0;
        if (next_hit(lo, p) < p) continue;  // not the head of its cluster: the head's lane gets here
        int64_t cur = p;
        for (;;) {  // cur is accepted, i its literal
            atomicOr(&a.sp.accbits[cur >> 5], 1u << (cur & 31));
            const int64_t end = cur + (int64_t)a.sp.len[i];
            // hits inside the literal are dropped; the next one at or behind its end is accepted if it belongs to this cluster
            int64_t last = cur, q = cur + 1;
            bool more = false;
            for (;;) {
                const int64_t lim = last + back + 1 < a.n ? last + back + 1 : a.n;
                const int64_t f = next_hit(q, lim);
                if (f >= lim) break;  // the cluster ends: the next hit is a head of its own
                last = f;
                if (f >= end) { cur = f; more = true; break; }
                q = f + 1;
            }
            if (!more) break;
            i = sp_match(a.sp, a.text, cur, sp_doc_limit(a, cur, a.sp.maxlen));
        }
    }
}

__global__ __launch_bounds__(256) void td_special_mark(const EncodeArgs a) {
    const uint32_t nc = *a.sp.cand_count < a.sp.cand_cap ? *a.sp.cand_count : a.sp.cand_cap;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        const int i = a.sp.cand_lit[j];
        const int64_t p = a.sp.cand_pos[j];
        if (i < 0 || !((a.sp.accbits[p >> 5] >> (p & 31)) & 1u)) continue;
        const int64_t e = p + (int64_t)a.sp.len[i];
        atomicOr(&a.docbits[p >> 5], 1u << (p & 31));
        if (e < a.n) atomicOr(&a.docbits[e >> 5], 1u << (e & 31));
    }
}

// (behind td_merge_pieces / td_long_pieces, in front of td_scan_tiles) the pieces of an accepted literal's own bytes: the
// first one becomes a marker with one id — the special's, stored where merged ids live — the others markers with none
__global__ __launch_bounds__(256) void td_special_ids(const EncodeArgs a) {
    const uint32_t nc = *a.sp.cand_count < a.sp.cand_cap ? *a.sp.cand_count : a.sp.cand_cap;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        const int i = a.sp.cand_lit[j];
        const int64_t p = a.sp.cand_pos[j];
        if (i < 0 || !((a.sp.accbits[p >> 5] >> (p & 31)) & 1u)) continue;
        const int64_t e = p + (int64_t)a.sp.len[i];
        bool first = true;
        int64_t tile = -1;
        uint32_t slot = 0, delta = 0;  // delta: what the literal's pieces in `tile` change about its id count (one atomic per tile)
        auto settle = [&]() {
            if (tile >= 0) {
                atomicAdd(&a.tile_extra[tile], delta);  // (the scan adds counts and extras modulo 2^32)
                atomicOr(&a.tile_count[tile], TILE_MISS_LISTED);
            }
            delta = 0;
        };
        for (int64_t q = p; q < e;) {  // q: a piece start inside the literal; its slot = pieces of its token tile in front of it
            if (q / K_TILE != tile) {
                settle();
                tile = q / K_TILE;
                const int64_t w0 = (tile * K_TILE) >> 5, w1 = q >> 5;  // (w0 is a multiple of 128 words: 16-byte loads)
                slot = 0;
                int64_t x = w0;
                for (; x + 4 <= w1; x += 4) {
                    const uint4 v4 = *reinterpret_cast<const uint4*>(a.startbits + x);
                    slot += (uint32_t)(__popc(v4.x) + __popc(v4.y) + __popc(v4.z) + __popc(v4.w));
                }
                for (; x < w1; ++x) slot += (uint32_t)__popc(a.startbits[x]);
                slot += (uint32_t)__popc(a.startbits[w1] & ((1u << (q & 31)) - 1u));
            }
            uint32_t* sp = a.stage + (size_t)tile * K_STAGE + slot;
            const uint32_t v = *sp;
            uint32_t had = 1;  // ids the piece was given
            if (v & TOK_LONGREF) had = a.long_list[v & 0x7FFFFFFFu].ntok;
            else if (v & TOK_MISS) had = v & 127u;
            const uint32_t pos = (uint32_t)(q - tile * K_TILE), now = first ? 1u : 0u;
            if (first) a.merge_out[(size_t)tile * K_STAGE + pos] = (uint32_t)a.sp.id[i];
            *sp = TOK_MISS | (pos << 7) | now;
            delta += now - had;
            first = false;
            ++slot;  // (the next piece of the same token tile)
            int64_t nq = e;  // next piece start behind q
            for (int64_t x = q + 1; x < e;) {
                const uint32_t mm = a.startbits[x >> 5] >> (x & 31);
                if (mm) { const int64_t f = x + (__ffs(mm) - 1); nq = f < e ? f : e; break; }
                x = ((x >> 5) + 1) << 5;
            }
            q = nq;
        }
        settle();
    }
}

static unsigned sp_blocks(const EncodeArgs& a) {
    int64_t b = (((a.n + 31) >> 5) + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

hipError_t launch_special_cuts(const EncodeArgs& a, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(a.sp.cand_count, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(td_special_scan, dim3(sp_blocks(a)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(td_special_match, dim3(2048), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(td_special_accept, dim3(2048), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(td_special_mark, dim3(2048), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// td_encode_batch's pipeline (td_api.cpp): a chunk's control block and token offsets written straight into pinned host memory
// by a kernel behind the chunk's kernels — as copies they were SDMA commands that wait for those kernels inside a copy engine's queue
__global__ void td_pipe_publish(const uint32_t* ctl, uint32_t ctl_words, uint32_t* h_ctl, const int64_t* d_toff, int64_t n_off, int64_t* h_toff) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    if (gid < ctl_words) h_ctl[gid] = ctl[gid];
    for (int64_t i = gid; i < n_off; i += gsz) h_toff[i] = d_toff[i];
    __threadfence_system();
}
// ... and a chunk's ids: device buffer -> pinned host buffer by a kernel (stores over PCIe), TD_PIPE_D2H_KERNEL=<workgroups> (default 32; 0: hipMemcpyAsync instead)
__global__ __launch_bounds__(256) void td_pipe_copy_out(const uint32_t* src, uint32_t* dst, int64_t n_words) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n_words >> 2;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (int64_t i = gid; i < n4; i += gsz) d4[i] = s4[i];
    if (gid < (n_words & 3)) dst[(n4 << 2) + gid] = src[(n4 << 2) + gid];
    __threadfence_system();
}
// td_encode_batch's path for 4 KiB .. 4 MiB (td_api.cpp: encode_batch_mid): behind the step's last kernel — whose outputs ARE pinned host
// buffers — the control block goes to pinned memory too and a sequence number is released at system scope: the host spins on it
__global__ __launch_bounds__(64) void td_mid_done(const uint32_t* ctl, uint32_t ctl_words, uint32_t* h_ctl, unsigned long long* h_seq, unsigned long long seq) {
    if (threadIdx.x < ctl_words) h_ctl[threadIdx.x] = ctl[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_mid_done(const void* ctl, uint32_t ctl_bytes, void* h_ctl, unsigned long long* h_seq, unsigned long long seq, hipStream_t stream) {
    hipLaunchKernelGGL(td_mid_done, dim3(1), dim3(64), 0, stream, (const uint32_t*)ctl, ctl_bytes / 4u, (uint32_t*)h_ctl, h_seq, seq);
    return hipGetLastError();
}
hipError_t launch_pipe_copy_out(const void* src, void* dst, int64_t n_words, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(td_pipe_copy_out, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, n_words);
    return hipGetLastError();
}
hipError_t launch_pipe_publish(const void* ctl, uint32_t ctl_bytes, void* h_ctl, const int64_t* d_toff, int64_t n_off, int64_t* h_toff, hipStream_t stream) {
    int64_t blocks = (n_off + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 512 ? 512 : blocks;
    hipLaunchKernelGGL(td_pipe_publish, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t*)ctl, ctl_bytes / 4u, (uint32_t*)h_ctl, d_toff, n_off, h_toff);
    return hipGetLastError();
}

hipError_t launch_special_ids(const EncodeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(td_special_ids, dim3(2048), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace td
