// Generic split patterns on the device: what td_regex.cpp compiled is matched by the backtracking matcher of
// td_regex.h, ONE LANE PER DOCUMENT (documents are independent subjects, tiktoken.cpp:86-122; without pattern-specific
// synchronisation rules there is nothing provable inside one).  The kernel writes the same START bitmap the family's
// td_split_tiles writes, so td_probe_tiles and everything behind it run unchanged.
//
// Text the pattern SKIPS (the reference tokenizes only what matches) becomes a piece of its own in the bitmap and is noted on
// a list; after the merge kernels td_generic_gaps turns such a piece's slot into a marker with zero ids and takes the ids it
// had been given out of the tile's count, so td_pack_tokens (which already expands markers of any size) leaves it out.
// This first form is as fast as its longest document is long; the members of the family keep their own kernels.
#include <hip/hip_runtime.h>

#include "td_kernels.h"
#include "td_regex.h"

namespace td {

namespace {
struct GlobalDoc {  // the subject: one document of the batch
    const uint8_t* p;
    __device__ __forceinline__ uint32_t byte(int64_t i) const { return p[i]; }
};
__device__ __forceinline__ void raise_g(const EncodeArgs& a, int code, int64_t pos) {
    if (atomicCAS(a.err, 0, code) == 0) *a.err_pos = pos;
}
}  // namespace

__global__ __launch_bounds__(256) void td_split_generic(const EncodeArgs a) {
    const RxTables T{a.rx_stage1, a.rx_stage2};
    // the compiled pattern in LDS: the matcher reads a node, a class or a literal at every step, and out of HBM each of
    // those was a cache round trip in the middle of a dependent chain
    __shared__ RxProgram sP;
    static_assert(sizeof(RxProgram) % 4 == 0, "copied as dwords");
    for (uint32_t w = threadIdx.x; w < sizeof(RxProgram) / 4; w += blockDim.x)
        reinterpret_cast<uint32_t*>(&sP)[w] = reinterpret_cast<const uint32_t*>(a.rx)[w];
    __syncthreads();
    const RxProgram& P = sP;
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < a.n_docs; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o0 = a.doc_offsets[d], o1 = a.doc_offsets[d + 1];
        if (o0 < 0 || o1 > a.n || o1 <= o0) continue;
        const GlobalDoc s{a.text + o0};
        const int64_t n = o1 - o0;
        for (int64_t pos = 0; pos < n;) {
            int64_t ms, me;
            rx_next_piece(P, T, s, pos, n, ms, me);
            if (ms > pos) {  // skipped text: a piece of its own, without tokens
                const uint32_t gi = atomicAdd(a.gap_count, 1u);
                if (gi < a.gap_cap) a.gap_list[gi] = o0 + pos;
                else raise_g(a, TD_E_SCRATCH, o0 + pos);
                atomicOr(&a.startbits[(o0 + pos) >> 5], 1u << ((o0 + pos) & 31));
            }
            atomicOr(&a.startbits[(o0 + ms) >> 5], 1u << ((o0 + ms) & 31));
            pos = me;
        }
    }
}

// one lane per skipped stretch: its slot (= pieces of its tile in front of it) becomes TOK_MISS | position | 0 ids
__global__ __launch_bounds__(256) void td_generic_gaps(const EncodeArgs a) {
    const uint32_t ng = *a.gap_count < a.gap_cap ? *a.gap_count : a.gap_cap;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += gridDim.x * blockDim.x) {
        const int64_t p = a.gap_list[g];
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

hipError_t launch_generic_split(const EncodeArgs& a, hipStream_t stream) {
    const size_t words = (size_t)((a.n + 31) / 32 + 8);
    hipError_t e = hipMemsetAsync(a.startbits, 0, words * 4, stream);
    if (e != hipSuccess) return e;
    int64_t blocks = (a.n_docs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(td_split_generic, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_generic_gaps(const EncodeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(td_generic_gaps, dim3(256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace td
