// VALU issue-rate microbenchmark for gfx950: how many cycles does a SIMD need per wave64 integer / bit
// instruction of the kinds the fused tile loop executes, at 1..8 wavefronts per SIMD?  Independent chains (8 accumulators), inline asm so
// that the compiler can neither fold nor reorder across kinds.  Times with HIP events; cycles are quoted at the clock measured with
// s_memrealtime-free arithmetic: a v_add_f32 / v_fma_f32 line is the calibration (the guides give their rate).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue_rate.hip -o /tmp/valu_issue_rate && /tmp/valu_issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4000;   // loop iterations per wave
constexpr int PER_ITER = 64;  // instructions of the measured kind per iteration (8 accumulators x 8)

// one asm statement = 8 independent instructions, one per accumulator
#define R8(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define ACC "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                          \
        uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u + 1u, r2 = r0 ^ 0x55u, r3 = r0 + 7u, r4 = r0 | 9u, r5 = r0 * 5u, r6 = r0 + 11u, r7 = r0 ^ 13u; \
        uint32_t c0 = seed | 3u, c1 = seed + 5u;                                                              \
        (void)c0; (void)c1;                                                                                   \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY }                                            \
        }                                                                                                     \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;                       \
    }

#define A8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : ACC : "v"(c0), "v"(c1));
#define S(i) "%" #i
// operand helpers: %N = accumulator N, %8 = c0, %9 = c1
#define I_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_AND(i)   "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_OR(i)    "v_or_b32 %" #i ", %" #i ", %8\n"
#define I_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_SHL(i)   "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define I_SHR(i)   "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define I_MOV(i)   "v_mov_b32 %" #i ", %8\n"
#define I_CNDM(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 3, 20\n"
#define I_BFI(i)   "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_OR3(i)   "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define I_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define I_BCNT(i)  "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define I_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define I_FFBL(i)  "v_ffbl_b32 %" #i ", %" #i "\n"
#define I_FFBH(i)  "v_ffbh_u32 %" #i ", %" #i "\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_CMP(i)   "v_cmp_eq_u32 vcc, %" #i ", %8\n"
#define I_CMPS(i)  "v_cmp_gt_u32 s[10:11], %" #i ", %8\n"
#define I_NOT(i)   "v_not_b32 %" #i ", %" #i "\n"
#define I_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define I_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define I_FADD(i)  "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_FFMA(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_DPPMOV(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPADD(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_RDLANE(i) "v_readlane_b32 s10, %" #i ", 5\n"
#define I_RDFIRST(i) "v_readfirstlane_b32 s10, %" #i "\n"
#define I_SDWA(i)  "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"

KERNEL(add, A8(I_ADD))
KERNEL(and, A8(I_AND))
KERNEL(or, A8(I_OR))
KERNEL(xor, A8(I_XOR))
KERNEL(shl, A8(I_SHL))
KERNEL(shr, A8(I_SHR))
KERNEL(mov, A8(I_MOV))
KERNEL(cndmask, A8(I_CNDM))
KERNEL(bfe, A8(I_BFE))
KERNEL(bfi, A8(I_BFI))
KERNEL(alignbit, A8(I_ALIGN))
KERNEL(and_or, A8(I_ANDOR))
KERNEL(or3, A8(I_OR3))
KERNEL(add3, A8(I_ADD3))
KERNEL(lshl_or, A8(I_LSHLOR))
KERNEL(lshl_add, A8(I_LSHLADD))
KERNEL(bcnt, A8(I_BCNT))
KERNEL(mbcnt, A8(I_MBCNT))
KERNEL(ffbl, A8(I_FFBL))
KERNEL(ffbh, A8(I_FFBH))
KERNEL(mul_lo, A8(I_MULLO))
KERNEL(mad_u24, A8(I_MAD24))
KERNEL(cmp_vcc, A8(I_CMP))
KERNEL(not, A8(I_NOT))
KERNEL(perm, A8(I_PERM))
KERNEL(bitop3, A8(I_BITOP3))
KERNEL(add_f32, A8(I_FADD))
KERNEL(fma_f32, A8(I_FFMA))
KERNEL(mov_dpp, A8(I_DPPMOV))
KERNEL(add_dpp, A8(I_DPPADD))
KERNEL(and_sdwa, A8(I_SDWA))

// readlane writes an SGPR: clobber list instead of accumulators
__global__ __launch_bounds__(256) void k_readlane(uint32_t* out, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t c0 = seed, c1 = seed;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile(I_RDLANE(0) I_RDLANE(1) I_RDLANE(2) I_RDLANE(3) I_RDLANE(4) I_RDLANE(5) I_RDLANE(6) I_RDLANE(7) : ACC : "v"(c0), "v"(c1) : "s10");
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;
}
__global__ __launch_bounds__(256) void k_cmp_sgpr(uint32_t* out, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t c0 = seed, c1 = seed;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile(I_CMPS(0) I_CMPS(1) I_CMPS(2) I_CMPS(3) I_CMPS(4) I_CMPS(5) I_CMPS(6) I_CMPS(7) : ACC : "v"(c0), "v"(c1) : "s10", "s11");
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;
}

// 64-bit kinds: 4 accumulators of 64 bits
#define ACC64 "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
#define KERNEL64(NAME, INS)                                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                          \
        uint64_t q0 = threadIdx.x + seed, q1 = q0 * 3u + 1u, q2 = q0 ^ 0x55u, q3 = q0 + 7u;                  \
        uint64_t d0 = seed | 3u; uint32_t c1 = 1;                                                             \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < 16; ++u)                                                    \
                asm volatile(INS(0) INS(1) INS(2) INS(3) : ACC64 : "v"(d0), "v"(c1) : "vcc");                 \
        }                                                                                                     \
        if ((q0 ^ q1 ^ q2 ^ q3) == 0x12345u) out[threadIdx.x] = (uint32_t)q0;                                 \
    }
#define J_SHL64(i)  "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define J_SHR64(i)  "v_lshrrev_b64 %" #i ", 1, %" #i "\n"
#define J_SHLV64(i) "v_lshlrev_b64 %" #i ", %5, %" #i "\n"
#define J_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %4\n"
#define J_MOV64(i)  "v_mov_b64 %" #i ", %4\n"
#define J_ADDCO(i)  "v_add_co_u32 %L" #i ", vcc, %L" #i ", %L4\n v_addc_co_u32 %H" #i ", vcc, %H" #i ", %H4, vcc\n"
KERNEL64(lshlrev_b64, J_SHL64)
KERNEL64(lshrrev_b64, J_SHR64)
KERNEL64(lshlrev_b64_v, J_SHLV64)
KERNEL64(lshl_add_u64, J_LSHLADD64)
KERNEL64(mov_b64, J_MOV64)

// the mix: proportions of the fused loop's hot kinds (static histogram of td_split_tiles<0,true,false>):
// 8 x { mov, and, add, shl, or, cndmask, shr, cmp } ~ one each
#define I_MIX(i) "v_and_b32 %" #i ", %" #i ", %8\n v_lshlrev_b32 %" #i ", 1, %" #i "\n v_add_u32 %" #i ", %" #i ", %9\n v_cmp_ne_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n v_or_b32 %" #i ", %" #i ", %9\n v_lshrrev_b32 %" #i ", 1, %" #i "\n v_bfe_u32 %" #i ", %" #i ", 1, 30\n"
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, uint32_t seed) {  // 64 instructions per asm statement
    uint32_t r0 = threadIdx.x + seed, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t c0 = seed | 0xff00ffu, c1 = seed + 5u;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(I_MIX(0) I_MIX(1) I_MIX(2) I_MIX(3) I_MIX(4) I_MIX(5) I_MIX(6) I_MIX(7) : ACC : "v"(c0), "v"(c1) : "vcc");
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;
}
// the same mix as ONE dependent chain per wave (what a latency-bound wave sees): 64 dependent instructions per iteration
__global__ __launch_bounds__(256) void k_mix_dep(uint32_t* out, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    uint32_t c0 = seed | 0xff00ffu, c1 = seed + 5u;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(I_MIX(0) I_MIX(0) I_MIX(0) I_MIX(0) I_MIX(0) I_MIX(0) I_MIX(0) I_MIX(0) : ACC : "v"(c0), "v"(c1) : "vcc");
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;
}

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); int per_iter; };
#define E(NAME, N) {#NAME, k_##NAME, N}

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;   // run one kernel only (for the --pmc pass)
    const int only_w = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;  // kHz -> MHz
    printf("# device %s, %d CUs, %d SIMDs, clockRate %.0f MHz; ITERS %d; a wavefront issues ITERS x per_iter instructions of the kind (+ 3 loop instr. per iteration)\n",
           prop.gcnArchName, cus, cus * 4, mhz, ITERS);
    printf("# grid = CUs x W workgroups of 256 threads (one wavefront per SIMD and workgroup): W wavefronts per SIMD\n");
    printf("# columns: kind | W | ms | cycles per wave-instruction per SIMD at clockRate (= time x clock / (W x instructions per wave))\n");
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    Entry es[] = {E(add_f32, 64), E(fma_f32, 64), E(add, 64), E(and, 64), E(or, 64), E(xor, 64), E(shl, 64), E(shr, 64), E(mov, 64), E(not, 64), E(cndmask, 64),
                  E(cmp_vcc, 64), E(cmp_sgpr, 64), E(bfe, 64), E(bfi, 64), E(alignbit, 64), E(perm, 64), E(and_or, 64), E(or3, 64), E(add3, 64), E(lshl_or, 64), E(lshl_add, 64),
                  E(bitop3, 64), E(bcnt, 64), E(mbcnt, 64), E(ffbl, 64), E(ffbh, 64), E(mul_lo, 64), E(mad_u24, 64), E(mov_dpp, 64), E(add_dpp, 64), E(and_sdwa, 64),
                  E(readlane, 64), E(lshlrev_b64, 64), E(lshrrev_b64, 64), E(lshlrev_b64_v, 64), E(lshl_add_u64, 64), E(mov_b64, 64), E(mix, 64), E(mix_dep, 64)};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int ws[] = {1, 2, 3, 4, 6, 8};
    for (const Entry& e : es) {
        if (only && strcmp(only, e.name) != 0) continue;
        for (int W : ws) {
            if (only_w && W != only_w) continue;
            const int blocks = cus * W;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1u);  // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr = (double)ITERS * e.per_iter;
            const double cyc = best * 1e-3 * mhz * 1e6 / (W * instr);
            printf("%-14s | %d | %8.4f | %6.3f\n", e.name, W, best, cyc);
        }
    }
    return 0;
}
