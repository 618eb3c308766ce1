// Second pass of the VALU issue-rate microbenchmark (tools/microbench/valu_issue_rate.hip has the notes): the forms the first pass left open —
// v_cndmask with VCC / with an SGPR pair / right behind the v_cmp that makes its mask, VGPR shift amounts, literal operands, carry chains,
// min / max / sub, and the LDS instructions of the fused tile loop (issue rate per wavefront-instruction, no bank conflicts).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 4000;
#define ACC "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define KERNEL(NAME, PRE, BODY, ...)                                                                        \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                          \
        uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u + 1u, r2 = r0 ^ 0x55u, r3 = r0 + 7u, r4 = r0 | 9u, r5 = r0 * 5u, r6 = r0 + 11u, r7 = r0 ^ 13u; \
        uint32_t c0 = (seed | 3u) & 31u, c1 = seed + 5u;                                                      \
        asm volatile(PRE : : "v"(c0), "v"(c1) : "vcc", "s10", "s11");                                       \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                     \
                asm volatile(BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7) : ACC : "v"(c0), "v"(c1) : __VA_ARGS__); \
        }                                                                                                     \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;                       \
    }
#define B_CND_VCC(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define B_CND_SGPR(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define B_CMP_CND(i)   "v_cmp_gt_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define B_CMP_CND_S(i) "v_cmp_gt_u32_e64 s[10:11], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define B_SHLV(i)      "v_lshlrev_b32 %" #i ", %8, %" #i "\n"
#define B_SHRV(i)      "v_lshrrev_b32 %" #i ", %8, %" #i "\n"
#define B_ASHR(i)      "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define B_ANDLIT(i)    "v_and_b32 %" #i ", 0x0f0f0f0f, %" #i "\n"
#define B_ADDLIT(i)    "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define B_SUB(i)       "v_sub_u32 %" #i ", %" #i ", %8\n"
#define B_MIN(i)       "v_min_u32 %" #i ", %" #i ", %9\n"
#define B_MAX(i)       "v_max_u32 %" #i ", %" #i ", %8\n"
#define B_ADDCO(i)     "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define B_ADDC(i)      "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define B_XAD(i)       "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define B_MUL24(i)     "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define B_MULHI(i)     "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define B_MED3(i)      "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define B_SADD(i)      "s_add_u32 s10, s10, 1\n"
#define B_ANDSH(i)     "v_and_b32 %" #i ", %" #i ", %9\n v_lshrrev_b32 %" #i ", 1, %" #i "\n"
KERNEL(cndmask_vcc, "v_cmp_gt_u32 vcc, %0, %1\n", B_CND_VCC, "memory")
KERNEL(cndmask_sgpr, "v_cmp_gt_u32_e64 s[10:11], %0, %1\n", B_CND_SGPR, "memory")
KERNEL(cmp_cndmask_vcc, "", B_CMP_CND, "vcc")
KERNEL(cmp_cndmask_sgpr, "", B_CMP_CND_S, "s10", "s11")
KERNEL(shl_v, "", B_SHLV, "memory")
KERNEL(shr_v, "", B_SHRV, "memory")
KERNEL(ashr, "", B_ASHR, "memory")
KERNEL(and_literal, "", B_ANDLIT, "memory")
KERNEL(add_literal, "", B_ADDLIT, "memory")
KERNEL(sub, "", B_SUB, "memory")
KERNEL(min_u32, "", B_MIN, "memory")
KERNEL(max_u32, "", B_MAX, "memory")
KERNEL(add_co, "", B_ADDCO, "vcc")
KERNEL(addc_co, "", B_ADDC, "vcc")
KERNEL(xad, "", B_XAD, "memory")
KERNEL(mul_u24, "", B_MUL24, "memory")
KERNEL(mul_hi, "", B_MULHI, "memory")
KERNEL(med3, "", B_MED3, "memory")
KERNEL(salu_add, "", B_SADD, "s10", "scc")
KERNEL(and_shr_pair, "", B_ANDSH, "memory")

// LDS: every lane its own dword / byte of a 16 KB array (no bank conflicts for dwords: consecutive lanes, consecutive dwords)
#define LKERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                          \
        __shared__ __attribute__((aligned(16))) uint32_t sm[4096];                                            \
        for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i * seed;                                       \
        __syncthreads();                                                                                      \
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                              \
        const uint32_t a4 = (uint32_t)(uintptr_t)(sm) + threadIdx.x * 4u, a16 = (uint32_t)(uintptr_t)(sm) + threadIdx.x * 16u, a1 = (uint32_t)(uintptr_t)(sm) + threadIdx.x, a8 = (uint32_t)(uintptr_t)(sm) + threadIdx.x * 8u; \
        (void)a4; (void)a16; (void)a1; (void)a8;                                                              \
        for (int it = 0; it < ITERS / 4; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY }                                            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        }                                                                                                     \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[threadIdx.x] = r0;                       \
    }
LKERNEL(ds_read_b32, asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a4) : "memory");)
LKERNEL(ds_read_u8, asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:256\n ds_read_u8 %2, %8 offset:512\n ds_read_u8 %3, %8 offset:768\n ds_read_u8 %4, %8 offset:1024\n ds_read_u8 %5, %8 offset:1280\n ds_read_u8 %6, %8 offset:1536\n ds_read_u8 %7, %8 offset:1792\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a1) : "memory");)
LKERNEL(ds_write_b32, asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:1024\n ds_write_b32 %8, %2 offset:2048\n ds_write_b32 %8, %3 offset:3072\n ds_write_b32 %8, %4 offset:4096\n ds_write_b32 %8, %5 offset:5120\n ds_write_b32 %8, %6 offset:6144\n ds_write_b32 %8, %7 offset:7168\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a4) : "memory");)
LKERNEL(ds_write_b8, asm volatile("ds_write_b8 %8, %0\n ds_write_b8 %8, %1 offset:256\n ds_write_b8 %8, %2 offset:512\n ds_write_b8 %8, %3 offset:768\n ds_write_b8 %8, %4 offset:1024\n ds_write_b8 %8, %5 offset:1280\n ds_write_b8 %8, %6 offset:1536\n ds_write_b8 %8, %7 offset:1792\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a1) : "memory");)
// the class-mask stores of the fused loop: a byte every 8 bytes per lane (stride 8: the lanes of a wavefront hit 16 banks four deep)
LKERNEL(ds_write_b8_stride8, asm volatile("ds_write_b8 %8, %0\n ds_write_b8 %8, %1 offset:1\n ds_write_b8 %8, %2 offset:2\n ds_write_b8 %8, %3 offset:3\n ds_write_b8 %8, %4 offset:4\n ds_write_b8 %8, %5 offset:5\n ds_write_b8 %8, %6 offset:6\n ds_write_b8 %8, %7 offset:7\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a8) : "memory");)
__global__ __launch_bounds__(256) void k_ds_read_b128(uint32_t* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * seed;
    __syncthreads();
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    const uint32_t a16 = (uint32_t)(uintptr_t)(sm) + threadIdx.x * 16u;
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(a16) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if ((q0.x ^ q1.y ^ q2.z ^ q3.w) == 0x12345u) out[threadIdx.x] = q0.x;
}

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); double per_wave; };
#define E(NAME, N) {#NAME, k_##NAME, N}
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("# device %s, %d CUs, clockRate %.0f MHz.  columns: kind | W (wavefronts per SIMD) | ms | cycles per wave-instruction per SIMD at clockRate\n", prop.gcnArchName, cus, mhz);
    printf("# (pairs: cmp_cndmask_*, and_shr_pair count TWO instructions per step; LDS kinds: a SIMD's share of the CU's one LDS pipe)\n");
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    const double N = (double)ITERS * 64;
    Entry es[] = {E(cndmask_vcc, N), E(cndmask_sgpr, N), E(cmp_cndmask_vcc, 2 * N), E(cmp_cndmask_sgpr, 2 * N), E(shl_v, N), E(shr_v, N), E(ashr, N), E(and_literal, N), E(add_literal, N),
                  E(sub, N), E(min_u32, N), E(max_u32, N), E(add_co, N), E(addc_co, N), E(xad, N), E(mul_u24, N), E(mul_hi, N), E(med3, N), E(salu_add, N), E(and_shr_pair, 2 * N),
                  E(ds_read_b32, N / 4), E(ds_read_u8, N / 4), E(ds_write_b32, N / 4), E(ds_write_b8, N / 4), E(ds_write_b8_stride8, N / 4), E(ds_read_b128, N / 4)};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int ws[] = {1, 2, 4, 6};
    for (const Entry& e : es) {
        if (only && strcmp(only, e.name) != 0) continue;
        for (int W : ws) {
            const int blocks = cus * W;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
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
            printf("%-20s | %d | %8.4f | %6.3f\n", e.name, W, best, best * 1e-3 * mhz * 1e6 / (W * e.per_wave));
        }
    }
    return 0;
}
