// microbench4.hip - round-2 questions about the strip kernel's loop on gfx950:
//  (1) issue cost of the cross-lane instructions a register transpose would use (v_permlane32_swap, v_permlane16_swap, DPP
//      row_ror:8 with a bank mask) against v_mov_b32;
//  (2) CU-level cost of the LDS instructions of the loop and of candidates (ds_write_addtid_b32, ds_write_b64 ...);
//  (3) does the lane order of the 8-byte pixel loads matter (lane = 8*row + block, the loop's order, against
//      lane = 8*block + row, which would let ds_write_addtid_b32 feed 16-byte transposed reads)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define RATE_KERNEL(NAME, BODY)                                                          \
    __global__ __launch_bounds__(256) void NAME(float *sink, int iters) {                \
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; \
        float c = 1.5f;                                                                  \
        for (int it = 0; it < iters; it++) {                                             \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); \
        }                                                                                \
        if (threadIdx.x == 9999) sink[0] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
    }
RATE_KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n")
RATE_KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n")
RATE_KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n")
RATE_KERNEL(k_dpp_ror8, "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                        "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n")
RATE_KERNEL(k_dpp_add, "v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                       "v_add_f32_dpp %4, %5, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %6, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %0, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")
RATE_KERNEL(k_fmaak, "v_fmaak_f32 %0, %0, %8, 0x4b400000\n v_fmaak_f32 %1, %1, %8, 0x4b400000\n v_fmaak_f32 %2, %2, %8, 0x4b400000\n v_fmaak_f32 %3, %3, %8, 0x4b400000\n v_fmaak_f32 %4, %4, %8, 0x4b400000\n v_fmaak_f32 %5, %5, %8, 0x4b400000\n v_fmaak_f32 %6, %6, %8, 0x4b400000\n v_fmaak_f32 %7, %7, %8, 0x4b400000\n")
RATE_KERNEL(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n")

// ---- LDS instruction cost per CU: every wave hammers its own 4 KiB with one instruction form -------------------------------
#define LDS_KERNEL(NAME, SETUP, BODY, NOPS)                                                            \
    __global__ __launch_bounds__(256) void NAME(float *sink, int iters) {                              \
        __shared__ __attribute__((aligned(16))) uint32_t buf[4][1024];                                 \
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                    \
        uint32_t base = (uint32_t)(uintptr_t)(&buf[wave][0]);                                          \
        uint32_t a4 = base + lane * 4, a8 = base + lane * 8, a16 = base + lane * 16, a2 = base + lane * 2; \
        float d0 = lane, d1 = 1, d2 = 2, d3 = 3;                                                       \
        typedef float f4 __attribute__((ext_vector_type(4)));                                          \
        f4 r = {0, 0, 0, 0};                                                                           \
        unsigned long long dd = lane;                                                                  \
        SETUP;                                                                                         \
        for (int it = 0; it < iters; it++) {                                                           \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY "s_waitcnt lgkmcnt(0)\n"             \
                         : "+v"(r), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(dd) : "v"(a4), "v"(a8), "v"(a16), "v"(a2) : "memory"); \
        }                                                                                              \
        if (threadIdx.x == 9999) sink[0] = r.x + r.y + r.z + r.w + d0 + (float)dd;                                 \
    }
LDS_KERNEL(l_write_b32, , "ds_write_b32 %6, %1\n", 8)
LDS_KERNEL(l_write_b16, , "ds_write_b16 %9, %1\n", 8)
LDS_KERNEL(l_write2_b32, , "ds_write2_b32 %6, %1, %2 offset1:64\n", 8)
LDS_KERNEL(l_write_b64, , "ds_write_b64 %7, %5\n", 8)
LDS_KERNEL(l_read_b128, , "ds_read_b128 %0, %8\n", 8)
LDS_KERNEL(l_read_b32, , "ds_read_b32 %1, %6\n", 8)
LDS_KERNEL(l_read2_b32, , "ds_read2_b32 %5, %6 offset1:64\n", 8)
LDS_KERNEL(l_addtid, asm volatile("s_mov_b32 m0, %0" : : "s"(__builtin_amdgcn_readfirstlane(base)) : "memory"), "ds_write_addtid_b32 %1\n", 8)

// ---- pixel-load lane order --------------------------------------------------------------------------------------------------
template <int ORDER>
__global__ __launch_bounds__(256) void k_load_order(const uint8_t *img, int stride, int strips_x, int nstrips, unsigned long long *sink) {
    const int lane = threadIdx.x & 63;
    const int r = ORDER == 0 ? lane >> 3 : lane & 7, b = ORDER == 0 ? lane & 7 : lane >> 3;
    unsigned long long acc = 0;
    const int nwaves = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < nstrips; t += nwaves) {
        const int ty = t / strips_x, tx = t - ty * strips_x;
        const uint8_t *p = img + ((long)ty * 8 + r) * stride + tx * 64 + b * 8;
        acc ^= *reinterpret_cast<const unsigned long long *>(p);
    }
    if (acc == 0x1234567812345678ull) sink[0] = acc;
}

template <typename K>
static int time_rate(const char *name, K kern, float *sink, int ncu, double per_iter) {
    printf("%-22s", name);
    for (int wg_per_cu : {1, 2, 5, 8}) {
        const int iters = 2000, wgs = ncu * wg_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, sink, 10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, sink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw: %6.2f ns", wg_per_cu, ms * 1e6 / ((double)wg_per_cu * iters * per_iter));
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    float *sink;
    CK(hipMalloc(&sink, 64));
    printf("VALU / cross-lane: ns per wave-instruction per SIMD (w = waves per SIMD)\n");
    time_rate("v_mov_b32", k_mov, sink, ncu, 64); printf("\n");
    time_rate("v_permlane32_swap", k_swap32, sink, ncu, 64); printf("\n");
    time_rate("v_permlane16_swap", k_swap16, sink, ncu, 64); printf("\n");
    time_rate("v_mov_b32_dpp ror:8", k_dpp_ror8, sink, ncu, 64); printf("\n");
    time_rate("v_add_f32_dpp ror:8", k_dpp_add, sink, ncu, 64); printf("\n");
    time_rate("v_fmaak_f32", k_fmaak, sink, ncu, 64); printf("\n");
    time_rate("v_fract_f32", k_fract, sink, ncu, 64); printf("\n");
    printf("LDS: ns per wave-instruction per CU (w = workgroups of 4 waves per CU); x clock GHz = LDS cycles\n");
    time_rate("ds_write_b32", l_write_b32, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_write_b16", l_write_b16, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_write2_b32", l_write2_b32, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_write_b64", l_write_b64, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_write_addtid_b32", l_addtid, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_read_b128", l_read_b128, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_read_b32", l_read_b32, sink, ncu, 64 * 4); printf("\n");
    time_rate("ds_read2_b32", l_read2_b32, sink, ncu, 64 * 4); printf("\n");
    // pixel-load lane order on a 16384 x 16384 frame (256 MiB: beyond the Infinity Cache) and on 4096 x 4096
    for (int dim : {4096, 16384}) {
        uint8_t *img;
        unsigned long long *s2;
        CK(hipMalloc(&img, (size_t)dim * dim));
        CK(hipMalloc(&s2, 64));
        CK(hipMemset(img, 1, (size_t)dim * dim));
        const int strips_x = dim / 64, nstrips = strips_x * (dim / 8);
        for (int order = 0; order < 2; order++) {
            for (int wgs : {1280, 2048}) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                const int reps = dim == 4096 ? 200 : 20;
                for (int k = 0; k < 3; k++) {
                    if (order == 0) hipLaunchKernelGGL(k_load_order<0>, dim3(wgs), dim3(256), 0, 0, img, dim, strips_x, nstrips, s2);
                    else hipLaunchKernelGGL(k_load_order<1>, dim3(wgs), dim3(256), 0, 0, img, dim, strips_x, nstrips, s2);
                }
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int k = 0; k < reps; k++) {
                    if (order == 0) hipLaunchKernelGGL(k_load_order<0>, dim3(wgs), dim3(256), 0, 0, img, dim, strips_x, nstrips, s2);
                    else hipLaunchKernelGGL(k_load_order<1>, dim3(wgs), dim3(256), 0, 0, img, dim, strips_x, nstrips, s2);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("load order %s, %5d^2, %4d workgroups: %8.2f us per pass = %7.1f GB/s read\n", order == 0 ? "lane=8*row+block" : "lane=8*block+row", dim, wgs,
                       ms * 1e3 / reps, (double)dim * dim / (ms * 1e-3 / reps) / 1e9);
            }
        }
        CK(hipFree(img));
        CK(hipFree(s2));
    }
    return 0;
}
