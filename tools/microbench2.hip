// microbench2.hip - VALU issue rates on gfx950 as a function of waves per SIMD and of dependence, for the instruction
// forms the hybrid kernel's loop actually uses (SDWA, literals, packed f32, conversions).  Build: see tools/gpu_run.sh.
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
#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define D8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n" OP " %0, %0" TAIL "\n"

RATE_KERNEL(k_add_ind, I8("v_add_f32", ", %8"))
RATE_KERNEL(k_add_dep, D8("v_add_f32", ", %8"))
RATE_KERNEL(k_add_lit, "v_add_f32 %0, 0x4b400000, %0\n v_add_f32 %1, 0x4b400000, %1\n v_add_f32 %2, 0x4b400000, %2\n v_add_f32 %3, 0x4b400000, %3\n v_add_f32 %4, 0x4b400000, %4\n v_add_f32 %5, 0x4b400000, %5\n v_add_f32 %6, 0x4b400000, %6\n v_add_f32 %7, 0x4b400000, %7\n")
RATE_KERNEL(k_mul_ind, I8("v_mul_f32", ", %8"))
RATE_KERNEL(k_fma_ind, I8("v_fma_f32", ", %8, %8"))
RATE_KERNEL(k_fmac_ind, I8("v_fmac_f32", ", %8"))
RATE_KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n")
RATE_KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7\n")
RATE_KERNEL(k_add_u32, I8("v_add_u32", ", %8"))
RATE_KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3\n"
                        "v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_0\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3\n")
RATE_KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n")
RATE_KERNEL(k_max3, "v_max3_f32 %0, %0, |%1|, |%2|\n v_max3_f32 %1, %1, |%2|, |%3|\n v_max3_f32 %2, %2, |%3|, |%4|\n v_max3_f32 %3, %3, |%4|, |%5|\n v_max3_f32 %4, %4, |%5|, |%6|\n v_max3_f32 %5, %5, |%6|, |%7|\n v_max3_f32 %6, %6, |%7|, |%0|\n v_max3_f32 %7, %7, |%0|, |%1|\n")
RATE_KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f3504f3, %8\n v_fmamk_f32 %1, %1, 0x3f3504f3, %8\n v_fmamk_f32 %2, %2, 0x3f3504f3, %8\n v_fmamk_f32 %3, %3, 0x3f3504f3, %8\n v_fmamk_f32 %4, %4, 0x3f3504f3, %8\n v_fmamk_f32 %5, %5, 0x3f3504f3, %8\n v_fmamk_f32 %6, %6, 0x3f3504f3, %8\n v_fmamk_f32 %7, %7, 0x3f3504f3, %8\n")

// packed f32 on register pairs
__global__ __launch_bounds__(256) void k_pk_add(float *sink, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, c = 1.0;
    for (int it = 0; it < iters; it++) {
#define PK4 "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
        asm volatile(PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 PK4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
    }
    if (threadIdx.x == 9999) sink[0] = (float)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(256) void k_pk_mul(float *sink, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, c = 1.0;
    for (int it = 0; it < iters; it++) {
#define PM4 "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
        asm volatile(PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 PM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
    }
    if (threadIdx.x == 9999) sink[0] = (float)(a0 + a1 + a2 + a3);
}

template <typename K>
static int time_rate(const char *name, K kern, float *sink, int ncu) {
    printf("%-16s", name);
    for (int wg_per_cu : {1, 2, 4, 5, 8}) {
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
        double per_simd = (double)wg_per_cu * iters * 64; // wave-instructions per SIMD (one wave of each WG per SIMD)
        printf("  %dw: %5.2f ns", wg_per_cu, ms * 1e6 / per_simd);
    }
    printf("   (ns per wave-instruction per SIMD)\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    float *sink;
    CK(hipMalloc(&sink, 64));
    time_rate("v_add_f32 indep", k_add_ind, sink, ncu);
    time_rate("v_add_f32 dep", k_add_dep, sink, ncu);
    time_rate("v_add_f32 lit", k_add_lit, sink, ncu);
    time_rate("v_mul_f32", k_mul_ind, sink, ncu);
    time_rate("v_fma_f32", k_fma_ind, sink, ncu);
    time_rate("v_fmac_f32", k_fmac_ind, sink, ncu);
    time_rate("v_fmamk_f32", k_fmamk, sink, ncu);
    time_rate("v_cvt_f32_i32", k_cvt_f32_i32, sink, ncu);
    time_rate("v_cvt_f32_ubyte", k_cvt_ubyte, sink, ncu);
    time_rate("v_add_u32", k_add_u32, sink, ncu);
    time_rate("v_add_u32_sdwa", k_add_sdwa, sink, ncu);
    time_rate("v_mov_b32", k_mov, sink, ncu);
    time_rate("v_max3_f32 |.|", k_max3, sink, ncu);
    time_rate("v_pk_add_f32", k_pk_add, sink, ncu);
    time_rate("v_pk_mul_f32", k_pk_mul, sink, ncu);
    return 0;
}
