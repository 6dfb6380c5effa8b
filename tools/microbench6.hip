// microbench6.hip - packed float32 on gfx950: do v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 issue at the rate of their
// scalar forms (two results per lane per instruction), also with op_sel half selection and neg modifiers?  The in-lane packing
// of the strip kernel's 8-point DCT (17 instructions instead of 36) pays only if they do.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define RATE_KERNEL(NAME, BODY)                                                          \
    __global__ __launch_bounds__(256) void NAME(float *sink, int iters) {                \
        f2 a0 = {(float)threadIdx.x, 1}, a1 = {1, 2}, a2 = {2, 3}, a3 = {3, 4}, a4 = {4, 5}, a5 = {5, 6}, a6 = {6, 7}, a7 = {7, 8}; \
        f2 c = {1.5f, 0.5f};                                                             \
        for (int it = 0; it < iters; it++) {                                             \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); \
        }                                                                                \
        if (threadIdx.x == 9999) sink[0] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y; \
    }
RATE_KERNEL(k_pk_add, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
RATE_KERNEL(k_pk_add_sel, "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel:[1,1] op_sel_hi:[0,0]\n"
                          "v_pk_add_f32 %4, %4, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %5, %5, %6 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %6, %6, %7 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %7, %7, %0 op_sel:[1,1] op_sel_hi:[0,0]\n")
RATE_KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
RATE_KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %0, %8, %1\n v_pk_fma_f32 %1, %1, %8, %2\n v_pk_fma_f32 %2, %2, %8, %3\n v_pk_fma_f32 %3, %3, %8, %4\n v_pk_fma_f32 %4, %4, %8, %5\n v_pk_fma_f32 %5, %5, %8, %6\n v_pk_fma_f32 %6, %6, %8, %7\n v_pk_fma_f32 %7, %7, %8, %0\n")
RATE_KERNEL(k_pk_fma_sel, "v_pk_fma_f32 %0, %0, %8, %1 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_hi:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %2 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_hi:[0,1,0]\n v_pk_fma_f32 %2, %2, %8, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %4 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n"
                          "v_pk_fma_f32 %4, %4, %8, %5 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_hi:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %6 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %7 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %0 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n")
RATE_KERNEL(k_pk_mov, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]\n v_pk_mov_b32 %1, %2, %3 op_sel:[1,0]\n v_pk_mov_b32 %2, %3, %4 op_sel:[0,1]\n v_pk_mov_b32 %3, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %4, %5, %6 op_sel:[0,1]\n v_pk_mov_b32 %5, %6, %7 op_sel:[1,0]\n v_pk_mov_b32 %6, %7, %0 op_sel:[0,1]\n v_pk_mov_b32 %7, %0, %1 op_sel:[1,0]\n")

template <typename K>
static int time_rate(const char *name, K kern, float *sink, int ncu, double per_iter) {
    printf("%-26s", name);
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
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    float *sink;
    CK(hipMalloc(&sink, 64));
    printf("ns per wave-instruction per SIMD (w = waves per SIMD); scalar forms, microbench4: v_add_f32/v_fmaak_f32 1.0 ns, v_fma_f32 1.56 ns\n");
    time_rate("v_pk_add_f32", k_pk_add, sink, ncu, 64);
    time_rate("v_pk_add_f32 op_sel/neg", k_pk_add_sel, sink, ncu, 64);
    time_rate("v_pk_mul_f32", k_pk_mul, sink, ncu, 64);
    time_rate("v_pk_fma_f32", k_pk_fma, sink, ncu, 64);
    time_rate("v_pk_fma_f32 op_sel/neg", k_pk_fma_sel, sink, ncu, 64);
    time_rate("v_pk_mov_b32", k_pk_mov, sink, ncu, 64);
    return 0;
}
