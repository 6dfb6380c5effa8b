// microbench3.hip - cost of the hybrid loop's arithmetic pieces (compiled C++, as in the kernel), no memory, no LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../tinyimgcodec_amd/csrc/tic_math.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace tic;

__global__ __launch_bounds__(256) void k_dct(float *sink, int iters) {
    float d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
    for (int it = 0; it < iters; it++) {
        dct8_aan(d0, d1, d2, d3, d4, d5, d6, d7);
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
    }
    if (threadIdx.x == 9999) sink[0] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}
__global__ __launch_bounds__(256) void k_pass1(float *sink, int iters) {
    uint32_t lo = threadIdx.x * 2654435761u, hi = lo ^ 0x12345678u;
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        float d0 = (float)(lo & 0xffu), d1 = (float)((lo >> 8) & 0xffu), d2 = (float)((lo >> 16) & 0xffu), d3 = (float)(lo >> 24);
        float d4 = (float)(hi & 0xffu), d5 = (float)((hi >> 8) & 0xffu), d6 = (float)((hi >> 16) & 0xffu), d7 = (float)(hi >> 24);
        dct8_aan(d0, d1, d2, d3, d4, d5, d6, d7);
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(lo), "+v"(hi));
        acc += d0;
    }
    if (threadIdx.x == 9999) sink[0] = acc;
}
__device__ __forceinline__ void quant_magic(float z, float mul, uint32_t &bits, float &d) {
    const float t = z * mul;
    const float s = t + kMagic;
    bits = __float_as_uint(s);
    d = t - (s - kMagic);
}
__global__ __launch_bounds__(256) void k_quant(float *sink, int iters) {
    float e0 = threadIdx.x, e1 = 1, e2 = 2, e3 = 3, e4 = 4, e5 = 5, e6 = 6, e7 = 7;
    float m = 0.37f, thr = 0.49f;
    uint32_t keep = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
        float r0, r1, r2, r3, r4, r5, r6, r7;
        quant_magic(e0, m, q0, r0); quant_magic(e1, m, q1, r1); quant_magic(e2, m, q2, r2); quant_magic(e3, m, q3, r3);
        quant_magic(e4, m, q4, r4); quant_magic(e5, m, q5, r5); quant_magic(e6, m, q6, r6); quant_magic(e7, m, q7, r7);
        float mA = fmaxf(fmaxf(fabsf(r1), fabsf(r2)), fabsf(r3));
        mA = fmaxf(fmaxf(mA, fabsf(r5)), fabsf(r6));
        mA = fmaxf(mA, fabsf(r7));
        const float mB = fmaxf(fabsf(r0), fabsf(r4));
        unsigned long long cA = __ballot(mA > thr), cB = __ballot(mB > thr);
        keep ^= (uint32_t)cA ^ (uint32_t)cB;
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
    }
    if (threadIdx.x == 9999) sink[0] = (float)keep;
}

template <typename K>
static int time_k(const char *name, K kern, float *sink, int ncu) {
    printf("%-10s", name);
    for (int wg_per_cu : {1, 2, 5, 8}) {
        const int iters = 4000, wgs = ncu * wg_per_cu;
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
        printf("  %dw: %6.1f ns", wg_per_cu, ms * 1e6 / ((double)wg_per_cu * iters));
    }
    printf("   (ns per body per SIMD)\n");
    return 0;
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    float *sink;
    CK(hipMalloc(&sink, 64));
    time_k("dct8_aan", k_dct, sink, p.multiProcessorCount);
    time_k("pass1", k_pass1, sink, p.multiProcessorCount);
    time_k("quant", k_quant, sink, p.multiProcessorCount);
    return 0;
}
