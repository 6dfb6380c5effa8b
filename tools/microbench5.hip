// microbench5.hip - what does one launch cost on gfx950 as a function of grid shape?  Back-to-back launches of kernels
// that do nothing (or one 16-byte store per wave), same stream, HIP events around N launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct BigArgs { long a[30]; };
__global__ void k_empty() {}
__global__ void k_empty_args(BigArgs a, float *out) { if (a.a[3] == 0x7fffffffffff) out[0] = 1.0f; }
__global__ void k_store(uint4 *out) { out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = make_uint4(1, 2, 3, 4); }
__global__ void k_store_sc(uint4 *out) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = {1, 2, 3, 4};
    uint4 *p = out + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(d) : "memory");
}

template <typename F>
static float time_launches(F launch, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 200; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}

int main() {
    uint4 *buf;
    CK(hipMalloc(&buf, 64u << 20));
    float *out = (float *)buf;
    BigArgs ba = {};
    const int N = 3000;
    const int shapes[][2] = {{256, 64}, {256, 256}, {256, 512}, {256, 1024}, {512, 256}, {512, 512}, {768, 256}, {1024, 256}, {1280, 256},
                             {640, 512}, {320, 1024}, {2048, 256}, {2560, 256}, {5120, 64}, {1280, 64}, {1280, 128}};
    printf("%-22s %10s %10s %10s %10s\n", "grid x block (waves)", "empty", "args", "store16", "store sc");
    for (auto &s : shapes) {
        const int g = s[0], b = s[1];
        float t0 = time_launches([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(b), 0, 0); }, N);
        float t1 = time_launches([&] { hipLaunchKernelGGL(k_empty_args, dim3(g), dim3(b), 0, 0, ba, out); }, N);
        float t2 = time_launches([&] { hipLaunchKernelGGL(k_store, dim3(g), dim3(b), 0, 0, buf); }, N);
        float t3 = time_launches([&] { hipLaunchKernelGGL(k_store_sc, dim3(g), dim3(b), 0, 0, buf); }, N);
        printf("%5d x %4d (%5d)    %8.2f us %8.2f us %8.2f us %8.2f us\n", g, b, g * b / 64, t0, t1, t2, t3);
    }
    return 0;
}
