// microbench.hip - MI355X calibration numbers used in DESIGN.md: streaming bandwidth for this kernel's traffic
// shape (1 byte in : 2 bytes out) and issue rates of the instructions the transform kernel leans on.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void copy16(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// traffic shape of the transform kernel, no arithmetic: per wave 8 rows x 64 B in, 1 KiB out.
// mode 0: lane = 8*block + row (8 lanes of a block read 8 different rows); mode 1: lane = 8*row + chunk (row-contiguous)
__global__ __launch_bounds__(256) void shape_io(const uint8_t *__restrict__ img, int w, long stride, int tiles_x, int ntiles,
                                                uint4 *__restrict__ out, int mode) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    int b, r;
    if (mode == 0) {
        b = lane >> 3;
        r = lane & 7;
    } else {
        r = lane >> 3;
        b = lane & 7;
    }
    const uint2 v = *reinterpret_cast<const uint2 *>(img + (long)(ty * 8 + r) * stride + (tx * 8 + b) * 8);
    uint4 o;
    o.x = v.x;
    o.y = v.y;
    o.z = v.x ^ 0x80808080u;
    o.w = v.y ^ 0x80808080u;
    size_t oblk = (size_t)ty * (w / 8) + tx * 8;
    out[oblk * 8 + lane] = o;
}

#define RATE_KERNEL(NAME, DECL, BODY)                                  \
    __global__ __launch_bounds__(256) void NAME(float *sink, int iters) { \
        DECL;                                                          \
        for (int it = 0; it < iters; it++) {                           \
            BODY BODY BODY BODY BODY BODY BODY BODY                    \
        }                                                              \
        if (threadIdx.x == 9999) sink[0] = (float)keep;                \
    }

// each BODY issues 8 independent instructions; 8 BODYs per iteration = 64 instructions
RATE_KERNEL(k_add_f64, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3; double a4 = 4; double a5 = 5; double a6 = 6; double a7 = 7; double c = 1.5; double keep = 0,
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_mul_f64, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3; double a4 = 4; double a5 = 5; double a6 = 6; double a7 = 7; double c = 1.0000001; double keep = 0,
            asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_fma_f64, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3; double a4 = 4; double a5 = 5; double a6 = 6; double a7 = 7; double c = 1.0000001; double keep = 0,
            asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_add_f32, float a0 = threadIdx.x; float a1 = 1; float a2 = 2; float a3 = 3; float a4 = 4; float a5 = 5; float a6 = 6; float a7 = 7; float c = 1.5f; float keep = 0,
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_fma_f32, float a0 = threadIdx.x; float a1 = 1; float a2 = 2; float a3 = 3; float a4 = 4; float a5 = 5; float a6 = 6; float a7 = 7; float c = 1.0001f; float keep = 0,
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_pk_fma_f32, double a0 = threadIdx.x; double a1 = 1; double a2 = 2; double a3 = 3; double a4 = 4; double a5 = 5; double a6 = 6; double a7 = 7; double c = 1.0; double keep = 0,
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_mov_dpp, int a0 = threadIdx.x; int a1 = 1; int a2 = 2; int a3 = 3; int a4 = 4; int a5 = 5; int a6 = 6; int a7 = 7; int c = 0; int keep = 0,
            asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_perm_b32, int a0 = threadIdx.x; int a1 = 1; int a2 = 2; int a3 = 3; int a4 = 4; int a5 = 5; int a6 = 6; int a7 = 7; int c = 0x03020100; int keep = 0,
            asm volatile("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_rndne_f32, float a0 = threadIdx.x; float a1 = 1; float a2 = 2; float a3 = 3; float a4 = 4; float a5 = 5; float a6 = 6; float a7 = 7; float c = 1.5f; float keep = 0,
            asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)
RATE_KERNEL(k_cvt_i32_f32, float a0 = threadIdx.x; float a1 = 1; float a2 = 2; float a3 = 3; float a4 = 4; float a5 = 5; float a6 = 6; float a7 = 7; float c = 1.5f; float keep = 0,
            asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;)

template <typename K>
static int time_rate(const char *name, K kern, float *sink, int ncu, double ops_per_instr) {
    const int iters = 2000, wgs = ncu * 8;
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
    double wave_instr = (double)wgs * 4 * iters * 64;         // wave-level instructions issued
    double per_simd = wave_instr / (ncu * 4.0);               // per SIMD
    double ns_per_instr = ms * 1e6 / per_simd;                // SIMD-time per wave instruction
    printf("%-14s %8.3f ms  %6.3f ns/wave-instr/SIMD  (= %.2f cycles @2.4GHz)  %.1f Tlane-op/s\n", name, ms, ns_per_instr,
           ns_per_instr * 2.4, wave_instr * 64 * ops_per_instr / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  memclk=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate);
    const int ncu = p.multiProcessorCount;
    float *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // ---- streaming copies ---------------------------------------------------------------------------------
    for (size_t mb : {24ul, 96ul, 400ul, 1600ul}) {
        size_t bytes = mb << 20;
        uint4 *a, *b;
        CK(hipMalloc(&a, bytes));
        CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes));
        CK(hipMemset(b, 2, bytes));
        size_t n = bytes / 16;
        int grid = ncu * 8;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipDeviceSynchronize());
        const int K = 20;
        CK(hipEventRecord(e0));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy16   %5zu MiB in + %5zu MiB out : %8.2f us/launch  %7.1f GB/s (read+write)\n", mb, mb, ms * 1e3 / K,
               2.0 * bytes * K / (ms * 1e-3) / 1e9);
        CK(hipFree(a));
        CK(hipFree(b));
    }
    // ---- the transform kernel's traffic shape (1 B in : 2 B out), no arithmetic -----------------------------
    for (int dim : {4096, 16384}) {
        size_t px = (size_t)dim * dim;
        uint8_t *img;
        uint4 *out;
        CK(hipMalloc(&img, px));
        CK(hipMalloc(&out, px * 2));
        CK(hipMemset(img, 7, px));
        int tiles_x = dim / 64, ntiles = (dim / 8) * tiles_x;
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 2; rep++)
                hipLaunchKernelGGL(shape_io, dim3((ntiles + 3) / 4), dim3(256), 0, 0, img, dim, (long)dim, tiles_x, ntiles, out, mode);
            CK(hipDeviceSynchronize());
            const int K = 20;
            CK(hipEventRecord(e0));
            for (int k = 0; k < K; k++)
                hipLaunchKernelGGL(shape_io, dim3((ntiles + 3) / 4), dim3(256), 0, 0, img, dim, (long)dim, tiles_x, ntiles, out, mode);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("shape_io %5dx%-5d mode %d (%s): %8.2f us/launch  %7.1f GB/s (3 B/px)  %7.1f Gpix/s\n", dim, dim, mode,
                   mode ? "lane=8*row+chunk" : "lane=8*block+row", ms * 1e3 / K, 3.0 * px * K / (ms * 1e-3) / 1e9,
                   px * (double)K / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(img));
        CK(hipFree(out));
    }
    // ---- issue rates ---------------------------------------------------------------------------------------
    time_rate("v_add_f64", k_add_f64, sink, ncu, 1);
    time_rate("v_mul_f64", k_mul_f64, sink, ncu, 1);
    time_rate("v_fma_f64", k_fma_f64, sink, ncu, 1);
    time_rate("v_add_f32", k_add_f32, sink, ncu, 1);
    time_rate("v_fma_f32", k_fma_f32, sink, ncu, 1);
    time_rate("v_pk_fma_f32", k_pk_fma_f32, sink, ncu, 2);
    time_rate("v_mov_dpp", k_mov_dpp, sink, ncu, 1);
    time_rate("v_perm_b32", k_perm_b32, sink, ncu, 1);
    time_rate("v_rndne_f32", k_rndne_f32, sink, ncu, 1);
    time_rate("v_cvt_i32_f32", k_cvt_i32_f32, sink, ncu, 1);
    return 0;
}
