// microbench8: issue rate of the float64 vector instructions the decoder's inverse transform is made of (gfx950), per SIMD.
// Each wave runs 8 independent chains of one instruction, `iters` times; W waves per SIMD.  Prints cycles per wave-instruction as the SIMD sees it
// (at 2.4 GHz nominal; the clock under load may be lower, so compare the rows with each other).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double a[8];
    int ia[8];
    for (int j = 0; j < 8; j++) { a[j] = seed + threadIdx.x * 1e-3 + j; ia[j] = (int)threadIdx.x + j; }
    const double c1 = seed * 1.0000001, c2 = seed * 0.5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
            if (OP == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(ia[j]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[j]) : "v"(a[j]));
            if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, -2" : "+v"(a[j]));
            if (OP == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
            if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ia[j]) : "v"(ia[(j + 1) & 7]));
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[j]) : "v"(ia[(j + 1) & 7]) : );
            if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(c1)); // ONE dependent chain
            if (OP == 10) asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(a[j]));
        }
    }
    double s = 0;
    for (int j = 0; j < 8; j++) s += a[j] + ia[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_ldexp_f64", "v_max_f64", "v_add_f32", "v_cndmask_b32", "v_add_f64, one chain", "v_mul_f64 by 0.5"};
    double *d;
    CHK(hipMalloc(&d, 256 * 16 * 256 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int W = 1; W <= 4; W += (W == 1 ? 1 : 1)) {
        for (int op = 0; op <= 10; op++) {
            void (*fn)(double *, int, double) = nullptr;
            switch (op) { case 0: fn = k<0>; break; case 1: fn = k<1>; break; case 2: fn = k<2>; break; case 3: fn = k<3>; break; case 4: fn = k<4>; break; case 5: fn = k<5>; break;
                          case 6: fn = k<6>; break; case 7: fn = k<7>; break; case 8: fn = k<8>; break; case 9: fn = k<9>; break; case 10: fn = k<10>; break; }
            const int grid = 256 * W; // 256-thread workgroups: one wave per SIMD each
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * W);
            printf("W=%d %-22s %8.1f us  %5.2f cycles per wave-instruction (2.4 GHz)\n", W, names[op], ms * 1e3, cyc);
        }
    }
    return 0;
}
