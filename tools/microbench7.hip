// microbench7.hip - floors of the transform stage's traffic shape (1 B read : 2 B written per pixel, 4096^2) on COLD data:
// 12 rotating buffer pairs (604 MB > the 256 MiB Infinity Cache) against one pair replayed (warm).  What the strip kernel
// could reach at most when nothing but the memory stream is left.  Build: make -C tools bin/microbench7
//   shape : one strip (64x8 px) per wave, 8-byte loads (lane = 8*row + block), 16-byte stores, no loop
//   wide  : two strips per wave, 16-byte loads (lane = 8*row + pair), two 16-byte stores
//   rd    : the loads only (one dword written per wave)       wr : the stores only
//   pers  : persistent grid (6 workgroups per CU), each wave loops over strips with the loads two strips ahead
// Store policies: 0 plain, 1 nt, 2 sc1 (write-through).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int ST>
__device__ __forceinline__ void store16(void *p, u32x4 d) {
    if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(d) : "memory");
    else if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(d) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(d) : "memory");
}

template <int ST, int MODE> // MODE 0 shape, 1 rd, 2 wr
__global__ __launch_bounds__(256) void k_shape(const uint8_t *__restrict__ img, int w, int tiles_x, int ntiles, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int r = lane >> 3, b = lane & 7;
    u32x2 v = {(uint32_t)tile, (uint32_t)lane};
    if (MODE != 2) v = *reinterpret_cast<const u32x2 *>(img + (long)(ty * 8 + r) * w + (tx * 8 + b) * 8);
    u32x4 o = {v.x, v.y, v.x ^ 0x80808080u, v.y ^ 0x80808080u};
    const size_t oblk = (size_t)ty * (w / 8) + tx * 8;
    if (MODE != 1) store16<ST>(out + oblk * 128 + lane * 16, o);
    else if ((v.x ^ v.y) == 0x12345678u && lane == 0) *reinterpret_cast<uint32_t *>(out + oblk * 128) = v.x;
}

template <int ST>
__global__ __launch_bounds__(256) void k_wide(const uint8_t *__restrict__ img, int w, int pairs_x, int npairs, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave; // pair of strips: 128 px x 8 rows
    if (t >= npairs) return;
    const int ty = t / pairs_x, tx = t - ty * pairs_x;
    const int r = lane >> 3, c = lane & 7; // 16-byte chunk c of pixel row r
    const u32x4 v = *reinterpret_cast<const u32x4 *>(img + (long)(ty * 8 + r) * w + tx * 128 + c * 16);
    const size_t oblk = (size_t)ty * (w / 8) + tx * 16;
    u32x4 o0 = {v.x, v.y, v.x ^ 0x80808080u, v.y ^ 0x80808080u}, o1 = {v.z, v.w, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
    store16<ST>(out + oblk * 128 + lane * 16, o0);
    store16<ST>(out + (oblk + 8) * 128 + lane * 16, o1);
}

// persistent: gridDim.x workgroups, wave walks strips tile0, tile0 + nwaves, ...; loads two strips ahead, counted waits
template <int ST, int WIDE>
__global__ __launch_bounds__(256, 6) void k_pers(const uint8_t *__restrict__ img, int w, int tiles_x, int ntiles, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    const int r = lane >> 3, b = lane & 7;
    auto addr = [&](int t) {
        t = t < ntiles ? t : ntiles - 1;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        return WIDE ? img + (long)(ty * 8 + r) * w + tx * 128 + b * 16 : img + (long)(ty * 8 + r) * w + (tx * 8 + b) * 8;
    };
    auto oaddr = [&](int t) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        return out + ((size_t)ty * (w / 8) + tx * (WIDE ? 16 : 8)) * 128 + lane * 16;
    };
    // The destination registers of loads in flight must be invisible to the compiler: a value the compiler believes to exist
    // from the asm statement on gets copied (phi moves, coalescing with the store's register quad) before it has landed, and a
    // landing load overwrites whatever the allocator has meanwhile put into its register (this microbenchmark faulted that way
    // with "=v" outputs).  Here the loads land in accumulator registers the compiler never allocates (a0.., declared as clobbers)
    // and become visible through v_accvgpr_read behind the counted wait.
#define LOADN(A, T) asm volatile("global_load_dwordx2 a[" #A "], %0, off" : : "v"(addr(T)) : "memory", "a0", "a1", "a2", "a3", "a4", "a5")
#define TAKEN(V, A0, A1, N) asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 : "=v"(V.x), "=v"(V.y) : : "memory")
#define LOADW(A, T) asm volatile("global_load_dwordx4 a[" #A "], %0, off" : : "v"(addr(T)) : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11")
#define TAKEW(V, A0, A1, A2, A3, N) asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\tv_accvgpr_read_b32 %3, a" #A3 : "=v"(V.x), "=v"(V.y), "=v"(V.z), "=v"(V.w) : : "memory")
    if (WIDE) {
        u32x4 v;
        auto body = [&](int t) {
            u32x4 o0 = {v.x, v.y, v.x ^ 0x80808080u, v.y ^ 0x80808080u}, o1 = {v.z, v.w, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
            uint8_t *o = oaddr(t);
            store16<ST>(o, o0);
            store16<ST>(o + 1024, o1);
        };
        LOADW(0:3, tile);
        LOADW(4:7, tile + nw);
        // order of issue: L0 L1 | L2 W0 S0 S0' | L3 W1 S1 S1' | ...: the operations younger than L(j) at its wait are 2, 4, then 6
        do {
            if (tile >= ntiles) break;
            LOADW(8:11, tile + 2 * nw); TAKEW(v, 0, 1, 2, 3, 2); body(tile); tile += nw;
            if (tile >= ntiles) break;
            LOADW(0:3, tile + 2 * nw); TAKEW(v, 4, 5, 6, 7, 4); body(tile); tile += nw;
            while (tile < ntiles) {
                LOADW(4:7, tile + 2 * nw); TAKEW(v, 8, 9, 10, 11, 6); body(tile); tile += nw;
                if (tile >= ntiles) break;
                LOADW(8:11, tile + 2 * nw); TAKEW(v, 0, 1, 2, 3, 6); body(tile); tile += nw;
                if (tile >= ntiles) break;
                LOADW(0:3, tile + 2 * nw); TAKEW(v, 4, 5, 6, 7, 6); body(tile); tile += nw;
            }
        } while (0);
    } else {
        u32x2 v;
        auto body = [&](int t) {
            u32x4 o = {v.x, v.y, v.x ^ 0x80808080u, v.y ^ 0x80808080u};
            store16<ST>(oaddr(t), o);
        };
        LOADN(0:1, tile);
        LOADN(2:3, tile + nw);
        do { // younger than L(j) at its wait: 2, 3, then 4
            if (tile >= ntiles) break;
            LOADN(4:5, tile + 2 * nw); TAKEN(v, 0, 1, 2); body(tile); tile += nw;
            if (tile >= ntiles) break;
            LOADN(0:1, tile + 2 * nw); TAKEN(v, 2, 3, 3); body(tile); tile += nw;
            while (tile < ntiles) {
                LOADN(2:3, tile + 2 * nw); TAKEN(v, 4, 5, 4); body(tile); tile += nw;
                if (tile >= ntiles) break;
                LOADN(4:5, tile + 2 * nw); TAKEN(v, 0, 1, 4); body(tile); tile += nw;
                if (tile >= ntiles) break;
                LOADN(0:1, tile + 2 * nw); TAKEN(v, 2, 3, 4); body(tile); tile += nw;
            }
        } while (0);
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11");
}

int main(int argc, char **argv) {
    const int dim = 4096, NP = 12;
    const size_t px = (size_t)dim * dim;
    std::vector<uint8_t *> img(NP), out(NP);
    std::vector<uint8_t> h(px);
    for (size_t i = 0; i < px; i++) h[i] = (uint8_t)(i * 2654435761u >> 13);
    for (int k = 0; k < NP; k++) {
        CK(hipMalloc((void **)&img[k], px));
        CK(hipMalloc((void **)&out[k], px * 2));
        CK(hipMemcpy(img[k], h.data(), px, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int tiles_x = dim / 64, ntiles = tiles_x * (dim / 8);
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    auto run = [&](const char *name, auto launch) {
        for (int cold = 0; cold < 2; cold++) {
            const int K = 240;
            for (int k = 0; k < 60; k++) launch(cold ? k % NP : 0);
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                for (int k = 0; k < K; k++) launch(cold ? k % NP : 0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-34s %s %8.2f us/launch  %7.1f GB/s (3 B/px)\n", name, cold ? "cold" : "warm", best * 1e3 / K, 3.0 * px * K / (best * 1e-3) / 1e9);
        }
        fflush(stdout);
    };
    const dim3 g1((ntiles + 3) / 4), g2((ntiles / 2 + 3) / 4), gp(cus * 6), blk(256);
#define SHAPE(ST, MODE, NAME) run(NAME, [&](int p) { hipLaunchKernelGGL((k_shape<ST, MODE>), g1, blk, 0, 0, img[p], dim, tiles_x, ntiles, out[p]); })
    SHAPE(0, 0, "shape, plain stores");
    SHAPE(1, 0, "shape, nt stores");
    SHAPE(2, 0, "shape, sc1 stores");
    SHAPE(0, 1, "rd only (16.8 MB)");
    SHAPE(0, 2, "wr only, plain (33.5 MB)");
    SHAPE(1, 2, "wr only, nt");
    SHAPE(2, 2, "wr only, sc1");
#define WIDE(ST, NAME) run(NAME, [&](int p) { hipLaunchKernelGGL((k_wide<ST>), g2, blk, 0, 0, img[p], dim, tiles_x / 2, ntiles / 2, out[p]); })
    WIDE(0, "wide (16-B loads), plain");
    WIDE(1, "wide (16-B loads), nt");
    WIDE(2, "wide (16-B loads), sc1");
#define PERS(ST, W, NAME) run(NAME, [&](int p) { hipLaunchKernelGGL((k_pers<ST, W>), gp, blk, 0, 0, img[p], dim, W ? tiles_x / 2 : tiles_x, W ? ntiles / 2 : ntiles, out[p]); })
    PERS(0, 0, "persistent, plain");
    PERS(1, 0, "persistent, nt");
    PERS(2, 0, "persistent, sc1");
    PERS(0, 1, "persistent wide, plain");
    PERS(2, 1, "persistent wide, sc1");
    run("empty kernel (launch floor)", [&](int p) { hipLaunchKernelGGL((k_shape<0, 0>), dim3(1), blk, 0, 0, img[p], dim, tiles_x, 0, out[p]); });
    return 0;
}
