// tic_kernels.hip - CDNA4 (gfx950) kernels of the tinyimgcodec transform stage.
//
// Replaces, on the GPU, the body of encode() (codec.py:26-43 of the reference): pad_image (utils.py:56-61),
// level shift (codec.py:29), 8x8 tiling (utils.py:13-20), 2-D DCT (utils.py:32-37), quantisation (utils.py:48-53)
// and the zig-zag gather (codec.py:32-33); and of decode() (codec.py:46-70): dequantise, inverse DCT, clip, cast.
//
// Work decomposition (wave64): one wavefront owns a strip of 8 horizontally adjacent 8x8 blocks (64x8 pixels).
// Lane l = 8*b + i serves block b of the strip; i is, in turn, the pixel row it loads, the pixel column it
// transforms (after an in-register 8x8 byte transpose across the 8 lanes of the block, DPP + v_perm), and the
// frequency row u it quantises (after a dword transpose through LDS).  Every wave-level store of coefficients is
// 1 KiB contiguous (8 blocks x 128 B, zig-zag order established in LDS).  No MFMA: the stage is a byte-in /
// int16-out streaming stencil, bounded by HBM (3 B per pixel).
//
// Two arithmetic paths, bit-identical results:
//   exact  : float64, scipy/pocketfft operation order for all 64 coefficients (tic_math.h dct8_exact).
//   hybrid : float32 AAN butterflies for 60 coefficients, accepted only outside a guard band around the .5
//            rounding ties; the 4 rational coefficients (0,0),(0,4),(4,0),(4,4) always on an exact float64
//            sub-path; any block that trips the guard is redone on the exact path inside the same wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tic_kernels.h"
#include "tic_math.h"

namespace tic {

// ---- cross-lane helpers ---------------------------------------------------------------------------------
// DPP controls (gfx9 encoding): quad_perm = p0 | p1<<2 | p2<<4 | p3<<6 ; row_shl:n = 0x100+n ; row_shr:n = 0x110+n
#define TIC_DPP_QP_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define TIC_DPP_QP_XOR2 0x4E /* quad_perm [2,3,0,1] */
#define TIC_DPP_ROW_SHL4 0x104 /* lane i reads lane i+4 */
#define TIC_DPP_ROW_SHR4 0x114 /* lane i reads lane i-4 */

__device__ __forceinline__ uint32_t perm_b32(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
    return __builtin_amdgcn_perm(hi_src, lo_src, sel);
}

// 8x8 byte transpose across the 8 lanes of a block.  In: lane i holds row i as (lo = px 0..3, hi = px 4..7).
// Out: lane i holds column i as (lo = rows 0..3, hi = rows 4..7).  10 VALU ops, no LDS.
__device__ __forceinline__ void transpose8x8_bytes(uint32_t &lo, uint32_t &hi, int i) {
    // stage A: exchange 4x4 byte blocks between lanes i and i^4
    uint32_t nhi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)lo, TIC_DPP_ROW_SHL4, 0xf, 0x5, false);
    uint32_t nlo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)hi, TIC_DPP_ROW_SHR4, 0xf, 0xa, false);
    lo = nlo;
    hi = nhi;
    // stage B: exchange 2x2 byte blocks between lanes i and i^2
    uint32_t selB = (i & 2) ? 0x03020706u : 0x05040100u;
    uint32_t plo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, TIC_DPP_QP_XOR2, 0xf, 0xf, true);
    uint32_t phi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, TIC_DPP_QP_XOR2, 0xf, 0xf, true);
    lo = perm_b32(plo, lo, selB);
    hi = perm_b32(phi, hi, selB);
    // stage C: exchange single bytes between lanes i and i^1
    uint32_t selC = (i & 1) ? 0x03070105u : 0x06020400u;
    plo = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, TIC_DPP_QP_XOR1, 0xf, 0xf, true);
    phi = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, TIC_DPP_QP_XOR1, 0xf, 0xf, true);
    lo = perm_b32(plo, lo, selC);
    hi = perm_b32(phi, hi, selC);
}

// Reference implementation of the same transpose with ds_bpermute shuffles (used by the self-test kernel).
__device__ __forceinline__ void transpose8x8_bytes_shfl(uint32_t &lo, uint32_t &hi, int i) {
    uint32_t row[2] = {lo, hi};
    uint32_t col_lo = 0, col_hi = 0;
    int base = (threadIdx.x & 63) & ~7;
    for (int r = 0; r < 8; r++) {
        uint32_t l = __shfl(row[0], base + r, 64), h = __shfl(row[1], base + r, 64);
        uint32_t w = (i < 4) ? l : h;
        uint32_t byte = (w >> (8 * (i & 3))) & 0xffu;
        if (r < 4)
            col_lo |= byte << (8 * r);
        else
            col_hi |= byte << (8 * (r - 4));
    }
    lo = col_lo;
    hi = col_hi;
}

// Wave-private LDS hand-off: the LDS pipeline executes one wave's DS instructions in order, so a compiler-level
// fence is all that is needed between a write phase and a read phase of the same wave.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// np.pad(..., "reflect") index (utils.py:56-61): no edge repeat, period 2(n-1); n == 1 degenerates to edge.
__device__ __forceinline__ int reflect_index(int i, int n) {
    if (i < n) return i;
    if (n >= 8) return 2 * (n - 1) - i; // padding is at most 7 samples: a single reflection
    if (n == 1) return 0;
    int p = 2 * (n - 1), j = i;
    while (j >= p) j -= p; // tiny axes only
    return j < n ? j : p - j;
}

constexpr int kWavesPerWG = 4;
constexpr int kLdsStrideDw = 72;    // dwords per block in the transpose buffer (64 + 8 pad: conflict-free ds_write_b32)
constexpr int kZzStrideB = 144;     // bytes per block in the zig-zag staging buffer (128 + 16 pad)
constexpr int kLdsWaveBytes = 8 * kLdsStrideDw * 4; // 2304 B per wave (>= 8*144)

struct Strip {
    int by, bx;   // block coordinates of this lane's block
    bool valid;   // lane's block exists
    size_t oblk;  // raster index of the block
};

// Loads row i of the lane's block as 8 bytes (reflect padding at the right/bottom borders).
__device__ __forceinline__ void load_block_row(const uint8_t *__restrict__ img, int h, int w, long stride, bool aligned8,
                                               const Strip &s, int i, uint32_t &lo, uint32_t &hi) {
    lo = 0;
    hi = 0;
    if (!s.valid) return;
    int y = reflect_index(s.by * 8 + i, h);
    int x0 = s.bx * 8;
    const uint8_t *p = img + (long)y * stride + x0;
    if (aligned8 && x0 + 8 <= w) {
        uint2 v = *reinterpret_cast<const uint2 *>(p);
        lo = v.x;
        hi = v.y;
    } else {
        const uint8_t *rowp = img + (long)y * stride;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t b = rowp[reflect_index(x0 + k, w)];
            if (k < 4)
                lo |= b << (8 * k);
            else
                hi |= b << (8 * (k - 4));
        }
    }
}

// Transposes 8 dwords per lane across the 8 lanes of each block through the wave's LDS buffer:
// in: lane (b,c) holds v[u] = M[u][c]; out: lane (b,u) holds v[c] = M[u][c].
__device__ __forceinline__ void transpose8x8_dwords(uint32_t *lds, int b, int i, uint32_t v[8]) {
    uint32_t *blk = lds + b * kLdsStrideDw;
#pragma unroll
    for (int u = 0; u < 8; u++) blk[u * 8 + i] = v[u];
    wave_lds_fence();
    const uint4 *rp = reinterpret_cast<const uint4 *>(blk + i * 8);
    uint4 a = rp[0], c = rp[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    wave_lds_fence();
}

// Exact path for the lane's block: float64, pocketfft order, true IEEE division, round-half-even.
// colLo/colHi: the lane's pixel column (8 bytes).  Out: q[v] = quantised coefficient (u = i, v) as int.
__device__ __forceinline__ void exact_block(uint32_t colLo, uint32_t colHi, uint32_t *lds, int b, int i,
                                         const DctqConsts *__restrict__ C, int q[8]) {
    double c[8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        c[r] = (double)((int)((colLo >> (8 * r)) & 0xffu) - 128);
        c[r + 4] = (double)((int)((colHi >> (8 * r)) & 0xffu) - 128);
    }
    dct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -2: down the column
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)__double2loint(c[k]);
    transpose8x8_dwords(lds, b, i, w);
    uint32_t wh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) wh[k] = (uint32_t)__double2hiint(c[k]);
    transpose8x8_dwords(lds, b, i, wh);
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = __hiloint2double((int)wh[k], (int)w[k]);
    dct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -1: along frequency row u = i
    const double *div = C->div + i * 8;
#pragma unroll
    for (int v = 0; v < 8; v++) q[v] = (int)rint(c[v] / div[v]); // np.round(X / div): IEEE divide, half-even
}

// Writes the lane's 8 coefficients (natural positions i*8+v) into zig-zag order in LDS, then each lane stores
// 16 bytes: a wave writes its 8 blocks as one contiguous 1 KiB segment.
__device__ __forceinline__ void store_zigzag(uint32_t *lds, int b, int i, const uint16_t zz[8], const int q[8],
                                             int16_t *__restrict__ out, const Strip &s) {
    char *blk = reinterpret_cast<char *>(lds) + b * kZzStrideB;
#pragma unroll
    for (int v = 0; v < 8; v++) *reinterpret_cast<int16_t *>(blk + zz[v]) = (int16_t)q[v];
    wave_lds_fence();
    uint4 val = *reinterpret_cast<const uint4 *>(blk + i * 16);
    wave_lds_fence();
    if (s.valid) *reinterpret_cast<uint4 *>(out + s.oblk * 64 + i * 8) = val;
}

__device__ __forceinline__ Strip make_strip(int tile, int ntiles, int tiles_x, int bw, int b) {
    Strip s;
    int ty = tile / tiles_x;
    int tx = tile - ty * tiles_x;
    s.by = ty;
    s.bx = tx * 8 + b;
    s.valid = (tile < ntiles) && (s.bx < bw);
    s.oblk = (size_t)ty * bw + s.bx;
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 1: exact path for every block.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64) void dctq_exact_kernel(DctqArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerWG][kLdsWaveBytes / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    const int tile = blockIdx.x * kWavesPerWG + wave;
    Strip s = make_strip(tile, a.ntiles, a.tiles_x, a.bw, b);
    uint32_t lo, hi;
    load_block_row(a.img, a.h, a.w, a.stride, a.aligned8, s, i, lo, hi);
    transpose8x8_bytes(lo, hi, i);
    int q[8];
    exact_block(lo, hi, lds, b, i, a.consts, q);
    const uint4 zzv = *reinterpret_cast<const uint4 *>(a.consts->zzofs + i * 8);
    uint16_t zz[8] = {(uint16_t)zzv.x, (uint16_t)(zzv.x >> 16), (uint16_t)zzv.y, (uint16_t)(zzv.y >> 16),
                      (uint16_t)zzv.z, (uint16_t)(zzv.z >> 16), (uint16_t)zzv.w, (uint16_t)(zzv.w >> 16)};
    store_zigzag(lds, b, i, zz, q, a.out, s);
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 2: hybrid path.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64) void dctq_hybrid_kernel(DctqArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerWG][kLdsWaveBytes / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    const int tile = blockIdx.x * kWavesPerWG + wave;
    const DctqConsts *__restrict__ C = a.consts;
    Strip s = make_strip(tile, a.ntiles, a.tiles_x, a.bw, b);

    uint32_t lo, hi;
    load_block_row(a.img, a.h, a.w, a.stride, a.aligned8, s, i, lo, hi);

    // per-lane constants for frequency row u = i (L2-resident, 80 bytes per lane)
    const float4 m0 = *reinterpret_cast<const float4 *>(C->mul + i * 8);
    const float4 m1 = *reinterpret_cast<const float4 *>(C->mul + i * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4 *>(C->thr + i * 8);
    const float4 h1 = *reinterpret_cast<const float4 *>(C->thr + i * 8 + 4);
    const uint4 zzv = *reinterpret_cast<const uint4 *>(C->zzofs + i * 8);

    transpose8x8_bytes(lo, hi, i); // lane i now holds pixel column i

    // ---- column pass (axis -2), float32 AAN on raw 0..255 pixels; level shift folded into output 0 ------
    float d0 = (float)(lo & 0xffu), d1 = (float)((lo >> 8) & 0xffu), d2 = (float)((lo >> 16) & 0xffu),
          d3 = (float)(lo >> 24);
    float d4 = (float)(hi & 0xffu), d5 = (float)((hi >> 8) & 0xffu), d6 = (float)((hi >> 16) & 0xffu),
          d7 = (float)(hi >> 24);
    dct8_aan(d0, d1, d2, d3, d4, d5, d6, d7);
    d0 -= 1024.0f; // sum of 8 pixels minus 8*128: exact integer

    uint32_t y[8] = {__float_as_uint(d0), __float_as_uint(d1), __float_as_uint(d2), __float_as_uint(d3),
                     __float_as_uint(d4), __float_as_uint(d5), __float_as_uint(d6), __float_as_uint(d7)};
    transpose8x8_dwords(lds, b, i, y); // lane i now holds frequency row u = i: Y[u][0..7]
    float e0 = __uint_as_float(y[0]), e1 = __uint_as_float(y[1]), e2 = __uint_as_float(y[2]),
          e3 = __uint_as_float(y[3]), e4 = __uint_as_float(y[4]), e5 = __uint_as_float(y[5]),
          e6 = __uint_as_float(y[6]), e7 = __uint_as_float(y[7]);

    // ---- rational coefficients (u,v) in {0,4}x{0,4}: exact float64 sub-path on lanes u = 0 and u = 4 ------
    // For integer pixels the column pass outputs 0 and 4 are (integer sum) * constant, one rounding each, and
    // the row pass outputs 0 and 4 need only 8 additions in pocketfft's order (consequence 2).
    int q0x = 0, q4x = 0;
    if ((i & 3) == 0) {
#pragma clang fp contract(off)
        const double K = (i == 0) ? (kSq2h * 0.5) : (kTW3 * 0.5);
        double y0 = (double)e0 * K, y1 = (double)e1 * K, y2 = (double)e2 * K, y3 = (double)e3 * K;
        double y4 = (double)e4 * K, y5 = (double)e5 * K, y6 = (double)e6 * K, y7 = (double)e7 * K;
        double p07 = y0 + y7, p34 = y3 + y4, p12 = y1 + y2, p56 = y5 + y6;
        double A = p07 + p34, B = p12 + p56;
        double E0 = A + B, E4 = A - B;
        double X0 = E0 * (kSq2h * 0.5), X4 = E4 * (kTW3 * 0.5);
        const double dv0 = C->div[i * 8], dv4 = C->div[i * 8 + 4];
        const double rd0 = C->rdiv[i * 8], rd4 = C->rdiv[i * 8 + 4];
        double t0 = X0 * rd0, t4 = X4 * rd4;
        double r0 = rint(t0), r4 = rint(t4);
        // the reciprocal product is within ~1e-12 of X/div: only a quotient that close to a tie needs the divide
        if (fabs(fabs(t0 - r0) - 0.5) < 1e-9) r0 = rint(X0 / dv0);
        if (fabs(fabs(t4 - r4) - 0.5) < 1e-9) r4 = rint(X4 / dv4);
        q0x = (int)r0;
        q4x = (int)r4;
    }

    // ---- row pass (axis -1), float32 AAN, quantise with guard band ----------------------------------------
    dct8_aan(e0, e1, e2, e3, e4, e5, e6, e7);
    int q[8];
    bool trip = false;
    {
        float t, r;
        t = e0 * m0.x; r = rintf(t); trip |= fabsf(t - r) > h0.x; q[0] = (int)r;
        t = e1 * m0.y; r = rintf(t); trip |= fabsf(t - r) > h0.y; q[1] = (int)r;
        t = e2 * m0.z; r = rintf(t); trip |= fabsf(t - r) > h0.z; q[2] = (int)r;
        t = e3 * m0.w; r = rintf(t); trip |= fabsf(t - r) > h0.w; q[3] = (int)r;
        t = e4 * m1.x; r = rintf(t); trip |= fabsf(t - r) > h1.x; q[4] = (int)r;
        t = e5 * m1.y; r = rintf(t); trip |= fabsf(t - r) > h1.y; q[5] = (int)r;
        t = e6 * m1.z; r = rintf(t); trip |= fabsf(t - r) > h1.z; q[6] = (int)r;
        t = e7 * m1.w; r = rintf(t); trip |= fabsf(t - r) > h1.w; q[7] = (int)r;
    }
    if ((i & 3) == 0) {
        q[0] = q0x;
        q[4] = q4x;
    }

    // ---- guard tripped somewhere in the wave: redo those blocks on the exact path ---------------------------
    const unsigned long long tripmask = __ballot(trip && s.valid);
    if (tripmask != 0ull) { // wave-uniform
        int qe[8];
        exact_block(lo, hi, lds, b, i, C, qe);
        const bool mine = ((tripmask >> (b * 8)) & 0xffull) != 0ull;
        if (mine) {
#pragma unroll
            for (int v = 0; v < 8; v++) q[v] = qe[v];
        }
        if (a.fallback_count != nullptr && lane == 0) {
            unsigned long long m = tripmask, n = 0;
            for (int k = 0; k < 8; k++) n += ((m >> (8 * k)) & 0xffull) ? 1ull : 0ull;
            atomicAdd(a.fallback_count, n);
        }
    }

    uint16_t zz[8] = {(uint16_t)zzv.x, (uint16_t)(zzv.x >> 16), (uint16_t)zzv.y, (uint16_t)(zzv.y >> 16),
                      (uint16_t)zzv.z, (uint16_t)(zzv.z >> 16), (uint16_t)zzv.w, (uint16_t)(zzv.w >> 16)};
    store_zigzag(lds, b, i, zz, q, a.out, s);
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 3: decode side - dequantise (utils.py:52), inverse DCT (utils.py:40-45, exact order), +128, clip,
// truncating cast (codec.py:68-70), crop.  Input: int16 [N][64] zig-zag, DC already integrated (np.cumsum).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64) void idct_kernel(IdctArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerWG][kLdsWaveBytes / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    const int tile = blockIdx.x * kWavesPerWG + wave;
    const DctqConsts *__restrict__ C = a.consts;
    Strip s = make_strip(tile, a.ntiles, a.tiles_x, a.bw, b);

    // lane i of block b loads zig-zag entries 8i..8i+7 (16 B), scatters them to natural order in LDS
    uint4 val = make_uint4(0, 0, 0, 0);
    if (s.valid) val = *reinterpret_cast<const uint4 *>(a.coeffs + s.oblk * 64 + i * 8);
    char *blk = reinterpret_cast<char *>(lds) + b * kZzStrideB;
    {
        uint32_t wv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int nat = C->zznat[i * 8 + k]; // natural index of scan position 8i+k
            int16_t cv = (int16_t)((wv[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            *reinterpret_cast<int16_t *>(blk + nat * 2) = cv;
        }
    }
    wave_lds_fence();
    // lane i takes column v = i of the natural 8x8 coefficient matrix: X[u][i], u = 0..7
    double c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        int16_t cv = *reinterpret_cast<const int16_t *>(blk + (u * 8 + i) * 2);
        c[u] = (double)cv * C->div[u * 8 + i]; // coeffs * (Q*factor/100)
    }
    wave_lds_fence();
    idct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -2
    uint32_t w[8], wh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)__double2loint(c[k]);
    transpose8x8_dwords(lds, b, i, w);
#pragma unroll
    for (int k = 0; k < 8; k++) wh[k] = (uint32_t)__double2hiint(c[k]);
    transpose8x8_dwords(lds, b, i, wh);
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = __hiloint2double((int)wh[k], (int)w[k]);
    idct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -1: lane i holds pixel row i
    uint32_t px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        double v = c[k] + 128.0;
        v = v < 0.0 ? 0.0 : v;
        v = v > 255.0 ? 255.0 : v;
        px[k] = (uint32_t)(int)v; // truncation toward zero, as astype(np.uint8) on a clipped value
    }
    if (!s.valid) return;
    int y = s.by * 8 + i;
    if (y >= a.h) return;
    int x0 = s.bx * 8;
    uint8_t *p = a.out + (long)y * a.stride + x0;
    if (a.aligned8 && x0 + 8 <= a.w) {
        uint2 o;
        o.x = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        o.y = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
        *reinterpret_cast<uint2 *>(p) = o;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (x0 + k < a.w) p[k] = (uint8_t)px[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Self-test kernel: checks the DPP byte transpose against the shuffle formulation on arbitrary data.
// ---------------------------------------------------------------------------------------------------------
__global__ void selftest_transpose_kernel(const uint2 *in, uint2 *out_dpp, uint2 *out_ref) {
    int lane = threadIdx.x & 63, i = lane & 7;
    uint2 v = in[blockIdx.x * blockDim.x + threadIdx.x];
    uint32_t lo = v.x, hi = v.y;
    transpose8x8_bytes(lo, hi, i);
    out_dpp[blockIdx.x * blockDim.x + threadIdx.x] = make_uint2(lo, hi);
    lo = v.x;
    hi = v.y;
    transpose8x8_bytes_shfl(lo, hi, i);
    out_ref[blockIdx.x * blockDim.x + threadIdx.x] = make_uint2(lo, hi);
}

// ---- launchers ---------------------------------------------------------------------------------------------
static inline int grid_for(int ntiles) { return (ntiles + kWavesPerWG - 1) / kWavesPerWG; }

hipError_t launch_dctq(const DctqArgs &a, int variant, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    dim3 grid(grid_for(a.ntiles)), block(kWavesPerWG * 64);
    if (variant == 1)
        hipLaunchKernelGGL(dctq_exact_kernel, grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL(dctq_hybrid_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_idct(const IdctArgs &a, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    dim3 grid(grid_for(a.ntiles)), block(kWavesPerWG * 64);
    hipLaunchKernelGGL(idct_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_selftest_transpose(const void *in, void *out_dpp, void *out_ref, int nthreads, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_transpose_kernel, dim3(nthreads / 256), dim3(256), 0, stream, (const uint2 *)in,
                       (uint2 *)out_dpp, (uint2 *)out_ref);
    return hipGetLastError();
}

} // namespace tic
