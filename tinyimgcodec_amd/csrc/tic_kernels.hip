// tic_kernels.hip - CDNA4 (gfx950) kernels of the tinyimgcodec transform stage.
//
// Replaces, on the GPU, the body of encode() (codec.py:26-43 of the reference): pad_image (utils.py:56-61),
// level shift (codec.py:29), 8x8 tiling (utils.py:13-20), 2-D DCT (utils.py:32-37), quantisation (utils.py:48-53)
// and the zig-zag gather (codec.py:32-33); and of decode() (codec.py:46-70): dequantise, inverse DCT, clip, cast.
//
// Work decomposition (wave64): one wavefront owns a strip of 8 horizontally adjacent 8x8 blocks (64x8 pixels); 8 lanes serve a block.
// Every wave-level store of coefficients is 1 KiB contiguous (8 blocks x 128 B, zig-zag order established in LDS).  No MFMA: the
// stage is a byte-in / int16-out streaming stencil, bounded by HBM (3 B per pixel).
//
// Two arithmetic paths, bit-identical results:
//   exact : float64, scipy/pocketfft operation order for all 64 coefficients (tic_math.h dct8_exact): dctq_exact_kernel - padding
//           strips, the reference kernel of the tests, the last resort of the production kernel.
//   strip : the production kernel (dctq_strip_kernel<kCols>): float32 AAN butterflies, every rounding accepted only outside a rigorous
//           guard band around the .5 ties.  Ties of the four rational coefficients are settled inside the loop in the reference's
//           float64 operation order (columns first: from the fast path's own exact column sums, rational_quad; rows first: from the
//           pixels, rational_slim); blocks with an irrational coefficient inside its band join a wave-local batch that is settled
//           behind the loop in float64 (wave_redo_block / second_level_8), and only what that cannot decide takes the exact order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "tic_hooks.h"
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

constexpr int kWavesPerWG = 4;      // (8-wave workgroups, three per CU, measured level: profiles/r02_ab_wg8.txt)
constexpr int kLdsStrideDw = 68;    // dwords per block in the transpose buffer (64 + 4 pad; bank analysis in DESIGN.md)
constexpr int kZzStrideB = 144;     // bytes per block in the zig-zag staging buffer (128 + 16 pad)
constexpr int kLdsWaveBytes = 8 * kLdsStrideDw * 4; // 2176 B per wave (>= 8*144)

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
    for (int v = 0; v < 8; v++) {
        q[v] = (int)rint(c[v] / div[v]); // np.round(X / div): IEEE divide, half-even
        __builtin_amdgcn_sched_barrier(0); // one division at a time: keeps the register footprint of the 8 expansions small
    }
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
    a.img += (long)blockIdx.y * a.frame_stride_in; // batch: one grid row per frame
    a.out = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(a.out) + (long)blockIdx.y * a.frame_stride_out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    int tile = blockIdx.x * kWavesPerWG + wave;
    int ntiles = a.ntiles;
    if (a.rem_mode) { // enumerate only the strips outside the hybrid kernel's rectangle
        const int bh = a.ntiles / a.tiles_x, rw = a.tiles_x - a.fast_tx;
        const int n_right = bh * rw, n_bottom = (bh - a.fast_ty) * a.fast_tx;
        ntiles = 0;
        if (tile < n_right) {
            tile = (tile / rw) * a.tiles_x + a.fast_tx + tile % rw;
            ntiles = a.ntiles;
        } else if (tile < n_right + n_bottom) {
            const int k = tile - n_right;
            tile = (a.fast_ty + k / a.fast_tx) * a.tiles_x + k % a.fast_tx;
            ntiles = a.ntiles;
        }
    }
    Strip s = make_strip(tile, ntiles, a.tiles_x, a.bw, b);
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
// Kernel 2: the production kernel (dctq_strip_kernel below): float32 fast path behind a guard band, rational ties settled inside the
// loop from the fast path's own exact column sums, wave-local batch pass for the irrational trips.
//
// Persistent waves, each walking its strips (strip = 8 horizontally adjacent blocks) in the order the launcher chose:
// team schedule when the grid fits the chip at once, chunked schedule for larger grids (launch_dctq).
// Main loop, per strip (unrolled x3, the landing registers of the pixel loads rotate by name):
//   load   : lane 8*r + b reads the 8 bytes of pixel row r of block b -> every 8 lanes read 64 contiguous bytes; inline
//            assembly + hand-counted vmcnt keep two strips in flight.
//   xpose 0: the 8 bytes go to wave-private LDS straight out of the landing registers (one ds_write_b64 behind the counted wait);
//            lane 8*b + c reads pixel column c of block b back with eight ds_read_u8 - a byte transpose without a vector instruction.
//   pass 1 : float32 AAN DOWN the pixel column held by the lane - the reference's pass order (utils.py:33-37: axis -2 first).  Outputs
//            0 and 4 are the column's sum and alternating sum: exact integers, the very numbers the reference's float64 row pass
//            starts from for the four rational coefficients (0,0) (0,4) (4,0) (4,4).  The level shift is folded into output 0.
//   xpose 1: 8x8 dword transpose per block through wave-private LDS (conflict-free slot layout); lane 8*b + u then holds frequency
//            row u.
//   pass 2 : float32 AAN along that row; quantise with the magic-number rounding trick; the guard test is the largest distance
//            to the rounded value per lane against three per-row thresholds.
//   ties   : if one of the four rational coefficients tripped anywhere in the strip (an exact .5 tie: 2.2 % of random blocks at q = 50,
//            most blocks of posterised, two-level and flat content), ALL 32 rational coefficients of the strip are recomputed in the
//            reference's float64 operation order by the 64 lanes at once (rational_quad: column sums re-read from xpose 1's buffer,
//            two DPP exchanges, one division) and written over the fast path's values in the zig-zag staging.  ~40 instructions;
//            rounds 2-4 took the pixels through a byte transpose, v_sad_u8 and three DPP butterflies per tie block after the loop.
//   store  : int16 results scattered to zig-zag order in LDS, read back 16 B per lane, 1 KiB contiguous per wave.
//   trips  : blocks in which one of the 60 irrational coefficients tripped its guard band (0.2 % of random blocks at q = 50, 0.9 % at
//            q = 90) join the wave's batch and are settled after the loop in float64.
// ---------------------------------------------------------------------------------------------------------
constexpr int kTWaveBytes = kLdsWaveBytes;        // 2176 B: transpose buffer, two halves of 256 dwords 272 dwords apart
constexpr int kTHalfDw = 272;                     // dword offset of the half that holds columns 4..7 (272 = 16 mod 32: the two halves of a
                                                  // 32-lane write group fall into different banks)
constexpr int kZzWaveBytes = 8 * kZzStrideB;      // 1152 B zig-zag staging per wave; its first 640 B double as the byte-transpose buffer
constexpr int kPxBlkBytes = 80;                   // byte-transpose buffer: 64 pixel bytes per block + 16 (bank spread of the b64 writes)
constexpr int kMaxStripsPerWave = 16;             // a grid is "larger than the chip" when its waves would walk more strips than this
constexpr int kChunkStrips = 8;                   // chunked schedule: strips per wave (a workgroup streams 32 adjacent strips).  With the
                                                  // sc1 nt stores 8 beats round 2's 16: 256 x 1080p 270 against 288 us, 8192^2 37.3 against
                                                  // 38.4, 16384^2 120 against 121 (profiles/r03_ab_chunk.txt)

// Quantiser of the fast path: s = RN(z*mul + magic) is rint(z*mul) of the EXACT product (one rounding, half-even; adding
// kMagic = 1.5 * 2^23 puts the integer's two's complement into the low mantissa bits: no v_rndne / v_cvt), d = RN(z*mul - rint(z*mul))
// is the distance that feeds the guard-band test.  Three instructions per coefficient (v_fmaak, v_sub, v_fmac); the product is
// never rounded on its own, which is what tools/fastpath_error_bound.py's "multiplier" term assumes.
__device__ __forceinline__ void quant_fma(float z, float mul, uint32_t &bits, float &d) {
    const float s = fmaf(z, mul, kMagic);
    bits = __float_as_uint(s);
    const float nr = kMagic - s; // -rint(z*mul), exact
    d = fmaf(z, mul, nr);
}

// Coefficient stores: 16 B per lane, write-through and non-temporal (`sc1 nt`).  Write-through: the launch does not end with a
// write-back of 33 MB of dirty L2 lines (plain stores: +1.5-1.8 us per 4096^2 launch).  Non-temporal on top of it: with plain
// `sc1` the kernel took 13.0 us on data that is not already in the Infinity Cache against 9.7 us (16384^2: 150 against 127 us;
// profiles/r03_ablate_cold.txt, r03_microbench7_cold_floors.txt); replaying one cache-resident frame the two are level.
__device__ __forceinline__ void store16_wt_nt(void *p, const uint4 &v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(d) : "memory");
}
// The same store with a scalar base and a 32-bit lane offset: the strip's address arithmetic stays on the scalar unit (the 64-bit
// vector add per strip and its register pair are gone).
__device__ __forceinline__ void store16_wt_nt(const void *sbase, uint32_t voff, const uint4 &v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" : : "v"(voff), "v"(d), "s"(sbase) : "memory");
}

constexpr int kMaxStripsPerWave2 = 64;                 // one bit per strip of a wave's walk in the exact-redo mask
constexpr int kBatch = 8;                              // entries of the wave's batch
constexpr int kBatchWaveBytes = kBatch * (128 + 64) + 64; // images, pixel rows, ids

// A double through DPP: lane i reads the value of the lane the control word names.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// The four rational coefficients (0,0) (0,4) (4,0) (4,4) of all eight blocks of a strip in the reference's float64 operation order
// (same arithmetic as dct8_exact restricted to outputs 0 and 4 of both passes).  The column pass of the
// reference turns the integer column sums S_c (level-shifted) and alternating sums A_c = (x0+x7+x3+x4) - (x1+x2+x5+x6) into
// y0_c = fl(S_c * SQ2H/2), y4_c = fl(A_c * TW3/2) - one rounding each, everything before the constant is exact.  Those integers are
// pass 1's outputs 0 and 4 of the fast path and sit in the transpose buffer as exact floats: row u = 0 and row u = 4 of every block.
// The row pass adds the eight y in pocketfft's order: A = (y0+y7) + (y3+y4), B = (y1+y2) + (y5+y6), output 0 = fl((A+B) * SQ2H/2),
// output 4 = fl((A-B) * TW3/2) (the power-of-two scalings are exact and commute with the roundings).
// Lane 8*b + i works for block b: w = i >> 2 picks the frequency row u = 4w, k = i & 3 the pair (0,7) (3,4) (1,2) (5,6) whose two
// values the lane converts, scales and adds; one quad_perm exchange gives A (lanes k = 0,1) and B (k = 2,3), a second brings the
// other half over; lane k = 0 finishes v = 0, lane k = 1 finishes v = 4.  Returns the quantised coefficient (true IEEE division,
// half-even) in the lanes i = 0, 1, 4, 5 - (0,0) (0,4) (4,0) (4,4) at scan positions 0, 14, 10, 39.
__device__ __forceinline__ int rational_quad(const uint32_t *ldsT, int b, int i, const double *cst_rat) {
#pragma clang fp contract(off)
    const int w = i >> 2, k = i & 3;
    const uint32_t ca = (0x05010300u >> (8 * k)) & 0xffu, cb = (0x06020407u >> (8 * k)) & 0xffu;
    const uint32_t *row = ldsT + w * 128 + 4 * b; // frequency row u = 4w of block b (rows 0 and 4 are not swizzled)
    const float fa = __uint_as_float(row[(ca >> 2) * kTHalfDw + (ca & 3)]);
    const float fb = __uint_as_float(row[(cb >> 2) * kTHalfDw + (cb & 3)]);
    const double K = w ? (kTW3 * 0.5) : (kSq2h * 0.5);
    const double ya = (double)fa * K, yb = (double)fb * K;
    const double p = ya + yb;                                   // y0+y7 | y3+y4 | y1+y2 | y5+y6
    const double s2 = p + dpp_f64<TIC_DPP_QP_XOR1>(p);          // A (k = 0, 1) | B (k = 2, 3)
    const double o = dpp_f64<TIC_DPP_QP_XOR2>(s2);              // the other half
    const double E = (k & 1) ? s2 - o : s2 + o;                 // k = 0: A + B, k = 1: A - B
    const double X = E * ((k & 1) ? (kTW3 * 0.5) : (kSq2h * 0.5));
    const double div = cst_rat[2 * w + (k & 1)], rdiv = cst_rat[4 + 2 * w + (k & 1)];
    return (int)rint(div_rn(X, div, rdiv)); // np.round(X / div): the correctly rounded quotient (tic_math.h div_rn: five multiply-adds), half-even
}

// Exact float64 sub-path of the four rational coefficients, 8 lanes per block, no LDS: rowLo/rowHi = pixel row i of the lane's
// block.  The row's bytes are first reordered (0,7,3,4,1,2,5,6) so that, after the byte transpose, neighbouring lanes hold the
// columns pocketfft adds first: the 8-point sums of SURVEY Appendix A become three DPP butterflies (xor 1, xor 2, +4).
// Column sums come from v_sad_u8.  Lanes 0..3 of the block return (0,0), (0,4), (4,0), (4,4) - one coefficient, one division at
// most, per lane.  Used by the rows-first instantiation, whose fast path has no column sums.
__device__ __forceinline__ int rational_slim(uint32_t rowLo, uint32_t rowHi, int i, const double *cst_rat) {
#pragma clang fp contract(off)
    uint32_t lo = perm_b32(rowHi, rowLo, 0x04030700u); // x0 x7 x3 x4
    uint32_t hi = perm_b32(rowHi, rowLo, 0x06050201u); // x1 x2 x5 x6
    transpose8x8_bytes(lo, hi, i);                     // lane i: column (0,7,3,4,1,2,5,6)[i], bytes = rows 0..7
    const uint32_t tot = __builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
    const uint32_t sa = __builtin_amdgcn_sad_u8(lo & 0xff0000ffu, 0u, __builtin_amdgcn_sad_u8(hi & 0xff0000ffu, 0u, 0u)); // rows 0,3,4,7
    const int e0 = (int)tot - 1024, e4 = 2 * (int)sa - (int)tot;
    double y0 = (double)e0 * (kSq2h * 0.5);
    double y4 = (double)e4 * (kTW3 * 0.5);
    y0 = y0 + dpp_f64<TIC_DPP_QP_XOR1>(y0); // a0+a7 | a3+a4 | a1+a2 | a5+a6
    y4 = y4 + dpp_f64<TIC_DPP_QP_XOR1>(y4);
    y0 = y0 + dpp_f64<TIC_DPP_QP_XOR2>(y0); // A = p07 + p34 (lanes 0..3) | B = p12 + p56 (lanes 4..7)
    y4 = y4 + dpp_f64<TIC_DPP_QP_XOR2>(y4);
    const double b0 = dpp_f64<TIC_DPP_ROW_SHL4>(y0), b4 = dpp_f64<TIC_DPP_ROW_SHL4>(y4); // lanes 0..3 read B
    const double A = (i & 2) ? y4 : y0, B = (i & 2) ? b4 : b0;       // lanes 0,1: frequency row u = 0; lanes 2,3: u = 4
    const double E = (i & 1) ? A - B : A + B;                         // v = 0 | v = 4
    const double X = E * ((i & 1) ? (kTW3 * 0.5) : (kSq2h * 0.5));
    const double div = cst_rat[i & 3], rdiv = cst_rat[4 + (i & 3)];
    return (int)rint(div_rn(X, div, rdiv)); // (round 5: the reciprocal product, and the compiler's division when a lane sat near a tie - in a tie strip one always does)
}
// A block redone by the whole wave in float64 straight from the definition (lane 8*u + c: t[u][c] = sum_r M[u][r] x[r][c], then
// X[u][v = c] = sum_k M[v][k] t[u][k]; error ~1e-13, the reference's own is ~1e-12).  A rounding is decided when no .5 tie lies
// within 1e-9 of X * (1/div); decided values go to their place in the block's 128-byte zig-zag image.  The four rational
// coefficients are left alone: the image already holds their exact values (fast path outside the guard band, rational_quad inside).
// Returns the lanes that stayed undecided.  ~450 cycles for one block.
__device__ __forceinline__ unsigned long long wave_redo_block(const uint8_t *px /* 64 pixels, LDS */, double *tbuf /* 64 doubles, LDS */,
                                                              const double *cosm, const double *rdiv, const uint16_t *zzofs, int16_t *img16,
                                                              int lane) {
    const int u = lane >> 3, c = lane & 7;
    // (two accumulators per sum and 16-byte table reads: the pass runs on the launch's tail, one wave alone on its SIMD - what counts is
    //  the length of the dependent chain, not the instruction count; the 1e-9 test below does not care about the order of the additions)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *mu = reinterpret_cast<const d2 *>(cosm + u * 8), *mc = reinterpret_cast<const d2 *>(cosm + c * 8);
    const d2 m0 = mu[0], m1 = mu[1], m2 = mu[2], m3 = mu[3];
    double x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = (double)((int)px[r * 8 + c] - 128);
    double ta = m0.x * x[0], tb = m0.y * x[1];
    ta = fma(m1.x, x[2], ta); tb = fma(m1.y, x[3], tb);
    ta = fma(m2.x, x[4], ta); tb = fma(m2.y, x[5], tb);
    ta = fma(m3.x, x[6], ta); tb = fma(m3.y, x[7], tb);
    tbuf[u * 8 + c] = ta + tb;
    const d2 n0 = mc[0], n1 = mc[1], n2 = mc[2], n3 = mc[3]; // (requested in front of the fence: they do not depend on it)
    wave_lds_fence();
    const d2 *tu = reinterpret_cast<const d2 *>(tbuf + u * 8);
    const d2 y0 = tu[0], y1 = tu[1], y2 = tu[2], y3 = tu[3];
    double xa = n0.x * y0.x, xb = n0.y * y0.y;
    xa = fma(n1.x, y1.x, xa); xb = fma(n1.y, y1.y, xb);
    xa = fma(n2.x, y2.x, xa); xb = fma(n2.y, y2.y, xb);
    xa = fma(n3.x, y3.x, xa); xb = fma(n3.y, y3.y, xb);
    const double X = xa + xb;
    const double tq = X * rdiv[lane], rq = rint(tq);
    const bool rational = (lane & 0x1b) == 0; // (u,v) in {0,4} x {0,4}
    const bool decided = fabs(tq - rq) < 0.5 - 1e-9;
    if (decided && !rational) img16[zzofs[lane] >> 1] = (int16_t)(int)rq;
    const unsigned long long und = __ballot(!decided && !rational);
    wave_lds_fence();
    return und;
}

// Second level, 8 lanes per block (up to eight blocks at once): the fast path's butterflies in float64 (error ~1e-13 against the
// reference's ~1e-12).  colLo/colHi: the lane's pixel column.  q[v] = rounding of (u = i, v) where no .5 tie lies within 1e-9 of the
// quotient; the return value has bit v set where it does (undecided: the caller goes to the exact operation order - or, for the four
// rational coefficients, keeps what the image already holds).  ~1,000 cycles, against ~450 per block for wave_redo_block: the batch
// pass takes this one from three entries on, and so do the strips the batch had no room for.
__device__ __forceinline__ uint32_t second_level_8(uint32_t colLo, uint32_t colHi, uint32_t *lds, int b, int i, const double *mul64, int q[8]) {
    double c[8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        c[r] = (double)(int)((colLo >> (8 * r)) & 0xffu);
        c[r + 4] = (double)(int)((colHi >> (8 * r)) & 0xffu);
    }
    dct8_aan(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // down the column
    c[0] -= 1024.0;
    uint32_t w[8], wh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)__double2loint(c[k]);
    transpose8x8_dwords(lds, b, i, w);
#pragma unroll
    for (int k = 0; k < 8; k++) wh[k] = (uint32_t)__double2hiint(c[k]);
    transpose8x8_dwords(lds, b, i, wh);
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = __hiloint2double((int)wh[k], (int)w[k]);
    dct8_aan(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // along frequency row u = i
    uint32_t und = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const double t = c[v] * mul64[i * 8 + v], r = rint(t);
        q[v] = (int)r;
        if (!(fabs(t - r) < 0.5 - 1e-9)) und |= 1u << v;
    }
    return und;
}

// kCols: the pass order of the fast path.  true  - columns first, byte transpose through LDS, ties from the exact column sums (rational_quad):
//                 grids that fit the chip at once, where every wave owns its strips in advance and the launch ends with the unluckiest one;
//        false - rows first, the pixels converted straight out of the landing registers (no byte transpose: 22 LDS cycles and a dependent
//                 LDS round trip less per strip), ties from the pixels (rational_slim, ~75 instructions per tie strip): grids of several
//                 rounds (16384^2, batches), where workgroups are dispatched as others finish and what counts is the average cost per
//                 strip - 16384^2 at q = 10 / 50 / 90: 107 / 119 / 127 us rows first against 123 / 130 / 137 us columns first
//                 (profiles/r05_config5.json, r04_config5.json).  Same results, bit for bit (tests run both orders on every rare-path frame).
template <bool kCols>
__global__ __launch_bounds__(kWavesPerWG * 64, 6) __attribute__((amdgpu_num_vgpr(72))) void dctq_strip_kernel(DctqArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t ldsT_all[kWavesPerWG][kTWaveBytes / 4];
    __shared__ __attribute__((aligned(16))) uint32_t ldsZ_all[kWavesPerWG][kZzWaveBytes / 4];
    __shared__ __attribute__((aligned(16))) unsigned char cst_blk[kStripBlkBytes]; // constants, shared by the workgroup
    __shared__ __attribute__((aligned(16))) uint32_t bat_all[kWavesPerWG][kBatchWaveBytes / 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *ldsT = ldsT_all[wave];
    char *ldsZ = reinterpret_cast<char *>(ldsZ_all[wave]);
    const uint16_t *cst_zz = reinterpret_cast<const uint16_t *>(cst_blk + kBlkZz);  // [64] index u*8+v
    const double *cst_rat = reinterpret_cast<const double *>(cst_blk + kBlkRat);    // div of (0,0) (0,4) (4,0) (4,4)
    uint4 *bat_img = reinterpret_cast<uint4 *>(bat_all[wave]);                      // [kBatch][8] 16-byte pieces: zig-zag images
    uint2 *bat_pix = reinterpret_cast<uint2 *>(bat_all[wave] + kBatch * 32);        // [kBatch][8] pixel rows
    uint32_t *bat_id = bat_all[wave] + kBatch * 48;                                 // [kBatch] block index
    const DctqConsts *__restrict__ C = a.consts;
    a.img += (long)blockIdx.z * a.frame_stride_in; // batch of frames: one grid plane per frame
    a.out = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(a.out) + (long)blockIdx.z * a.frame_stride_out);

    const int lr = lane >> 3, lb = lane & 7; // load phase: pixel row lr of block lb
    const int b = lane >> 3, i = lane & 7;   // compute phase: pixel column c = i of block b (pass 1), frequency row u = i (pass 2)
    unsigned long long mask_exact = 0;       // strips of this wave's walk to redo in the exact order (wave-uniform)
    uint32_t n_second = 0, n_quad = 0;       // blocks recomputed in float64, strips through rational_quad (statistics)
    int nE = 0;                              // entries in the batch (wave-uniform)
    int t_first, n_my;
    const uint32_t st_off = (uint32_t)lane * 16u; // lane offset inside a strip's 1 KiB output
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        f32x4 m0, m1;
        f32x4 thr;
        u32x4 zzv;
        // Constants: the workgroup copies the quality's block (kStripBlkBytes = 2,624 bytes: tic_math.h) into LDS, 41 lanes of every wave
        // one 16-byte piece each (the first version let every lane load its own multipliers, thresholds and offsets - 8, then 12 wave-wide
        // loads per wave in front of the first pixel load: four such loads more cost 0.67 us on a 4096^2 launch).
        // Every VMEM instruction from here to the end of the loop is issued by hand and counted (see TIC_TAKE).
        u32x4 c_fill;
        {
            constexpr int kPpw = kStripBlkPieces / kWavesPerWG; // 41 pieces of 16 bytes per wave
            static_assert(kPpw * kWavesPerWG == kStripBlkPieces && kPpw <= 64, "constant block must split evenly over the waves");
            const uint32_t piece = lane < kPpw ? (uint32_t)(wave * kPpw + lane) : (uint32_t)kStripBlkPieces - 1u;
            const uint32_t fo = piece * 16u;
            asm volatile("global_load_dwordx4 v[76:79], %0, %1" : : "v"(fo), "s"(C->strip_blk) : "memory", "v76", "v77", "v78", "v79"); // (lands in reserved registers: see TIC_LOAD)
        }
        // LDS layouts of the loop (conflict-free: tools/lds_bank_model.py)
        //   byte transpose : block lb's pixel row lr at lb*80 + lr*8; lane (b, c) reads byte r*8 + c of block b, r = 0..7
        //   dword transpose: value (u, c) of block b at dword (c>>2)*272 + u*32 + 4*(b ^ s(u)) + (c&3), s(u) = 4 for u in {2,3,6,7}
        //   zig-zag staging: scan position p of block b at byte (p>>3)*128 + 16*(b ^ f(p>>3)) + 2*(p&7), f(c) = bit 1 of c | 4 * bit 2 of c
        const uint32_t px_wr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(ldsZ + lb * kPxBlkBytes + lr * 8);
        const uint8_t *px_rd = reinterpret_cast<const uint8_t *>(ldsZ + b * kPxBlkBytes + i);
        // (pass 1 runs in the compute layout - lane 8*b + c - columns first, in the load layout - lane 8*r + b - rows first; the half
        //  offset of 272 dwords keeps the compute layout's 32-lane write groups conflict-free, the load layout needs none: 256)
        constexpr int kHalf = kCols ? kTHalfDw : 256;
        const int p1a = kCols ? i : lr, p1b = kCols ? b : lb;
        uint32_t *twA = ldsT + (p1a >> 2) * kHalf + (p1a & 3) + 4 * p1b;       // outputs {0,1,4,5}: + k*32 dwords
        uint32_t *twB = ldsT + (p1a >> 2) * kHalf + (p1a & 3) + 4 * (p1b ^ 4); // outputs {2,3,6,7}
        const uint4 *tr = reinterpret_cast<const uint4 *>(
            __builtin_assume_aligned(ldsT + i * 32 + 4 * (b ^ (4 * ((i >> 1) & 1))), 16)); // first half; the second kHalf dwords on
        const uint32_t ld_off = (uint32_t)(lr * (int)a.stride + lb * 8); // lane offset from the strip's first pixel

        // Strip walk: one scalar cursor (that of the prefetch).  Its start state takes ~25 scalar instructions: no division
        // (magic multipliers from the launcher), no per-workgroup memory.  (Measured: the round-1 prologue's ~250 scalar
        // instructions per wave - four integer divisions - made the last of a CU's five workgroups issue its first load 3,000
        // cycles after the first, the 20 waves of a CU share one scalar unit; a table of per-wave start states read with s_load
        // was as slow: the scalar cache serves misses to distinct lines one by one.)
        uint32_t tf, t_lim;
        if (a.team_count > 0) { // 2-D grid: x = team, y = round
            const uint32_t r = blockIdx.y;
            const uint32_t row0 = (uint32_t)((r < 8u ? a.split_lo >> (8u * r) : a.split_hi) & 0xffull);
            const uint32_t row1 = (uint32_t)((r < 7u ? a.split_lo >> (8u * r + 8u) : a.split_hi) & 0xffull);
            tf = (row0 * (uint32_t)a.team_count + blockIdx.x) * kWavesPerWG + (uint32_t)wave;
            t_lim = tf + (row1 - row0) * (uint32_t)a.tstep;
        } else if (a.round_wgs > 0) {
            const uint32_t rho = a.magic_tstep ? __umulhi(blockIdx.x, a.magic_tstep /* = magic of round_wgs in this schedule */) : blockIdx.x;
            const uint32_t wl = blockIdx.x - rho * (uint32_t)a.round_wgs;
            const uint32_t base = rho * (uint32_t)a.round_wgs * (uint32_t)a.wg_span;
            tf = base + wl * kWavesPerWG + (uint32_t)wave;
            t_lim = base + (uint32_t)a.round_wgs * (uint32_t)a.wg_span;
        } else {
            tf = blockIdx.x * (uint32_t)a.wg_stride + (uint32_t)wave;
            t_lim = blockIdx.x * (uint32_t)a.wg_stride + (uint32_t)a.wg_span;
        }
        tf = __builtin_amdgcn_readfirstlane(tf);
        const uint32_t nfast = (uint32_t)a.fast_ty * (uint32_t)a.fast_tx; // strips handled here: complete, 8-byte aligned, no padding
        const uint32_t t_end = t_lim < nfast ? t_lim : nfast;                 // first strip past this wave's walk
        n_my = 0;
        if (tf < t_end) n_my = (a.round_wgs > 0 || a.magic_tstep == 0u) ? (int)((t_end - tf + (uint32_t)a.tstep - 1u) / (uint32_t)a.tstep)
                                                                        : (int)__umulhi(t_end - tf + (uint32_t)a.tstep - 1u, a.magic_tstep);
        if (n_my == 0) tf = 0; // a wave without strips loads (and discards) strip 0
        t_first = (int)tf;
        const uint32_t ty_first = a.magic_fast_tx ? __umulhi(tf, a.magic_fast_tx) : tf; // (magic 0: one strip per row)
        int txp = (int)(tf - ty_first * (uint32_t)a.fast_tx);
        uint32_t in_off = ty_first * (uint32_t)(8 * a.stride) + (uint32_t)txp * 64u; // frames are < 4 GiB (launcher)
        uint32_t oblk = ty_first * (uint32_t)a.bw + (uint32_t)txp * 8u;
        uint32_t src_off = 0; // a load past the end of the walk re-reads the wave's last strip (strip 0 if it has none)
        int n_issued = 0;
        const uint8_t *img_s = a.img;
        // Pixel loads land in RESERVED registers v72..v79: the kernel is compiled with amdgpu_num_vgpr(72), so the compiler
        // allocates v0..v71 only, and the asm statements below name v72.. explicitly (declared as clobbers, which makes the
        // kernel descriptor cover them: 80 registers, six waves per SIMD).  The compiler never sees a loaded value: it goes from the
        // landing pair to LDS behind the counted wait, inside the same asm statement.  (Rounds 1-2 gave the asm load a "=v" output: the
        // compiler then believes the value exists from that statement on and is free to copy it - a phi move, a coalescing with a
        // register tuple - before it has landed, and to reuse a register that a load in flight will still write.  It happened not
        // to; tools/microbench7.hip faulted exactly that way.  Accumulator registers would do too, but the compiler then splits the
        // 80 registers 40:40 and spills.)  amdgpu_num_vgpr is a budget the allocator aims for, NOT a wall (a rare-path change once
        // made it put float64 division temporaries into v72..v77): the guarantee is csrc/lint_strip_kernel.py (run by the Makefile
        // behind the link and by tests/test_host_cpu.py), which checks in the disassembly of the shipped binary that no instruction
        // outside these statements touches v72..v79.  Keep it green.
#define TIC_RSV_CLOBBER "memory", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define TIC_LOAD(R, OB)                                                                                      \
    do {                                                                                                     \
        src_off = n_issued < n_my ? in_off : src_off;                                                        \
        OB = oblk;                                                                                           \
        asm volatile("global_load_dwordx2 v[" R "], %0, %1" : : "v"(ld_off), "s"(img_s + src_off) : TIC_RSV_CLOBBER); \
        n_issued++; txp += a.step_tx; in_off += a.in_step32; oblk += a.oblk_step;                            \
        if (__builtin_expect(txp >= a.fast_tx, 0)) { txp -= a.fast_tx; in_off += a.in_wrap32; oblk += a.oblk_wrap; } \
    } while (0)
    // waits until all but the N youngest vector-memory operations are done, then sends the strip's pixel rows from the landing pair
    // v[R] (which keeps the strip's bytes until the pair is loaded again, two strips later: the batch fetches the raw words from
    // there, raw_words) to the byte-transpose buffer
#define TIC_TAKE_COLS(R, N) asm volatile("s_waitcnt vmcnt(" #N ")\n\tds_write_b64 %0, v[" R "]" : : "v"(px_wr) : TIC_RSV_CLOBBER)
    // rows first: waits, then converts the strip's eight pixels straight out of the landing registers v[R0], v[R1]
#define TIC_TAKE_ROWS(D, R0, R1, N)                                                                          \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_cvt_f32_ubyte0 %0, v" #R0 "\n\tv_cvt_f32_ubyte1 %1, v" #R0 "\n\tv_cvt_f32_ubyte2 %2, v" #R0 \
                 "\n\tv_cvt_f32_ubyte3 %3, v" #R0 "\n\tv_cvt_f32_ubyte0 %4, v" #R1 "\n\tv_cvt_f32_ubyte1 %5, v" #R1                       \
                 "\n\tv_cvt_f32_ubyte2 %6, v" #R1 "\n\tv_cvt_f32_ubyte3 %7, v" #R1                                                       \
                 : "=v"(D[0]), "=v"(D[1]), "=v"(D[2]), "=v"(D[3]), "=v"(D[4]), "=v"(D[5]), "=v"(D[6]), "=v"(D[7]) : : TIC_RSV_CLOBBER)
        uint32_t ob0, ob1, ob2;
        uint32_t kstrip = 0; // ordinal of the strip in this wave's walk
        TIC_LOAD("72:73", ob0);
        TIC_LOAD("74:75", ob1);
        // the constant piece is older than the pixel loads: it has landed when only those are in flight.  (Issuing the second
        // pixel load behind the barrier instead: +0.13 us, profiles/r02_ab_opt_switches.txt.)
        asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v76\n\tv_mov_b32 %1, v77\n\tv_mov_b32 %2, v78\n\tv_mov_b32 %3, v79"
                     : "=v"(c_fill.x), "=v"(c_fill.y), "=v"(c_fill.z), "=v"(c_fill.w) : : "memory", "v76", "v77", "v78", "v79");
        if (lane < kStripBlkPieces / kWavesPerWG) *reinterpret_cast<u32x4 *>(cst_blk + (wave * (kStripBlkPieces / kWavesPerWG) + lane) * 16) = c_fill;
        // workgroup barrier by hand (the compiler's would also wait for the pixel loads it does not know about)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
        // pass 2's lane holds frequency row u = i (columns first) or frequency column v = i (rows first): its eight multipliers, its
        // three accept thresholds (groups {1,2,3} | {5,6,7} | {0,4} of the other index), its eight offsets in the zig-zag image
        m0 = *reinterpret_cast<const f32x4 *>(cst_blk + (kCols ? kBlkMul : kBlkMulT) + i * 32);
        m1 = *reinterpret_cast<const f32x4 *>(cst_blk + (kCols ? kBlkMul : kBlkMulT) + i * 32 + 16);
        thr = *reinterpret_cast<const f32x4 *>(cst_blk + (kCols ? kBlkThr : kBlkThrT) + i * 16);
        zzv = *reinterpret_cast<const u32x4 *>(cst_blk + (kCols ? kBlkZz : kBlkZzT) + i * 16);
        auto zz_ptr = [&](uint32_t ofs) { // ofs = 2 * scan position of the coefficient
            const uint32_t c = ofs >> 4; // (the chunk's swizzle: found by search for either lane mapping, tools/lds_bank_model.py)
            return reinterpret_cast<int16_t *>(ldsZ + c * 128 + (ofs & 15) + 16 * (b ^ (kCols ? (((c >> 1) & 1) | (c & 4)) : (4 * ((c >> 1) & 1)))));
        };
        int16_t *zp0 = zz_ptr(zzv.x & 0xffff), *zp1 = zz_ptr(zzv.x >> 16), *zp2 = zz_ptr(zzv.y & 0xffff), *zp3 = zz_ptr(zzv.y >> 16);
        int16_t *zp4 = zz_ptr(zzv.z & 0xffff), *zp5 = zz_ptr(zzv.z >> 16), *zp6 = zz_ptr(zzv.w & 0xffff), *zp7 = zz_ptr(zzv.w >> 16);
        const uint4 *zr = reinterpret_cast<const uint4 *>(
            __builtin_assume_aligned(ldsZ + 16 * (i * 8 + (b ^ (kCols ? (((i >> 1) & 1) | (i & 4)) : (4 * ((i >> 1) & 1))))), 16));
        wave_lds_fence();

        int left = n_my;
        // raw pixel words of the strip in work, out of its landing registers (batch entries only)
        auto raw_words = [&](auto tag, uint32_t &lo, uint32_t &hi) {
            constexpr int R = decltype(tag)::value;
            static_assert(R == 72 || R == 74 || R == 76, "landing register pair");
            if constexpr (R == 72) asm volatile("v_mov_b32 %0, v72\n\tv_mov_b32 %1, v73" : "=v"(lo), "=v"(hi) : : "memory");
            else if constexpr (R == 74) asm volatile("v_mov_b32 %0, v74\n\tv_mov_b32 %1, v75" : "=v"(lo), "=v"(hi) : : "memory");
            else asm volatile("v_mov_b32 %0, v76\n\tv_mov_b32 %1, v77" : "=v"(lo), "=v"(hi) : : "memory");
        };
        auto process = [&](auto tag, const float (&pxr)[8], const uint32_t ob) {
            float d0, d1, d2, d3, d4, d5, d6, d7;
            if constexpr (kCols) { // byte transpose: pixel column c = i of block b, rows 0..7 (written by TIC_TAKE_COLS)
                d0 = (float)px_rd[0]; d1 = (float)px_rd[8]; d2 = (float)px_rd[16]; d3 = (float)px_rd[24];
                d4 = (float)px_rd[32]; d5 = (float)px_rd[40]; d6 = (float)px_rd[48]; d7 = (float)px_rd[56];
            } else {               // pixel row lr of block lb, converted by TIC_TAKE_ROWS
                d0 = pxr[0]; d1 = pxr[1]; d2 = pxr[2]; d3 = pxr[3]; d4 = pxr[4]; d5 = pxr[5]; d6 = pxr[6]; d7 = pxr[7];
            }
            // ---- pass 1: down the pixel column | along the pixel row -----------------------------------------------------
            dct8_aan(d0, d1, d2, d3, d4, d5, d6, d7);
            d0 -= 1024.0f;
            twA[0 * 32] = __float_as_uint(d0); twA[1 * 32] = __float_as_uint(d1); twB[2 * 32] = __float_as_uint(d2);
            twB[3 * 32] = __float_as_uint(d3); twA[4 * 32] = __float_as_uint(d4); twA[5 * 32] = __float_as_uint(d5);
            twB[6 * 32] = __float_as_uint(d6); twB[7 * 32] = __float_as_uint(d7);
            wave_lds_fence();
            const uint4 ra = tr[0], rb = tr[kHalf / 4];
            wave_lds_fence();
            float e0 = __uint_as_float(ra.x), e1 = __uint_as_float(ra.y), e2 = __uint_as_float(ra.z), e3 = __uint_as_float(ra.w);
            float e4 = __uint_as_float(rb.x), e5 = __uint_as_float(rb.y), e6 = __uint_as_float(rb.z), e7 = __uint_as_float(rb.w);
            // ---- pass 2: along the frequency row u = i -------------------------------------------------------------------
            dct8_aan(e0, e1, e2, e3, e4, e5, e6, e7);
            uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
            float r0, r1, r2, r3, r4, r5, r6, r7;
            quant_fma(e0, m0.x, q0, r0); quant_fma(e1, m0.y, q1, r1); quant_fma(e2, m0.z, q2, r2); quant_fma(e3, m0.w, q3, r3);
            quant_fma(e4, m1.x, q4, r4); quant_fma(e5, m1.y, q5, r5); quant_fma(e6, m1.z, q6, r6); quant_fma(e7, m1.w, q7, r7);
            // guard test: three groups per row, v in {1,2,3} | {5,6,7} | {0,4} (tic_math.h thrR)
            const float mA1 = fmaxf(fmaxf(fabsf(r1), fabsf(r2)), fabsf(r3)); // v_max3_f32 with |.| modifiers
            const float mA2 = fmaxf(fmaxf(fabsf(r5), fabsf(r6)), fabsf(r7));
            const float mB = fmaxf(fabsf(r0), fabsf(r4));
            const unsigned long long cA = __ballot(mA1 > thr.x) | __ballot(mA2 > thr.y); // lanes whose guard band tripped: v in 1,2,3,5,6,7
            const unsigned long long cB = __ballot(mB > thr.z);                           // ... v in 0,4
            *zp0 = (int16_t)q0; *zp1 = (int16_t)q1; *zp2 = (int16_t)q2; *zp3 = (int16_t)q3;
            *zp4 = (int16_t)q4; *zp5 = (int16_t)q5; *zp6 = (int16_t)q6; *zp7 = (int16_t)q7;
            const unsigned long long kRat = 0x1111111111111111ull; // lanes i in {0,4}: their coefficients 0 and 4 are the rational ones (either order)
            // ---- a rational coefficient sits on a tie somewhere in the strip (one strip in six at q = 50 on noise; every strip of
            //      posterised, two-level or flat content): the exact values of all 32, over the fast path's ------------------------
            if (__builtin_expect((cB & kRat) != 0ull, 0)) {
                n_quad++;
                if constexpr (kCols) {
                    const int rq = rational_quad(ldsT, b, i, cst_rat);
                    // scan positions 0, 14, 10, 39 of lanes i = 0, 1, 4, 5: byte offsets 0, 28, 20, 78 = 28 (i & 1) + 20 (i >> 2) + 30 (i & 1)(i >> 2), by
                    // arithmetic (as nested selects the compiler built three exec-mask regions out of them, inside the tie path)
                    if ((i & 2) == 0) *zz_ptr(28u * (uint32_t)(i & 1) + 20u * (uint32_t)(i >> 2) + 30u * (uint32_t)((i & 1) & (i >> 2))) = (int16_t)rq;
                } else { // rows first: no column sums at hand - from the pixels (load layout -> a row per lane of the block, through LDS)
                    uint32_t lo0, hi0;
                    raw_words(tag, lo0, hi0);
                    uint2 *pb = reinterpret_cast<uint2 *>(ldsT);
                    pb[lb * 8 + lr] = make_uint2(lo0, hi0);
                    wave_lds_fence();
                    const uint2 rowv = pb[lane]; // row i of block b
                    wave_lds_fence();
                    const int rq = rational_slim(rowv.x, rowv.y, i, cst_rat); // lanes 0..3 of a block: (0,0), (0,4), (4,0), (4,4)
                    if (i < 4) *zz_ptr(i == 0 ? 0u : (i == 1 ? 28u : (i == 2 ? 20u : 78u))) = (int16_t)rq;
                }
            }
            wave_lds_fence();
            const uint4 val = *zr;
            wave_lds_fence();
            const char *dst = reinterpret_cast<const char *>(a.out) + ((unsigned long long)ob << 7); // (scalar: the strip's 1 KiB)
            // ---- an irrational coefficient inside its guard band (one strip in seventy at q = 50, one in fourteen at q = 90) ------
            const unsigned long long mG = cA | (cB & ~kRat);
            if (__builtin_expect(mG != 0ull, 0)) {
                // per block: lane l answers for block l & 7 (byte l & 7 of the lane mask); the low byte of the ballot is the 8-bit
                // block mask (folding the bytes on the scalar unit took ~40 dependent scalar instructions per tripped strip)
                const uint32_t gm = (uint32_t)__ballot(((mG >> (8u * (uint32_t)i)) & 0xffull) != 0ull) & 0xffu;
                const int nnew = __builtin_popcount(gm);
                if (nE + nnew <= kBatch && gm != 0xffu) {
                    // the blocks join the batch: id, pixel rows (in the load layout this lane holds row lr of block lb), staged image
                    uint32_t lo0, hi0;
                    raw_words(tag, lo0, hi0);
                    const uint32_t below = (1u << b) - 1u, lbelow = (1u << lb) - 1u;
                    const bool mine = (gm >> b) & 1u;
                    const int e = nE + __builtin_popcount(gm & below);
                    if (mine) bat_img[e * 8 + i] = val;
                    if (mine && i == 0) bat_id[e] = ob + (uint32_t)b;
                    if ((gm >> lb) & 1u) bat_pix[(nE + __builtin_popcount(gm & lbelow)) * 8 + lr] = make_uint2(lo0, hi0);
                    // the tripped blocks leave with the batch pass; the others now (at least one lane stores: the strip's one
                    // vector-memory instruction is issued on every path, which the counted waits rely on)
                    if (!mine) store16_wt_nt(dst, st_off, val);
                    nE += nnew;
                    n_second += (uint32_t)nnew;
                } else {
                    // no room (or all eight blocks tripped): the whole strip is redone in the exact order after the loop
                    mask_exact |= 1ull << kstrip;
                    store16_wt_nt(dst, st_off, val);
                }
            } else
                store16_wt_nt(dst, st_off, val); // 16 B per lane, 1 KiB contiguous per wave
            left--;
            kstrip++;
        };
        // Two strips ahead: strip j is consumed after L(j+2) is issued; in steady state the instructions younger than
        // L(j) are S(j-2) L(j+1) S(j-1) L(j+2) -> vmcnt(4); the first two strips see 2 and 3.  (A rare branch issues at most
        // the same single store per strip.  Three strips ahead: no faster, profiles/r02_ab_prefetch_depth.txt, r03_ablate_cold.txt.)
        float pxf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define TIC_STEP(RLOAD, OBL, RTAKE, R0, R1, OBP, N)                                                          \
        TIC_LOAD(RLOAD, OBL);                                                                                \
        if constexpr (kCols) { TIC_TAKE_COLS(RTAKE, N); } else { TIC_TAKE_ROWS(pxf, R0, R1, N); }            \
        process(std::integral_constant<int, R0>(), pxf, OBP)
        do {
            if (left == 0) break;
            TIC_STEP("76:77", ob2, "72:73", 72, 73, ob0, 2);
            if (left == 0) break;
            TIC_STEP("72:73", ob0, "74:75", 74, 75, ob1, 3);
            while (left != 0) {
                TIC_STEP("74:75", ob1, "76:77", 76, 77, ob2, 4);
                if (left == 0) break;
                TIC_STEP("76:77", ob2, "72:73", 72, 73, ob0, 4);
                if (left == 0) break;
                TIC_STEP("72:73", ob0, "74:75", 74, 75, ob1, 4);
            }
        } while (0);
#undef TIC_STEP
        // Loads past the end of the walk (clamped addresses) may still be in flight: they land in v72..v79, which nothing behind
        // the loop names (the compiler never allocates them), so nothing waits for them - nor for the acknowledgement of the
        // strip stores, which is what rounds 1-3's `s_waitcnt vmcnt(1)` here really waited for (~580 cycles at the head of every
        // batch pass; profiles/r04_tail_experiments.txt).  The statement stays as a scheduling fence with the same clobbers.
        asm volatile("; end of the strip walk" : : : TIC_RSV_CLOBBER);
#undef TIC_LOAD
#undef TIC_TAKE_COLS
#undef TIC_TAKE_ROWS
#undef TIC_RSV_CLOBBER
    }
    // ---- the batch pass: blocks with an irrational coefficient inside its guard band ----------------------------------------
    if (nE != 0) {
        const double *cst_cos = reinterpret_cast<const double *>(cst_blk + kBlkCos);   // orthonormal DCT-II matrix, index k*8+n
        const double *cst_rdiv = reinterpret_cast<const double *>(cst_blk + kBlkRdiv); // 1/div, index u*8+v
        // (1) float64 recompute: whole wave, one block at a time - or, when the wave met a cluster of trips (three entries and more),
        // all entries at once, 8 lanes each
        uint32_t m_exact = 0; // entries that need the exact operation order
        if (nE < 3) {
            for (int ee = 0; ee < nE; ee++) {
                const unsigned long long und = wave_redo_block(reinterpret_cast<const uint8_t *>(bat_pix + ee * 8), reinterpret_cast<double *>(ldsT), cst_cos,
                                                               cst_rdiv, cst_zz, reinterpret_cast<int16_t *>(bat_img + ee * 8), lane);
                if (und != 0ull) m_exact |= 1u << ee;
            }
        } else {
            const bool have2 = b < nE;
            const int e2 = have2 ? b : 0;
            const uint2 rowv = bat_pix[e2 * 8 + i];
            uint32_t lo = rowv.x, hi = rowv.y;
            transpose8x8_bytes(lo, hi, i);
            int qs[8];
            uint32_t und = second_level_8(lo, hi, ldsT, b, i, reinterpret_cast<const double *>(cst_blk + kBlkMul64), qs);
            if ((i & 3) == 0) und &= ~0x11u; // the rational coefficients (u, v in {0,4}): the image holds their exact values
            const uint4 zo = *reinterpret_cast<const uint4 *>(cst_zz + i * 8);
            const uint32_t zw[4] = {zo.x, zo.y, zo.z, zo.w};
            int16_t *img2 = reinterpret_cast<int16_t *>(bat_img + e2 * 8);
            if (have2) {
#pragma unroll
                for (int v = 0; v < 8; v++)
                    if (!((und >> v) & 1u) && !((i & 3) == 0 && (v & 3) == 0)) img2[((zw[v >> 1] >> (16 * (v & 1))) & 0xffffu) >> 1] = (int16_t)qs[v];
            }
            const unsigned long long um = __ballot(have2 && und != 0u);
            for (int ee = 0; ee < nE; ee++)
                if ((um >> (8 * ee)) & 0xffull) m_exact |= 1u << ee;
        }
        // (2) 8 lanes per entry: the exact order where even float64 from the definition could not decide (a true tie of an
        // irrational coefficient - practically never)
        const bool have = b < nE;
        const int e = have ? b : 0;
        const uint32_t blk = bat_id[e];
        if (m_exact != 0u) {
            const uint2 rowv = bat_pix[e * 8 + i]; // pixel row i of the block
            int16_t *img16 = reinterpret_cast<int16_t *>(bat_img + e * 8);
            uint32_t lo = rowv.x, hi = rowv.y;
            transpose8x8_bytes(lo, hi, i); // -> pixel column i
            const uint4 zo = *reinterpret_cast<const uint4 *>(cst_zz + i * 8); // byte offsets in the image of (u = i, v = 0..7)
            const uint32_t zw[4] = {zo.x, zo.y, zo.z, zo.w};
            int qx[8];
            exact_block(lo, hi, ldsT, b, i, C, qx);
            if (have && ((m_exact >> e) & 1u)) {
#pragma unroll
                for (int v = 0; v < 8; v++) img16[((zw[v >> 1] >> (16 * (v & 1))) & 0xffffu) >> 1] = (int16_t)qx[v];
            }
        }
        wave_lds_fence();
        const uint4 val = bat_img[e * 8 + i];
        if (have) store16_wt_nt(reinterpret_cast<char *>(a.out) + ((unsigned long long)blk << 7) + (uint32_t)i * 16u, val);
    }
    if (a.fallback_count != nullptr && lane == 0) { // diagnostics (tic_set_stats): never in a timed launch
        if (n_second != 0) atomicAdd(a.fallback_count, (unsigned long long)n_second);
        if (n_quad != 0) atomicAdd(a.fallback_count + 1, (unsigned long long)n_quad);
        if (mask_exact != 0ull) atomicAdd(a.fallback_count + 2, (unsigned long long)__builtin_popcountll(mask_exact));
        if (nE != 0) atomicMax(a.fallback_count + 3, (unsigned long long)nE);
    }
    // ---- strips the batch had no room for: the exact operation order, whole strip ---------------------------------------
    if (mask_exact == 0ull) return;
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); // the wave's fast-path stores to these strips must have landed
    const uint4 zzn = *reinterpret_cast<const uint4 *>(cst_zz + i * 8);
    const uint16_t zz[8] = {(uint16_t)zzn.x, (uint16_t)(zzn.x >> 16), (uint16_t)zzn.y, (uint16_t)(zzn.y >> 16),
                            (uint16_t)zzn.z, (uint16_t)(zzn.z >> 16), (uint16_t)zzn.w, (uint16_t)(zzn.w >> 16)};
    for (unsigned long long todo = mask_exact; todo != 0ull; todo &= todo - 1ull) {
        const int k = __builtin_ctzll(todo);
        const long t = (long)t_first + (long)k * a.tstep; // strip index inside the fast rectangle
        const int ty = (int)(t / a.fast_tx), tx = (int)(t - (long)ty * a.fast_tx);
        Strip s;
        s.by = ty;
        s.bx = tx * 8 + b;
        s.valid = true;
        s.oblk = (size_t)ty * a.bw + s.bx;
        uint32_t lo, hi;
        load_block_row(a.img, a.h, a.w, a.stride, true, s, i, lo, hi);
        transpose8x8_bytes(lo, hi, i);
        int q[8];
        // the second level for the whole strip; the exact order only if it leaves something undecided (a tie of a rational
        // coefficient included: this path has no column sums at hand)
        const uint32_t und = second_level_8(lo, hi, ldsT, b, i, reinterpret_cast<const double *>(cst_blk + kBlkMul64), q);
        if (__ballot(und != 0u) != 0ull) exact_block(lo, hi, ldsT, b, i, C, q);
        store_zigzag(reinterpret_cast<uint32_t *>(ldsZ), b, i, zz, q, a.out, s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 1b: exact path for integer images outside 0..255 (the reference transforms any integers: codec.py:29 is
// `astype(int32) - 128`).  int32 pixels in, int32 coefficients out (zig-zag order), float64 in pocketfft's order throughout.
// A drop-in edge, not a hot path: lane 8*b + i gathers pixel column i of block b itself (reflect padding as pad_image).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64) void dctq_exact_wide_kernel(WideArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerWG][kLdsWaveBytes / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    const int tile = blockIdx.x * kWavesPerWG + wave;
    Strip s = make_strip(tile, a.ntiles, a.tiles_x, a.bw, b);
    double c[8];
#pragma unroll
    for (int r = 0; r < 8; r++) c[r] = 0.0;
    if (s.valid) {
        const int x = reflect_index(s.bx * 8 + i, a.w);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int y = reflect_index(s.by * 8 + r, a.h);
            c[r] = (double)a.img[(long)y * a.stride + x] - 128.0; // (int32 - 128 is exact in float64)
        }
    }
    dct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -2: down the column
    uint32_t w[8], wh[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)__double2loint(c[k]);
    transpose8x8_dwords(lds, b, i, w);
#pragma unroll
    for (int k = 0; k < 8; k++) wh[k] = (uint32_t)__double2hiint(c[k]);
    transpose8x8_dwords(lds, b, i, wh);
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = __hiloint2double((int)wh[k], (int)w[k]);
    dct8_exact(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); // axis -1: along frequency row u = i
    if (!s.valid) return;
    const double *div = a.consts->div + i * 8;
    const uint16_t *zz = a.consts->zzofs + i * 8; // byte offset of (u = i, v) in an int16 zig-zag block = 2 * scan position
    int32_t *ob = a.out + s.oblk * 64;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        ob[zz[v] >> 1] = (int32_t)rint(c[v] / div[v]); // np.round(X / div).astype(int32)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 3: decode side - dequantise (utils.py:52), inverse DCT (utils.py:40-45, exact order), +128, clip,
// truncating cast (codec.py:68-70), crop.  Input: int16 [N][64] zig-zag, DC already integrated (np.cumsum).
// ---------------------------------------------------------------------------------------------------------
// constants.py:37-51: ANNSCALES = this integer table / 2048 - the scale the reference's C encoder leaves in its coefficients
// (8 * a_u * a_v of the AAN factorisation, 14-bit fixed point).  decode()'s scaled_dct branch divides by it.
__constant__ int kAnnScalesInt[64] = {
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
    21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,
    8867,  12299, 11585, 10426, 8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

__global__ __launch_bounds__(kWavesPerWG * 64) void idct_kernel(IdctArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[kWavesPerWG][kLdsWaveBytes / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = lane >> 3, i = lane & 7;
    uint32_t *lds = lds_all[wave];
    const int tile = blockIdx.x * kWavesPerWG + wave;
    const DctqConsts *__restrict__ C = a.consts;
    Strip s = make_strip(tile, a.ntiles, a.tiles_x, a.bw, b);
    if (a.first_block >= 0) { // block-range form: 8 consecutive blocks of the raster order per wave
        const long blk = a.first_block + (long)tile * 8 + b;
        s.valid = blk < a.first_block + a.nblocks_sel;
        s.by = (int)(blk / a.bw);
        s.bx = (int)(blk - (long)s.by * a.bw);
        s.oblk = (size_t)blk;
    }

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
        if (a.scaled) // codec.py:60-62: (coeffs / ANNSCALES) * 2**quality, then the inverse quantiser of quality 50: three roundings
            c[u] = (((double)cv / ((double)kAnnScalesInt[u * 8 + i] / 2048.0)) * a.pow2) * C->div[u * 8 + i];
        else
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

// Schedule knobs of the strip kernel's launcher.  The product uses the defaults below; the environment is consulted only behind
// the test-hook gate (tic_hooks.h, TIC_TEST_HOOKS=1: tests/test_gpu_parity.py::test_strip_schedules_are_equivalent drives every
// schedule against the exact kernel) - read once, or at every launch when TIC_TUNE is set too.
struct Tunables {
    int max_wgs, sched, chunk, order;
    int split[8];
};
static Tunables read_tunables() {
    auto geti = [](const char *k, int d) { const char *v = test_hook(k); return v ? atoi(v) : d; };
    Tunables t;
    t.max_wgs = geti("TIC_MAX_WGS", 0);             // persistent grid size (0: resident workgroups of the chip)
    t.sched = geti("TIC_SCHED", 1);                 // grids larger than the chip: 0 strided, 1 chunked (default), 2 round-interleaved
    t.chunk = geti("TIC_CHUNK", kChunkStrips);      // strips per wave of schedules 1 and 2
    t.order = geti("TIC_ORDER", -1);                // pass order of the strip kernel: -1 by grid (default), 0 rows first, 1 columns first
    // per-round row weights of the team schedule ("0" disables it): the six workgroups of a CU reach their first pixel
    // 1,400 ... 6,200 cycles after their own entry (profiles/r03_stamps_tail.txt), later rounds get fewer strip rows.  Round 5: the
    // first round gives a row to the last (4096^2: 9,8,6,4,3,2 rows instead of round 3's 10,8,6,4,3,1 - a wave's rare work grows with
    // its strips, and the launch ends with the unluckiest wave: 9.9-10.0 against 10.2-10.3 us, profiles/r05_split_weights.txt)
    const char *sp = test_hook("TIC_SPLIT") ? test_hook("TIC_SPLIT") : "15,13,10,7,5,3";
    for (int k = 0; k < 8; k++) t.split[k] = 0;
    for (int k = 0; k < 8 && sp && *sp; k++) {
        t.split[k] = atoi(sp);
        sp = strchr(sp, ',');
        if (sp) sp++;
    }
    return t;
}
static Tunables tunables() {
    static const bool live = test_hook("TIC_TUNE") != nullptr;
    static const Tunables once = read_tunables();
    return live ? read_tunables() : once;
}

int dctq_kernel_id(int abi_variant) { // (TIC_KERNEL_AUTO 0, TIC_KERNEL_EXACT 1, TIC_KERNEL_HYBRID 2: include/tinyimgcodec_hip.h)
    return abi_variant == 1 ? 1 : ((abi_variant == 0 || abi_variant == 2) ? 2 : -1);
}

// variant 1: the exact kernel for every block; variant 2: the strip kernel on the rectangle of complete 64x8 strips with 8-byte
// aligned rows, the exact kernel on what is left (right / bottom padding).
hipError_t launch_dctq(DctqArgs a, int variant, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.ntiles <= 0) return hipSuccess;
    if (variant != 1 && variant != 2) return hipErrorInvalidValue;
    dim3 block(kWavesPerWG * 64);
    a.nwaves = a.step_ty = a.step_tx = 0;
    a.fast_ty = a.fast_tx = 0;
    a.rem_mode = 0;
    a.dbg = nullptr;
    const int nf = a.nframes > 0 ? a.nframes : 1;
    if (variant == 1) {
        hipLaunchKernelGGL(dctq_exact_kernel, dim3(grid_for(a.ntiles), nf), block, 0, stream, a);
        return hipGetLastError();
    }
    // The strip walk keeps 32-bit pixel offsets: a frame of 4 GiB or more is transformed in bands of whole block rows, one launch
    // per band (the stage has no dependency between block rows: the DC is not differenced here).  Round 2 sent such frames to
    // the exact kernel as a whole.  (TIC_BAND_BYTES, a test hook, lowers the limit so that the banding can be tested on small frames.)
    {
        long limit = 1L << 32;
        if (const char *e = test_hook("TIC_BAND_BYTES")) limit = atol(e) > 0 ? atol(e) : limit;
        if (a.aligned8 && nf == 1 && a.w >= 64 && (long)a.h * a.stride >= limit) {
            const long band = (limit - 1) / a.stride / 8 * 8; // pixel rows per band
            if (band >= 8) {
                for (long y0 = 0; y0 < a.h; y0 += band) {
                    DctqArgs b = a;
                    b.h = (int)((long)a.h - y0 < band ? (long)a.h - y0 : band);
                    b.img = a.img + y0 * a.stride;
                    b.out = a.out + (y0 / 8) * (long)a.bw * 64;
                    b.ntiles = ((b.h + 7) / 8) * b.tiles_x;
                    const hipError_t e = launch_dctq(b, variant, stream);
                    if (e != hipSuccess) return e;
                }
                return hipSuccess;
            }
        }
    }
    const int bh = a.ntiles / a.tiles_x;
    a.fast_tx = (a.aligned8 && (long)a.h * a.stride < (1L << 32)) ? a.w / 64 : 0; // 32-bit pixel offsets in the walk
    a.fast_ty = a.h / 8;
    const int nfast = a.fast_tx * a.fast_ty;
    if (nfast > 0) {
        // persistent waves, each looping over its strips; one mask bit per strip of a wave's walk: at most 64 strips per wave
        const int max_strips = kMaxStripsPerWave2;
        int wgs = grid_for(nfast);
        // persistent grid = exactly the workgroups the chip holds at once (CUs x resident workgroups per CU): a larger
        // grid runs in two uneven rounds, a smaller one leaves wave slots empty (measured: 15.0 us at 1280 workgroups
        // vs 16.3 us at 2048 on a 4096^2 frame)
        static int cus = 256;
        static const int resident = [] {
            int dev = 0, per_cu = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dctq_strip_kernel<true>, kWavesPerWG * 64, 0) != hipSuccess || per_cu < 1)
                per_cu = 4;
            // 80 VGPRs and 22 KiB of LDS allow 6 workgroups per CU (7 measured no better when the kernel still fitted them:
            // a seventh lengthens the start ramp by as much as it hides, profiles/r02_ab_occupancy.txt)
            if (per_cu > 6) per_cu = 6;
            return cus * per_cu;
        }();
        const Tunables tune = tunables();
        const int cap_env = tune.max_wgs > 0 ? tune.max_wgs : resident;
        int cap = cap_env / nf; // a batch shares the chip's wave slots between its frames
        if (cap < 64) cap = 64;
        if (wgs > cap) wgs = cap;
        const int min_wgs = (nfast + kWavesPerWG * max_strips - 1) / (kWavesPerWG * max_strips);
        if (wgs < min_wgs) wgs = min_wgs; // strips per wave are bounded (exact-redo mask)
        a.nwaves = wgs * kWavesPerWG;
        a.wg_stride = kWavesPerWG;
        a.tstep = a.nwaves;
        a.wg_span = nfast;
        a.round_wgs = 0;
        a.team_count = 0;
        const int S = tune.chunk < 1 ? 1 : (tune.chunk > max_strips ? max_strips : tune.chunk);
        const int min_wgs16 = (nfast + kWavesPerWG * kMaxStripsPerWave - 1) / (kWavesPerWG * kMaxStripsPerWave);
        const bool multi_round = (long)min_wgs16 * nf > (long)cap_env; // more workgroups than the chip holds at once
        if (tune.sched == 1 && multi_round) { // each workgroup streams a contiguous chunk of 4*S strips
            a.wg_stride = a.wg_span = kWavesPerWG * S;
            a.tstep = kWavesPerWG;
            wgs = (nfast + a.wg_stride - 1) / a.wg_stride;
            a.nwaves = wgs * kWavesPerWG;
        } else if (tune.sched == 2 && multi_round && nf == 1) { // rounds of `resident` workgroups walking a dense range together
            a.round_wgs = cap_env;
            a.wg_span = kWavesPerWG * S;
            a.tstep = a.round_wgs * kWavesPerWG;
            wgs = (nfast + a.wg_span - 1) / a.wg_span; // the last round may hold workgroups with nothing to do
            wgs = (wgs + a.round_wgs - 1) / a.round_wgs * a.round_wgs;
            a.nwaves = wgs * kWavesPerWG;
        }
        if (!multi_round && nf == 1 && tune.split[0] > 0 && wgs == cap_env && wgs % cus == 0 && wgs / cus <= 8) {
            // the whole grid is resident: teams of wgs/cus workgroups, rows split by the per-round weights
            const int R = wgs / cus;
            a.team_count = cus;
            a.tstep = cus * kWavesPerWG;
            const int rows_total = (nfast + a.tstep - 1) / a.tstep;
            double wsum = 0;
            for (int r = 0; r < R; r++) wsum += tune.split[r] > 0 ? tune.split[r] : 1;
            double acc = 0;
            a.split[0] = 0;
            for (int r = 0; r < R; r++) {
                acc += tune.split[r] > 0 ? tune.split[r] : 1;
                a.split[r + 1] = (int)(rows_total * acc / wsum + 0.5);
                if (a.split[r + 1] - a.split[r] > max_strips) a.team_count = 0; // strips per wave are bounded: fall back
            }
            a.split[R] = rows_total;
            if (rows_total > 255) a.team_count = 0; // the kernel takes the row boundaries as bytes
            if (a.team_count == 0) a.tstep = a.nwaves;
        }
        a.step_ty = a.tstep / a.fast_tx;
        a.step_tx = a.tstep % a.fast_tx;
        a.in_step32 = (uint32_t)((long)a.step_ty * 8 * a.stride + (long)a.step_tx * 64);
        a.in_wrap32 = (uint32_t)(8 * a.stride - (long)a.fast_tx * 64);
        a.oblk_step = (uint32_t)((long)a.step_ty * a.bw + (long)a.step_tx * 8);
        a.oblk_wrap = (uint32_t)((long)a.bw - (long)a.fast_tx * 8);
        // division-free prologue (magic multipliers), team schedule on a 2-D grid.  Exactness of the magic divisions:
        // n * d < 2^32 for every dividend n the prologue can form; frames beyond that (more than ~10^5 pixels wide and
        // millions of strips) go through the exact kernel as a whole
        dim3 grid(wgs, 1, nf);
        const unsigned long long dmax = (unsigned long long)(a.round_wgs > 0 ? a.round_wgs : a.tstep);
        if ((unsigned long long)nfast * (unsigned long long)a.fast_tx >= (1ull << 32) ||
            ((unsigned long long)nfast + dmax) * dmax >= (1ull << 32)) {
            a.fast_tx = a.fast_ty = 0;
        } else {
            auto magic = [](long d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (unsigned long long)d + 1ull); }; // 0: divisor 1
            a.magic_fast_tx = magic(a.fast_tx);
            a.magic_tstep = magic(a.round_wgs > 0 ? a.round_wgs : a.tstep);
            a.split_lo = a.split_hi = 0;
            if (a.team_count > 0) {
                const int R = wgs / a.team_count;
                for (int k = 0; k <= R; k++) {
                    if (k < 8) a.split_lo |= (unsigned long long)(a.split[k] & 0xff) << (8 * k);
                    else a.split_hi = (unsigned long long)(a.split[k] & 0xff);
                }
                grid = dim3(a.team_count, R, nf);
            }
            // pass order: columns first where the whole grid is resident at once (static shares: the tail decides), rows first for grids
            // of several rounds (dynamic dispatch: the average cost per strip decides); TIC_ORDER (test hook) forces one
            const bool cols = tune.order < 0 ? !multi_round : tune.order == 1;
            if (ev_start && ev_stop) { // the dispatch packet's own time stamps (tic_dctq_dev_timed_warm's per-launch times)
                if (cols)
                    hipExtLaunchKernelGGL(dctq_strip_kernel<true>, grid, block, 0, stream, ev_start, ev_stop, 0, a);
                else
                    hipExtLaunchKernelGGL(dctq_strip_kernel<false>, grid, block, 0, stream, ev_start, ev_stop, 0, a);
            } else if (cols)
                hipLaunchKernelGGL(dctq_strip_kernel<true>, grid, block, 0, stream, a);
            else
                hipLaunchKernelGGL(dctq_strip_kernel<false>, grid, block, 0, stream, a);
        }
    } else {
        a.fast_tx = a.fast_ty = 0;
    }
    const int nrem = bh * (a.tiles_x - a.fast_tx) + (bh - a.fast_ty) * a.fast_tx;
    if (nrem > 0) {
        a.rem_mode = 1;
        hipLaunchKernelGGL(dctq_exact_kernel, dim3(grid_for(nrem), nf), block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_dctq_wide(const WideArgs &a, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(dctq_exact_wide_kernel, dim3(grid_for(a.ntiles)), dim3(kWavesPerWG * 64), 0, stream, a);
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
