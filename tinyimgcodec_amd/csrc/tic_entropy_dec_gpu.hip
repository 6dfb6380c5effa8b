// tic_entropy_dec_gpu.hip - Huffman + run-length decode of a long stream on the GPU (decode side of SURVEY.md section 8f:
// decode_huffman huffman.py:77-98, decode_run_length huffman.py:36-38, the block loop of decompress() codec.py:178-186, np.cumsum of
// the DC differences codec.py:53).
//
// The format has no restart markers: a decoder is one dependent chain from bit 128 to the end.  But Huffman streams
// re-synchronise: a decoder started at an arbitrary bit as if a block began there lands on true block starts within a block or two.
// The host decoder uses that on 16 threads (tic_entropy.cpp decode_parallel: 7.7 ms for a 4096^2 stream); this is the same idea on
// tens of thousands of lanes, and the coefficients never leave the device:
//   measure   a lane per RANGE of kRange stream bits: measures blocks from the range's first bit as if a block started there (a
//             guess for every range but the first), recording each block's first bit; an invalid prefix or a block of more than 63
//             coefficients moves the guess on by one bit.
//   stitch    a lane per range, all in parallel: HYPOTHESIS: the true chain enters range t where range t-1's trace ended.  From there
//             the lane measures blocks "by hand" until it lands on a block start the range's own trace recorded; from that entry on
//             the trace IS the true chain (a block start carries no state), so the trace's end is where the true chain enters range
//             t+1 - which is the hypothesis for t+1.  Range 0 starts on a true block start, so if EVERY lane finds its
//             synchronisation point inside its range, induction makes every hypothesis true.  Otherwise: give up.
//   scan      exclusive prefix sum of the ranges' true block counts -> index of each range's first block.
//   decode    a lane per range decodes its true blocks (by-hand ones and trace ones are consecutive in the stream) through an LDS
//             image of the block into the int16 [N][64] coefficient array, DC differences aside.
//   scan      inclusive prefix sum of the DC differences (np.cumsum), saturated into entry 0 of every block.
// Anything unusual ON THE TRUE CHAIN - an invalid prefix, more than 63 coefficients in a block, a range without a synchronisation
// point, a measurement that failed behind the synchronisation point - raises a flag and the caller decodes the whole stream on the
// host, whose bit-serial path reproduces the reference's behaviour on malformed streams (exactly the host parallel decoder's rule).
// Blocks that start in the last 2048 bits of the stream are left to the host as well (`m` blocks are produced here, with the read
// position and the running DC behind them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "tic_entropy_dec_gpu.h"

namespace tic {
namespace {

// Stream bits per range: 512, 1024 or 2048 (a block is at most 64 x 27 = 1,728 bits long), chosen by the caller from the stream's
// average block length: the kernels are latency-bound per lane (one dependent chain of table look-ups), so their time goes with the
// bits a lane walks, and a 7 MB stream cut into 2048-bit ranges is only 440 waves for 1,024 SIMDs.  The stitch needs every range
// to hold a synchronisation point: a range shorter than the stream's long blocks makes it give up (the caller then tries 2048).
constexpr int kRangeMax = 2048;
__host__ __device__ constexpr uint32_t cap_of(uint32_t range) { return range / 6u + 2u; } // block starts a range can hold (a block has at least 6 bits: 2-bit DC code + EOB)

// 32 stream bits (MSB first) from bit `pos`; the two big-endian words around it are cached in registers and reloaded when the
// position leaves them (a symbol is 5-8 bits on average: one reload per ~5 symbols).
struct BitWin {
    uint32_t idx, a, b;
};
__device__ __forceinline__ uint32_t peek32(const uint32_t *__restrict__ words, uint32_t pos, BitWin &c) {
    const uint32_t wi = pos >> 5;
    if (wi != c.idx) {
        c.idx = wi;
        c.a = __builtin_bswap32(words[wi]);
        c.b = __builtin_bswap32(words[wi + 1]);
    }
    const uint32_t sh = pos & 31u;
    return sh ? __builtin_amdgcn_alignbit(c.a, c.b, 32u - sh) : c.a; // ({a,b} >> (32 - sh)) low word = (a << sh) | (b >> (32 - sh))
}
// value bits behind a codeword of `len` bits (bitbuffer.py:55-65): x with its top bit clear stands for x - (2^size - 1)
__device__ __forceinline__ int value_of(uint32_t pk, int len, int size) {
    if (size == 0) return 0;
    const uint32_t x = (pk << len) >> (32 - size);
    return (x >> (size - 1)) ? (int)x : (int)x - ((1 << size) - 1);
}

// One block on the table-driven fast path: the device form of block_fast() in tic_entropy.cpp (same tables, same rules).  STORE: the
// coefficients 1..63 go to c (zeroed by the caller).  Returns false on anything unusual with nothing consumed.
template <bool STORE>
__device__ __forceinline__ bool block_dev(const uint32_t *__restrict__ words, const DecLutsDev *__restrict__ L, const uint16_t *ac11 /* LDS */,
                                          uint32_t pos0, BitWin &win, int16_t *c, int &dc_diff, uint32_t &used) {
    uint32_t pos = pos0;
    uint32_t pk = peek32(words, pos, win);
    uint32_t e = L->dc11[pk >> 21];
    if (!e) return false; // DC categories are at most 9 bits long
    int len = (int)(e >> 8), size = (int)(e & 15u);
    dc_diff = value_of(pk, len, size);
    pos += (uint32_t)(len + size);
    int k = 1;
    for (;;) {
        pk = peek32(words, pos, win);
        e = ac11[pk >> 21];
        if (!e) e = L->ac16[pk >> 16];
        if (!e) return false;
        len = (int)(e >> 8);
        size = (int)(e & 15u);
        pos += (uint32_t)(len + size);
        if ((e & 0xffu) == 0u) break; // EOB
        k += (int)((e >> 4) & 15u);
        if (k > 63) return false;
        if (STORE) c[k] = (int16_t)value_of(pk, len, size);
        k++;
    }
    used = pos - pos0;
    return true;
}

__device__ __forceinline__ void load_ac11(uint16_t *lds, const DecLutsDev *__restrict__ L) {
    for (int i = threadIdx.x; i < 2048 / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(L->ac11)[i];
    __syncthreads();
}

__global__ __launch_bounds__(64) void dec_measure_kernel(const uint32_t *__restrict__ words, const DecLutsDev *__restrict__ L, uint32_t fast_end,
                                                         uint32_t range, uint32_t nranges, uint16_t *__restrict__ starts, uint32_t *__restrict__ nrec,
                                                         uint32_t *__restrict__ endpos, int *__restrict__ lastbrk, DecStatus *__restrict__ st) {
    __shared__ uint16_t ac11[2048];
    load_ac11(ac11, L);
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    const uint32_t lo = 128u + t * range;
    const uint32_t hi = lo + range < fast_end ? lo + range : fast_end;
    BitWin win = {0xffffffffu, 0u, 0u};
    uint32_t pos = lo, cnt = 0;
    int brk = -1;
    while (pos < hi) {
        int d;
        uint32_t used;
        if (block_dev<false>(words, L, ac11, pos, win, nullptr, d, used)) {
            if (cnt < cap_of(range)) starts[(size_t)t * cap_of(range) + cnt] = (uint16_t)(pos - lo);
            cnt++;
            pos += used;
        } else {
            if (t == 0u) { // the true chain itself: unusual
                atomicOr(&st->giveup, 1);
                break;
            }
            brk = (int)cnt; // a guess that led nowhere (or, behind the point of synchronisation, a malformed stream): next bit
            pos++;
        }
    }
    if (cnt > cap_of(range)) atomicOr(&st->giveup, 2); // (cannot happen: a block has at least 6 bits)
    nrec[t] = cnt;
    endpos[t] = pos;
    lastbrk[t] = brk;
}

__global__ __launch_bounds__(64) void dec_stitch_kernel(const uint32_t *__restrict__ words, const DecLutsDev *__restrict__ L, uint32_t fast_end,
                                                        uint32_t range, uint32_t nranges, const uint16_t *__restrict__ starts, const uint32_t *__restrict__ nrec,
                                                        const uint32_t *__restrict__ endpos, const int *__restrict__ lastbrk,
                                                        uint32_t *__restrict__ nblk, uint32_t *__restrict__ pstart, DecStatus *__restrict__ st) {
    __shared__ uint16_t ac11[2048];
    load_ac11(ac11, L);
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    if (t == 0u) {
        nblk[0] = nrec[0];
        pstart[0] = 128u;
        return;
    }
    const uint32_t lo = 128u + t * range;
    const uint32_t hi = lo + range < fast_end ? lo + range : fast_end;
    uint32_t pos = endpos[t - 1]; // hypothesis: where the true chain enters this range
    pstart[t] = pos;
    if (pos >= fast_end) { // the chain left the fast part of the stream in front of this range
        nblk[t] = 0u;
        return;
    }
    const uint16_t *tr = starts + (size_t)t * cap_of(range);
    const uint32_t n = nrec[t] < cap_of(range) ? nrec[t] : cap_of(range);
    BitWin win = {0xffffffffu, 0u, 0u};
    uint32_t by_hand = 0;
    for (;;) {
        if (pos >= hi) { // walked through the whole range without meeting its trace: the hypothesis for the next range fails
            atomicOr(&st->giveup, 4);
            nblk[t] = by_hand;
            return;
        }
        // is `pos` a block start this range's trace recorded?  (sorted)
        const uint32_t want = pos - lo;
        uint32_t a = 0, b = n;
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if ((uint32_t)tr[mid] < want) a = mid + 1;
            else b = mid;
        }
        if (a < n && (uint32_t)tr[a] == want) {
            // from here on the trace walked the true chain: a failed measurement behind this point is the stream's fault
            if (lastbrk[t] > (int)a) atomicOr(&st->giveup, 8);
            nblk[t] = by_hand + (nrec[t] - a);
            return;
        }
        int d;
        uint32_t used;
        if (!block_dev<false>(words, L, ac11, pos, win, nullptr, d, used)) { // unusual on the true chain
            atomicOr(&st->giveup, 16);
            nblk[t] = by_hand;
            return;
        }
        by_hand++;
        pos += used;
    }
}

// ---- prefix sums (three small kernels: per-tile sums, scan of the tile sums by one workgroup, per-tile scan + offset) -----------
constexpr int kTile = 1024;
__device__ __forceinline__ long long wg_inclusive_scan(long long v, long long *lds /* [16] */, long long &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    long long off = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; k++) {
        if (k < wave) off += lds[k];
        tot += lds[k];
    }
    __syncthreads();
    total = tot;
    return v + off;
}
__global__ __launch_bounds__(kTile) void scan_tile_sums_kernel(const int32_t *__restrict__ in, size_t n, long long *__restrict__ tile_sum) {
    __shared__ long long lds[16];
    const size_t i = (size_t)blockIdx.x * kTile + threadIdx.x;
    long long tot;
    (void)wg_inclusive_scan(i < n ? (long long)in[i] : 0ll, lds, tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(kTile) void scan_of_sums_kernel(long long *__restrict__ tile_sum, size_t ntiles, long long *__restrict__ grand_total) {
    __shared__ long long lds[16];
    long long carry = 0;
    for (size_t base = 0; base < ntiles; base += kTile) { // one workgroup: a few iterations at most (2 M ranges = 2 K tiles)
        const size_t i = base + threadIdx.x;
        long long tot;
        const long long v = i < ntiles ? tile_sum[i] : 0ll;
        const long long inc = wg_inclusive_scan(v, lds, tot);
        if (i < ntiles) tile_sum[i] = carry + inc - v; // exclusive
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
// out[i] = (INCLUSIVE ? in[0..i] : in[0..i-1]) summed, as int64 narrowed to the output type by the caller's functor
template <bool INCLUSIVE, typename Out>
__global__ __launch_bounds__(kTile) void scan_apply_kernel(const int32_t *__restrict__ in, size_t n, const long long *__restrict__ tile_off, Out out) {
    __shared__ long long lds[16];
    const size_t i = (size_t)blockIdx.x * kTile + threadIdx.x;
    long long tot;
    const long long v = i < n ? (long long)in[i] : 0ll;
    const long long inc = wg_inclusive_scan(v, lds, tot);
    if (i < n) out(i, tile_off[blockIdx.x] + (INCLUSIVE ? inc : inc - v));
}
struct StoreU32 {
    uint32_t *p;
    __device__ void operator()(size_t i, long long v) const { p[i] = (uint32_t)v; }
};
struct StoreDc {
    int16_t *zz;
    size_t m;
    DecStatus *st;
    __device__ void operator()(size_t i, long long v) const { // c[0] = sat16(np.cumsum(dc)[i]) (the host decoder's sat16)
        const long long s = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
        zz[i * 64] = (int16_t)s;
        if (i == m - 1) st->dc_out = (int)v; // (m = all blocks: the differences behind the last block produced are 0)
    }
};

__global__ __launch_bounds__(64) void dec_decode_kernel(const uint32_t *__restrict__ words, const DecLutsDev *__restrict__ L, uint32_t nranges,
                                                        const uint32_t *__restrict__ nblk, const uint32_t *__restrict__ pstart,
                                                        const uint32_t *__restrict__ first_blk, const long long *__restrict__ total_blocks,
                                                        unsigned long long n_want, int16_t *__restrict__ zz, int32_t *__restrict__ dcdiff,
                                                        DecStatus *__restrict__ st) {
    __shared__ uint16_t ac11[2048];
    __shared__ __attribute__((aligned(16))) int16_t img[64][72]; // a block per lane (64 coefficients + 8 of padding: the lanes' 16-byte
                                                                 // pieces fall on different bank groups)
    load_ac11(ac11, L);
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    const unsigned long long total = (unsigned long long)*total_blocks;
    const unsigned long long m = total < n_want ? total : n_want; // blocks produced here
    unsigned long long b = first_blk[t];
    uint32_t pos = pstart[t];
    int16_t *c = img[threadIdx.x];
    BitWin win = {0xffffffffu, 0u, 0u};
    const uint32_t cnt = nblk[t];
    for (uint32_t i = 0; i < cnt && b < m; i++, b++) {
#pragma unroll
        for (int q = 0; q < 8; q++) reinterpret_cast<uint4 *>(c)[q] = make_uint4(0u, 0u, 0u, 0u);
        int d;
        uint32_t used;
        if (!block_dev<true>(words, L, ac11, pos, win, c, d, used)) { // (the stitch measured this block: cannot fail)
            atomicOr(&st->giveup, 32);
            return;
        }
        dcdiff[b] = d;
        uint4 *dst = reinterpret_cast<uint4 *>(zz + b * 64ull);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = reinterpret_cast<const uint4 *>(c)[q]; // (entry 0 is written by the DC pass)
        pos += used;
        if (b == m - 1) {
            st->pos_out = pos;
            st->m = m;
        }
    }
}

} // namespace

size_t entropy_decode_gpu_work_bytes(size_t stream_bytes, size_t nblocks) {
    const size_t nbits = stream_bytes * 8;
    const size_t nranges = nbits / 512 + 2; // (the smallest range: most ranges, and the most room per stream bit)
    const size_t ntiles = (nranges > nblocks ? nranges : nblocks) / kTile + 2;
    return nranges * ((size_t)cap_of(512) * 2 + 8 * 4) + nblocks * 4 + ntiles * 8 * 2 + 16384; // (six 4-byte arrays per range; every piece is rounded up to 256 B)
}

hipError_t entropy_decode_gpu(const void *d_stream_words, size_t stream_bytes, size_t nblocks, const DecLutsDev *d_luts, void *d_work,
                              size_t work_bytes, int16_t *d_zz, DecStatus *d_status, int range_bits, hipStream_t stream) {
    const size_t nbits = stream_bytes * 8;
    if (range_bits != 512 && range_bits != 1024 && range_bits != 2048) return hipErrorInvalidValue;
    const uint32_t range = (uint32_t)range_bits;
    const uint32_t kCap = cap_of(range);
    const int kRange = range_bits;
    if (nbits < 128 + 2048 + (size_t)kRangeMax || nbits >= (1ull << 32) || nblocks == 0) return hipErrorInvalidValue;
    if (work_bytes < entropy_decode_gpu_work_bytes(stream_bytes, nblocks)) return hipErrorInvalidValue;
    const uint32_t fast_end = (uint32_t)(nbits - 2048); // a block may START on the fast path up to here (as in the host decoder)
    const uint32_t nranges = (uint32_t)((fast_end - 128 + kRange - 1) / kRange);
    const size_t ntiles_r = ((size_t)nranges + kTile - 1) / kTile, ntiles_b = (nblocks + kTile - 1) / kTile;
    // workspace carve-up
    char *w = (char *)d_work;
    auto take = [&](size_t bytes) { char *p = w; w += (bytes + 255) / 256 * 256; return (void *)p; };
    uint16_t *starts = (uint16_t *)take((size_t)nranges * kCap * 2);
    uint32_t *nrec = (uint32_t *)take((size_t)nranges * 4), *endpos = (uint32_t *)take((size_t)nranges * 4);
    int *lastbrk = (int *)take((size_t)nranges * 4);
    uint32_t *nblk = (uint32_t *)take((size_t)nranges * 4), *pstart = (uint32_t *)take((size_t)nranges * 4);
    uint32_t *first_blk = (uint32_t *)take((size_t)nranges * 4);
    int32_t *dcdiff = (int32_t *)take(nblocks * 4);
    long long *tiles_r = (long long *)take((ntiles_r + 1) * 8), *tiles_b = (long long *)take((ntiles_b + 1) * 8);
    long long *totals = (long long *)take(16);
    if ((size_t)(w - (char *)d_work) > work_bytes) return hipErrorInvalidValue;
    const uint32_t *words = (const uint32_t *)d_stream_words;
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(DecStatus), stream);
    if (e != hipSuccess) return e;
    const dim3 gr((nranges + 63) / 64), bl(64);
    hipLaunchKernelGGL(dec_measure_kernel, gr, bl, 0, stream, words, d_luts, fast_end, range, nranges, starts, nrec, endpos, lastbrk, d_status);
    hipLaunchKernelGGL(dec_stitch_kernel, gr, bl, 0, stream, words, d_luts, fast_end, range, nranges, starts, nrec, endpos, lastbrk, nblk, pstart, d_status);
    // first block of every range: exclusive scan of the true block counts
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles_r), dim3(kTile), 0, stream, (const int32_t *)nblk, (size_t)nranges, tiles_r);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(kTile), 0, stream, tiles_r, ntiles_r, totals);
    hipLaunchKernelGGL((scan_apply_kernel<false, StoreU32>), dim3((unsigned)ntiles_r), dim3(kTile), 0, stream, (const int32_t *)nblk, (size_t)nranges,
                       (const long long *)tiles_r, StoreU32{first_blk});
    e = hipMemsetAsync(dcdiff, 0, nblocks * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dec_decode_kernel, gr, bl, 0, stream, words, d_luts, nranges, nblk, pstart, first_blk, (const long long *)totals,
                       (unsigned long long)nblocks, d_zz, dcdiff, d_status);
    // np.cumsum of the DC differences over the blocks (blocks past the ones produced here hold 0 differences: their entry 0 is
    // overwritten by the host's tail, which continues from dc_out)
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles_b), dim3(kTile), 0, stream, (const int32_t *)dcdiff, nblocks, tiles_b);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(kTile), 0, stream, tiles_b, ntiles_b, totals + 1);
    hipLaunchKernelGGL((scan_apply_kernel<true, StoreDc>), dim3((unsigned)ntiles_b), dim3(kTile), 0, stream, (const int32_t *)dcdiff, nblocks,
                       (const long long *)tiles_b, StoreDc{d_zz, nblocks, d_status});
    return hipGetLastError();
}

} // namespace tic
