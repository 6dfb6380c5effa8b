// tic_entropy_dec_gpu.hip - Huffman + run-length decode of a long stream on the GPU (decode side of SURVEY.md section 8f:
// decode_huffman huffman.py:77-98, decode_run_length huffman.py:36-38, the block loop of decompress() codec.py:178-186, np.cumsum of
// the DC differences codec.py:53).
//
// The format has no restart markers: a decoder is one dependent chain from bit 128 to the end.  But Huffman streams
// re-synchronise: a decoder started at an arbitrary bit as if a block began there lands on true block starts within a block or two.
// The host decoder uses that on 16 threads (tic_entropy.cpp decode_parallel: 7.7 ms for a 4096^2 stream); this is the same idea on
// tens of thousands of lanes, and the coefficients never leave the device:
//   measure   a lane per RANGE of kRange stream bits: walks the symbols from the range's first bit as if a block started there (a
//             guess for every range but the first), one symbol per step, recording the first bit of every block that decoded cleanly
//             from a block-start guess to its EOB; an invalid prefix or a block of more than 63 coefficients does not restart the
//             walk: it goes on in the AC state (a walk in step with the true symbols stays in step; the next true EOB ends on a true
//             block start) and the block in work is not recorded.
//   stitch    a lane per range, all in parallel: HYPOTHESIS: the true chain enters range t where range t-1's trace ended.  From there
//             the lane measures blocks "by hand" (recording their first bits too) until it lands on a block start the range's own
//             trace recorded; from that entry on the trace IS the true chain (a block start carries no state), so the trace's end is
//             where the true chain enters range t+1 - which is the hypothesis for t+1.  Range 0 starts on a true block start, so if
//             EVERY lane finds its synchronisation point inside its range, induction makes every hypothesis true.  Otherwise: give up.
//   scan      exclusive prefix sum of the ranges' true block counts -> index of each range's first block.
//   bpos      a lane per range writes the first bit of each of its true blocks (by-hand ones, then the trace from the entry on).
//   decode    a lane per BLOCK decodes it from its first bit into the (zeroed) int16 [N][64] coefficient array, DC differences aside.
//   scan      inclusive prefix sum of the DC differences (np.cumsum), saturated into entry 0 of every block.
// The walks are one dependent chain of look-ups per lane: stream words and tables are staged in LDS (the codewords of 12-16 bits
// included: as look-ups in memory they stalled a whole wave in every second step), and the chain is kept short - a range for the
// measure kernel, a block for the decode kernel.  7 MB stream (4096^2 noise, q=50), rocprofv3: measure 282 -> 125 us, decode
// 102 -> 67 us against the first version of this file (profiles/r03_decoder.txt).
// Anything unusual ON THE TRUE CHAIN - an invalid prefix, more than 63 coefficients in a block, a range without a synchronisation
// point, a measurement that failed behind the synchronisation point - raises a flag and the caller decodes the whole stream on the
// host, whose bit-serial path reproduces the reference's behaviour on malformed streams (exactly the host parallel decoder's rule).
// Blocks that start in the last 2048 bits of the stream are left to the host as well (`m` blocks are produced here, with the read
// position and the running DC behind them).
#include <hip/hip_runtime.h>
#include <stddef.h>
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

// The stream bits a workgroup walks, staged in LDS.  A workgroup is one wave, its lanes own 64 consecutive ranges: one contiguous
// window of 64 ranges plus the longest block a lane may run into behind its range (kOver words).  The kernels are one dependent chain
// of look-ups per lane; with the window words fetched from global memory (round 3's first version: two dependent loads per ~5
// symbols, at a few waves per CU) that chain was memory latency: measure 241 us, decode 102 us for a 7 MB stream.  The words are
// staged byte-swapped (the stream is big-endian) with coalesced loads, one padding word per range so that the lanes - a range apart -
// fall on different banks.  A word outside the window (cannot happen by the bounds in the kernels) is read from memory.
constexpr uint32_t kOver = 66;                                   // 1,728 bits of the longest block + the 64-bit window, in words
constexpr uint32_t kStageMax = 64 * (kRangeMax / 32) + kOver;    // window words of the longest range
constexpr uint32_t kStageLds = kStageMax + kStageMax / 16 + 2;   // ... with padding (the shortest range pads most)
struct Bits {
    const uint32_t *lds, *glob;
    uint32_t wbase, wcount, sh, nwords; // first word of the window, its length, log2(words per range); words of the stream
};
__device__ __forceinline__ uint32_t word_be(const Bits &s, uint32_t wi) {
    const uint32_t r = wi - s.wbase;
    if (r < s.wcount) return s.lds[r + (r >> s.sh)];
    return wi < s.nwords ? __builtin_bswap32(s.glob[wi]) : 0u;
}
__device__ __forceinline__ Bits stage_words(uint32_t *lds, const uint32_t *__restrict__ words, uint32_t first_word, uint32_t wcount, uint32_t sh, uint32_t nwords) {
    Bits s;
    s.lds = lds;
    s.glob = words;
    s.wbase = first_word;
    s.wcount = wcount;
    s.sh = sh;
    s.nwords = nwords;
    for (uint32_t r = threadIdx.x; r < wcount; r += blockDim.x) {
        const uint32_t wi = first_word + r;
        const uint32_t v = wi < nwords ? __builtin_bswap32(words[wi]) : 0u;
        lds[r + (r >> sh)] = v;
        if (r != 0u && (r & ((1u << sh) - 1u)) == 0u) lds[r + (r >> sh) - 1u] = v; // the padding word in front of a row repeats the row's
                                                                                   // first word: word r + 1 always sits right behind word r
    }
    __syncthreads();
    return s;
}
// the window of 64 consecutive ranges from bit first_bit (128 + 64 k range: a multiple of 32)
__device__ __forceinline__ Bits stage_bits(uint32_t *lds, const uint32_t *__restrict__ words, uint32_t first_bit, uint32_t range, uint32_t nwords) {
    return stage_words(lds, words, first_bit >> 5, 64u * (range >> 5) + kOver, range == 512u ? 4u : (range == 1024u ? 5u : 6u), nwords);
}
// 32 stream bits (MSB first) from bit `pos`; the two words around it are cached in registers and refetched when the position
// leaves them (a symbol is 5-8 bits on average: one refetch per ~5 symbols).
struct BitWin {
    uint32_t idx, a, b;
};
__device__ __forceinline__ uint32_t peek32(const Bits &words, uint32_t pos, BitWin &c) {
    const uint32_t wi = pos >> 5;
    if (wi != c.idx) {
        c.idx = wi;
        c.a = word_be(words, wi);
        c.b = word_be(words, wi + 1);
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

constexpr int kLongFirst = 0xff40, kLongCodes = 0x10000 - kLongFirst; // ac16 entries of the 11-bit prefixes 0x7fa..0x7ff
constexpr int kLutLds = 4096 + kLongCodes;
__device__ __forceinline__ uint32_t long_code(const uint16_t *lut, uint32_t pk) {
    const uint32_t i = (pk >> 16) - (uint32_t)kLongFirst;
    return i < (uint32_t)kLongCodes ? lut[4096u + i] : 0u;
}

// One block on the table-driven fast path: the device form of block_fast() in tic_entropy.cpp (same tables, same rules).  STORE: the
// coefficients 1..63 go to c (zeroed by the caller).  Returns false on anything unusual with nothing consumed.
template <bool STORE>
__device__ __forceinline__ bool block_dev(const Bits &words, const DecLutsDev *__restrict__ L, const uint16_t *lut /* LDS: dc11, ac11 */,
                                          uint32_t pos0, BitWin &win, int16_t *c, int &dc_diff, uint32_t &used) {
    const uint16_t *ac11 = lut + 2048;
    uint32_t pos = pos0;
    uint32_t pk = peek32(words, pos, win);
    uint32_t e = lut[pk >> 21];
    if (!e) return false; // DC categories are at most 9 bits long
    int len = (int)(e >> 8), size = (int)(e & 15u);
    dc_diff = value_of(pk, len, size);
    pos += (uint32_t)(len + size);
    int k = 1;
    for (;;) {
        pk = peek32(words, pos, win);
        e = ac11[pk >> 21];
        if (!e) e = long_code(lut, pk);
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

// dc11 and ac11 (adjacent in DecLutsDev) into LDS: lut[0..2047] = DC, lut[2048..4095] = AC, lut[4096..4287] = long AC codewords
__device__ __forceinline__ void load_lut(uint16_t *lds, const DecLutsDev *__restrict__ L) {
    static_assert(offsetof(DecLutsDev, ac11) == offsetof(DecLutsDev, dc11) + 4096, "dc11 and ac11 are adjacent");
    for (int i = threadIdx.x; i < 4096 / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(L->dc11)[i];
    // ... and the AC codewords of 12 to 16 bits: their 11-bit prefixes are 0x7fa..0x7ff (the fixed AC table is a complete prefix code:
    // every other prefix resolves in ac11), i.e. entries 0xff40..0xffff of ac16.  One symbol in a hundred at q = 50 - but with 64 lanes
    // side by side every second step has one, and as a look-up in the 128 KB table in memory it stalled the whole wave for a microsecond.
    for (int i = threadIdx.x; i < kLongCodes; i += blockDim.x) lds[4096 + i] = L->ac16[kLongFirst + i];
    __syncthreads();
}

// The measure kernel walks one SYMBOL per step, the lanes of a wave side by side.  block_dev() above is a loop per block: the lanes of
// a wave then wait for each other at every block end (a wave-step lasts as long as its longest block) - 630 symbol steps per wave
// where the longest lane has ~250 symbols.  In the measure kernel the position inside the block (k: 0 = the DC category comes next)
// is lane state and a block end is just another step.  Same tables, same rules as block_dev().
__global__ __launch_bounds__(64) void dec_measure_kernel(const uint32_t *__restrict__ gwords, uint32_t nwords, const DecLutsDev *__restrict__ L, uint32_t fast_end,
                                                         uint32_t range, uint32_t nranges, uint16_t *__restrict__ starts, uint32_t *__restrict__ nrec,
                                                         uint32_t *__restrict__ endpos, int *__restrict__ lastbrk, DecStatus *__restrict__ st) {
    __shared__ uint16_t lut[kLutLds];
    __shared__ uint32_t sbits[kStageLds];
    load_lut(lut, L);
    const Bits words = stage_bits(sbits, gwords, 128u + blockIdx.x * 64u * range, range, nwords);
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    const uint32_t lo = 128u + t * range;
    const uint32_t hi = lo + range < fast_end ? lo + range : fast_end;
    // One symbol per step, state updated by selects (the compiler's version of the same loop with if / else had ~25 branches per
    // step: 40 vector + 40 scalar instructions).  Every read stays inside the staged window: a block that starts in front of `hi`
    // ends within 1,728 bits of it, a walk that goes on after an incident is cut at hi + 1,800, and the window reaches 2,112 bits
    // (kOver words) behind the workgroup's last range.
    const uint32_t stop = hi + 1800u;
    uint32_t pos = lo, bstart = lo, cnt = 0; // a block that STARTS in front of `hi` is measured to its end
    int k = 0, brk = -1;   // k: scan position the next AC symbol starts from; 0 = the DC category comes next
    bool clean = true;     // the block in work began at a block-start guess and has decoded without an incident so far
    bool live = pos < hi;
    // The stream words under the read position sit in registers (wa, wb) and the word behind them (wc) is fetched a step ahead: the
    // walk is sequential, so the only LDS access left on the lane's dependent chain is the table look-up.
    auto word_at = [&](uint32_t r) { return sbits[r + (r >> words.sh)]; };
    uint32_t wi = (pos >> 5) - words.wbase; // window-relative index of the word `pos` lies in
    uint32_t wa = word_at(wi), wb = word_at(wi + 1u), wc = word_at(wi + 2u);
    while (live) {
        const uint32_t sh = pos & 31u;
        const uint32_t pk = sh ? __builtin_amdgcn_alignbit(wa, wb, 32u - sh) : wa; // 32 stream bits from `pos`
        uint32_t e = lut[(k ? 2048u : 0u) + (pk >> 21)];
        if (__any(e == 0u && k != 0)) { // a codeword of 12 to 16 bits somewhere in the wave
            const uint32_t e2 = long_code(lut, pk);
            e = (e == 0u && k != 0) ? e2 : e;
        }
        const bool nocode = e == 0u, dc = k == 0;
        const bool eob = !dc && !nocode && (e & 0xffu) == 0u;
        const int k_at = k + (int)((e >> 4) & 15u);
        // An incident: an invalid prefix, or more than 63 coefficients in the block.  The walk is a guess that led nowhere (or, behind
        // the point of synchronisation, the stream is malformed).  It goes on from here IN THE AC STATE: most of a block is AC symbols, a
        // walk that has fallen into step with the true symbols stays in step (an overflowing symbol is consumed, an invalid prefix
        // skips a bit), and the next true EOB then ends on a true block start.  The block in work is not recorded; the one behind its
        // EOB is a fresh guess.  (The first version went back to the failed block's first bit + 1 and took that for a block start:
        // every incident threw away up to 63 symbols of walking and the alignment they had reached - the unluckiest lane of a wave
        // walked 750 symbols for the 210 of its range.)
        const bool bad = nocode || (!dc && !eob && k_at > 63);
        pos += nocode ? 1u : (e >> 8) + (e & 15u);
        { // a step consumes at most 27 bits: at most one word boundary is crossed
            const bool crossed = ((pos >> 5) - words.wbase) != wi;
            wa = crossed ? wb : wa;
            wb = crossed ? wc : wb;
            wi += crossed ? 1u : 0u;
            wc = word_at(wi + 2u); // (not needed before the next boundary: off the dependent chain)
        }
        const bool rec = eob && clean;
        if (rec && cnt < cap_of(range)) starts[(size_t)t * cap_of(range) + cnt] = (uint16_t)(bstart - lo);
        cnt += rec ? 1u : 0u;
        brk = bad ? (int)cnt : brk;
        clean = bad ? false : (eob ? true : clean);
        bstart = eob ? pos : bstart;
        k = bad ? 1 : (eob ? 0 : (dc ? 1 : k_at + 1));
        live = (eob ? pos < hi : true) && pos < stop;
    }
    if (t == 0u && brk >= 0) atomicOr(&st->giveup, 1); // an incident on the true chain itself: unusual
    if (cnt > cap_of(range)) atomicOr(&st->giveup, 2); // (cannot happen: a block has at least 6 bits)
    nrec[t] = cnt;
    endpos[t] = pos;
    lastbrk[t] = brk;
}

__global__ __launch_bounds__(64) void dec_stitch_kernel(const uint32_t *__restrict__ gwords, uint32_t nwords, const DecLutsDev *__restrict__ L, uint32_t fast_end,
                                                        uint32_t range, uint32_t nranges, const uint16_t *__restrict__ starts, const uint32_t *__restrict__ nrec,
                                                        const uint32_t *__restrict__ endpos, const int *__restrict__ lastbrk,
                                                        uint32_t *__restrict__ nblk, uint16_t *__restrict__ hand,
                                                        uint32_t *__restrict__ entry, DecStatus *__restrict__ st) {
    __shared__ uint16_t lut[kLutLds];
    __shared__ uint32_t sbits[kStageLds];
    load_lut(lut, L);
    const Bits words = stage_bits(sbits, gwords, 128u + blockIdx.x * 64u * range, range, nwords);
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    entry[t] = nrec[t]; // (until the walk below meets the trace: no trace block belongs to the true chain)
    if (t == 0u) {
        nblk[0] = nrec[0];
        entry[0] = 0u;
        return;
    }
    const uint32_t lo = 128u + t * range;
    const uint32_t hi = lo + range < fast_end ? lo + range : fast_end;
    uint32_t pos = endpos[t - 1]; // hypothesis: where the true chain enters this range
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
            entry[t] = a;
            return;
        }
        int d;
        uint32_t used;
        if (by_hand < cap_of(range)) hand[(size_t)t * cap_of(range) + by_hand] = (uint16_t)want; // first bit of the by-hand block (pos >= lo: the walk enters behind the range before)
        if (!block_dev<false>(words, L, lut, pos, win, nullptr, d, used)) { // unusual on the true chain
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

// First bit of every block of the true chain: range t's blocks are the ones its stitch walked by hand, then its trace from the entry
// on; first_blk[t] (the scan of the counts) is the index of the first of them.
__global__ __launch_bounds__(64) void dec_bpos_kernel(uint32_t range, uint32_t nranges, const uint16_t *__restrict__ starts, const uint16_t *__restrict__ hand,
                                                      const uint32_t *__restrict__ nrec, const uint32_t *__restrict__ entry, const uint32_t *__restrict__ nblk,
                                                      const uint32_t *__restrict__ first_blk, unsigned long long nblocks, uint32_t *__restrict__ bpos) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= nranges) return;
    const uint32_t lo = 128u + t * range, cap = cap_of(range);
    const uint32_t nr = nrec[t] < cap ? nrec[t] : cap, a = entry[t] < nr ? entry[t] : nr;
    const uint32_t nb = nblk[t], from_trace = nr - a < nb ? nr - a : nb, by_hand = nb - from_trace;
    const unsigned long long first = first_blk[t];
    for (uint32_t i = 0; i < by_hand && i < cap && first + i < nblocks; i++) bpos[first + i] = lo + (uint32_t)hand[(size_t)t * cap + i];
    for (uint32_t j = 0; j < from_trace && first + by_hand + j < nblocks; j++) bpos[first + by_hand + j] = lo + (uint32_t)starts[(size_t)t * cap + a + j];
}

// A lane per BLOCK: lane b decodes the block at bpos[b] straight into the (zeroed) int16 [N][64] array: non-zero coefficients only,
// entry 0 is written by the DC pass.  The 256 consecutive blocks of a workgroup are one contiguous piece of the stream, staged in LDS
// (up to kBlkWin words: 500 bits per block on average; what lies behind is read from memory).  History: the first version decoded a
// RANGE per lane, the blocks of its range one after the other through an LDS image - 880 waves for a 7 MB stream, one per SIMD, each
// a chain of ~500 dependent symbol steps: 102-148 us.  A wave per 64 blocks with an image per lane (26 KB of LDS per wave, six waves
// per CU): 108 us.  Without the image, four waves sharing tables and window: 24 waves per CU.
constexpr int kDecodeWG = 256;
constexpr uint32_t kBlkWin = 4096 + kOver;
constexpr uint32_t kBlkLds = kBlkWin + kBlkWin / 32 + 2;
__global__ __launch_bounds__(kDecodeWG) void dec_decode_kernel(const uint32_t *__restrict__ gwords, uint32_t nwords, const DecLutsDev *__restrict__ L,
                                                               const uint32_t *__restrict__ bpos, const long long *__restrict__ total_blocks,
                                                               unsigned long long n_want, int16_t *__restrict__ zz /* zeroed */, int32_t *__restrict__ dcdiff,
                                                               DecStatus *__restrict__ st) {
    __shared__ uint16_t lut[kLutLds];
    __shared__ uint32_t sbits[kBlkLds];
    const unsigned long long total = (unsigned long long)*total_blocks;
    const unsigned long long m = total < n_want ? total : n_want; // blocks produced here
    const unsigned long long b0 = (unsigned long long)blockIdx.x * kDecodeWG, b = b0 + threadIdx.x;
    if (b0 >= m) return; // (the whole workgroup)
    load_lut(lut, L);
    const unsigned long long last = b0 + kDecodeWG - 1 < m - 1 ? b0 + kDecodeWG - 1 : m - 1;
    const uint32_t w0 = bpos[b0] >> 5, w1 = bpos[last] >> 5;
    const uint32_t want = w1 >= w0 ? w1 - w0 + kOver : kOver;
    const Bits words = stage_words(sbits, gwords, w0, want < kBlkWin ? want : kBlkWin, 5u, nwords);
    if (b >= m) return;
    BitWin win = {0xffffffffu, 0u, 0u};
    const uint32_t pos = bpos[b];
    int d;
    uint32_t used;
    if (!block_dev<true>(words, L, lut, pos, win, zz + b * 64ull, d, used)) { // (measure or stitch walked this block: cannot fail)
        atomicOr(&st->giveup, 32);
        return;
    }
    dcdiff[b] = d;
    if (b == m - 1) {
        st->pos_out = pos + used;
        st->m = m;
    }
}

} // namespace

size_t entropy_decode_gpu_work_bytes(size_t stream_bytes, size_t nblocks) {
    const size_t nbits = stream_bytes * 8;
    const size_t nranges = nbits / 512 + 2; // (the smallest range: most ranges, and the most room per stream bit)
    const size_t ntiles = (nranges > nblocks ? nranges : nblocks) / kTile + 2;
    return nranges * ((size_t)cap_of(512) * 2 * 2 + 9 * 4) + nblocks * 8 + ntiles * 8 * 2 + 16384; // (two traces and seven 4-byte arrays per range, two per block; every piece is rounded up to 256 B)
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
    uint16_t *starts = (uint16_t *)take((size_t)nranges * kCap * 2), *hand = (uint16_t *)take((size_t)nranges * kCap * 2);
    uint32_t *entry = (uint32_t *)take((size_t)nranges * 4), *bpos = (uint32_t *)take(nblocks * 4);
    uint32_t *nrec = (uint32_t *)take((size_t)nranges * 4), *endpos = (uint32_t *)take((size_t)nranges * 4);
    int *lastbrk = (int *)take((size_t)nranges * 4);
    uint32_t *nblk = (uint32_t *)take((size_t)nranges * 4);
    uint32_t *first_blk = (uint32_t *)take((size_t)nranges * 4);
    int32_t *dcdiff = (int32_t *)take(nblocks * 4);
    long long *tiles_r = (long long *)take((ntiles_r + 1) * 8), *tiles_b = (long long *)take((ntiles_b + 1) * 8);
    long long *totals = (long long *)take(16);
    if ((size_t)(w - (char *)d_work) > work_bytes) return hipErrorInvalidValue;
    const uint32_t *words = (const uint32_t *)d_stream_words;
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(DecStatus), stream);
    if (e != hipSuccess) return e;
    const dim3 gr((nranges + 63) / 64), bl(64);
    const uint32_t nwords = (uint32_t)((stream_bytes + 3) / 4); // (the caller zero-pads the last word and keeps 16 bytes behind it)
    hipLaunchKernelGGL(dec_measure_kernel, gr, bl, 0, stream, words, nwords, d_luts, fast_end, range, nranges, starts, nrec, endpos, lastbrk, d_status);
    hipLaunchKernelGGL(dec_stitch_kernel, gr, bl, 0, stream, words, nwords, d_luts, fast_end, range, nranges, starts, nrec, endpos, lastbrk, nblk, hand, entry, d_status);
    // first block of every range: exclusive scan of the true block counts
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles_r), dim3(kTile), 0, stream, (const int32_t *)nblk, (size_t)nranges, tiles_r);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(kTile), 0, stream, tiles_r, ntiles_r, totals);
    hipLaunchKernelGGL((scan_apply_kernel<false, StoreU32>), dim3((unsigned)ntiles_r), dim3(kTile), 0, stream, (const int32_t *)nblk, (size_t)nranges,
                       (const long long *)tiles_r, StoreU32{first_blk});
    e = hipMemsetAsync(dcdiff, 0, nblocks * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_zz, 0, nblocks * 128, stream); // the decode kernel writes the non-zero coefficients only
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dec_bpos_kernel, gr, bl, 0, stream, range, nranges, (const uint16_t *)starts, (const uint16_t *)hand, (const uint32_t *)nrec,
                       (const uint32_t *)entry, (const uint32_t *)nblk, (const uint32_t *)first_blk, (unsigned long long)nblocks, bpos);
    hipLaunchKernelGGL(dec_decode_kernel, dim3((unsigned)((nblocks + kDecodeWG - 1) / kDecodeWG)), dim3(kDecodeWG), 0, stream, words, nwords, d_luts, (const uint32_t *)bpos,
                       (const long long *)totals, (unsigned long long)nblocks, d_zz, dcdiff, d_status);
    // np.cumsum of the DC differences over the blocks (blocks past the ones produced here hold 0 differences: their entry 0 is
    // overwritten by the host's tail, which continues from dc_out)
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)ntiles_b), dim3(kTile), 0, stream, (const int32_t *)dcdiff, nblocks, tiles_b);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(kTile), 0, stream, tiles_b, ntiles_b, totals + 1);
    hipLaunchKernelGGL((scan_apply_kernel<true, StoreDc>), dim3((unsigned)ntiles_b), dim3(kTile), 0, stream, (const int32_t *)dcdiff, nblocks,
                       (const long long *)tiles_b, StoreDc{d_zz, nblocks, d_status});
    return hipGetLastError();
}

} // namespace tic
