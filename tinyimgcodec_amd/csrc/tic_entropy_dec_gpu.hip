// tic_entropy_dec_gpu.hip - Huffman + run-length decode of a long stream on the GPU (decode side of SURVEY.md section 8f:
// decode_huffman huffman.py:77-98, decode_run_length huffman.py:36-38, the block loop of decompress() codec.py:178-186, np.cumsum of
// the DC differences codec.py:53).
//
// The format has no restart markers: a decoder is one dependent chain from bit 128 to the end.  But Huffman streams
// re-synchronise: a decoder started at an arbitrary bit as if a block began there lands on true block starts within a block or two.
// The host decoder uses that on 16 threads (tic_entropy.cpp decode_parallel: 7.7 ms for a 4096^2 stream); this is the same idea on
// tens of thousands of lanes, and the coefficients never leave the device:
//   measure   a lane per RANGE of stream bits (3 average blocks, 544 ... 2,016): walks the symbols from the range's first bit as if a
//             block started there (a guess for every range but the first), one symbol per step, recording the first bit of every
//             block it walked from a block-start guess to its EOB; an invalid prefix or a block of more than 63 coefficients does not
//             restart the walk: it goes on (a walk in step with the true symbols stays in step; the next true EOB ends on a true
//             block start).
//   stitch    the same lanes, the same kernel, all in parallel: HYPOTHESIS: the true chain enters range t where range t-1's walk
//             ended (the lane next door walked it; lane 0 of a workgroup shadows the last range of the workgroup in front).  From there
//             the lane measures blocks "by hand" (recording their first bits too) until it lands on a block start the range's own
//             trace recorded; from that entry on the trace IS the true chain (a block start carries no state), so the trace's end is
//             where the true chain enters range t+1 - which is the hypothesis for t+1.  Range 0 starts on a true block start, so if
//             EVERY lane finds its synchronisation point inside its range, induction makes every hypothesis true.  Otherwise: give up.
//   block positions   still the same kernel: exclusive prefix sum of the ranges' true block counts -> index of each range's first block
//             (every workgroup publishes its own sum and adds up the sums of the workgroups in front of it, wave_lookback below); the
//             lane of a range then writes the first bit of each of its true blocks (by-hand ones, then the trace from the entry on).
//   decode + inverse transform   the second and last kernel, a workgroup per 256 consecutive blocks, a LANE PER BLOCK from start to end:
//             the lane decodes its block's DC symbol, the workgroup sums the DC differences and publishes the sum (np.cumsum, codec.py:53:
//             the same look-back, taken when the lane needs the value - at the block's end), the lane decodes the AC symbols, two per
//             table look-up where two codewords lie inside the window, into an LDS image of the block (natural order), reads the image
//             back into 64 registers, dequantises and inverse-transforms it in the reference's float64 operation order, and stores PIXELS.
//             The coefficients never exist in memory.
//             (Rounds 2-3: a decode kernel that scattered the non-zero coefficients into a zeroed int16 [N][64] array - 45 two-byte
//             stores per lane to 64 different lines each - then idct_kernel read the array back: 68 + 24 us and a 33.5 MB fill.  Round 4:
//             four launches - measure + stitch, counts scan + positions, DC scan, decode + inverse with 8 lanes per block and the 8x8
//             float64 matrix transposed through LDS: 111 us of kernels for a 4096^2 stream; now 83, profiles/r05_decoder.txt.)
// The walks are one dependent chain of look-ups per lane: stream words and tables are staged in LDS (the codewords of 12-16 bits
// included: as look-ups in memory they stalled a whole wave in every second step), and the chain is kept short - a range for the
// measure walk, a block for the decode phase.  7 MB stream (4096^2 noise, q=50), rocprofv3: measure 282 -> 125 us, decode
// 102 -> 67 us against the first version of this file (profiles/r03_decoder.txt); rounds 4 and 5: profiles/r04_decoder.txt, r05_decoder.txt.
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
#include "tic_hooks.h"
#include "tic_math.h"

namespace tic {
namespace {

// Stream bits per range: an ODD number of 32-bit words between 9 and 63 (288 ... 2,016 bits; the lanes' words in the LDS window are
// then an odd stride apart: no bank padding),
// chosen by the caller from the stream's average block length (a block is at most 64 x 27 = 1,728 bits long).  The kernels are
// latency-bound per lane (one dependent chain of table look-ups; the measure kernel takes the same time with a quarter of its grid,
// profiles/r04_decoder.txt), so their time goes with the bits a lane walks - hence ranges as short as the stitch tolerates: it needs
// every range to hold a block start of the true chain that the range's own walk recorded, and a range shorter than the stream's
// long blocks makes it give up (the caller then tries the longest range).
constexpr int kRangeMin = 288, kRangeMax = 2016;
__host__ __device__ constexpr uint32_t cap_of(uint32_t range) { return range / 6u + 2u; } // block starts a range can hold (a block has at least 6 bits: 2-bit DC code + EOB)

// The stream bits a workgroup walks, staged in LDS.  A workgroup is one wave, its lanes own 64 consecutive ranges: one contiguous
// window of 64 ranges plus the longest block a lane may run into behind its range (kOver words).  The kernels are one dependent chain
// of look-ups per lane; with the window words fetched from global memory (round 3's first version: two dependent loads per ~5
// symbols, at a few waves per CU) that chain was memory latency: measure 241 us, decode 102 us for a 7 MB stream.  The words are
// staged byte-swapped (the stream is big-endian) with coalesced loads, one padding word per range so that the lanes - a range apart -
// fall on different banks.  A word outside the window (cannot happen by the bounds in the kernels) is read from memory.
constexpr uint32_t kOver = 66;                                   // 1,728 bits of the longest block + the 64-bit window, in words
__host__ __device__ constexpr uint32_t stage_lds_words(uint32_t range) { return 64u * (range >> 5) + kOver + 2u; } // LDS words of a workgroup's window
struct Bits {
    const uint32_t *lds, *glob;
    uint32_t wbase, wcount, sh, nwords; // first word of the window, its length, log2(words per range); words of the stream
    uint32_t last_mask;                 // big-endian mask of the stream's last word (the bytes behind the stream's end read as zero)
};
// word wi of the stream, big-endian; zero behind the end, and the bytes of the last word that lie behind the stream's last byte are
// zero too - the stream is read where the caller holds it, not from a zero-padded copy
__device__ __forceinline__ uint32_t stream_word(const uint32_t *__restrict__ words, uint32_t wi, uint32_t nwords, uint32_t last_mask) {
    if (wi >= nwords) return 0u;
    const uint32_t v = __builtin_bswap32(words[wi]);
    return wi == nwords - 1u ? (v & last_mask) : v;
}
__device__ __forceinline__ uint32_t word_be(const Bits &s, uint32_t wi) {
    const uint32_t r = wi - s.wbase;
    if (r < s.wcount) return s.lds[r + (r >> s.sh)];
    return stream_word(s.glob, wi, s.nwords, s.last_mask);
}
// T = threads of the workgroup (compile time: with blockDim.x as the stride the compiler emitted one load, one wait, one LDS
// write per trip - 33 dependent trips to memory per lane for a 1,024-bit range, ~13 us of a kernel's start).  Eight words per lane
// are requested before the first is written; the loads are unconditional (the index is clamped, the value masked afterwards).
template <int T>
__device__ __forceinline__ Bits stage_words(uint32_t *lds, const uint32_t *__restrict__ words, uint32_t first_word, uint32_t wcount, uint32_t sh, uint32_t nwords,
                                            uint32_t last_mask, uint32_t tid /* 0 .. T-1: this thread among the T that stage this window */) {
    Bits s;
    s.lds = lds;
    s.glob = words;
    s.wbase = first_word;
    s.wcount = wcount;
    s.sh = sh;
    s.nwords = nwords;
    s.last_mask = last_mask;
    constexpr int kBatch = 8;
    const uint32_t lastw = nwords - 1u; // (a stream has its 16-byte header: nwords >= 4)
    for (uint32_t r0 = tid; r0 < wcount; r0 += kBatch * T) {
        uint32_t v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t wi = first_word + r0 + (uint32_t)(k * T);
            v[k] = words[wi < lastw ? wi : lastw];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t r = r0 + (uint32_t)(k * T), wi = first_word + r;
            uint32_t x = __builtin_bswap32(v[k]);
            x = wi < lastw ? x : (wi == lastw ? (x & last_mask) : 0u);
            if (r < wcount) {
                lds[r + (r >> sh)] = x;
                if (r != 0u && (r & ((1u << sh) - 1u)) == 0u) lds[r + (r >> sh) - 1u] = x; // the padding word in front of a row repeats the
                                                                                           // row's first word: word r + 1 always sits right behind word r
            }
        }
    }
    __syncthreads();
    return s;
}
// the window of 64 consecutive ranges from bit first_bit (128 + 64 k range: a multiple of 32)
template <int T>
__device__ __forceinline__ Bits stage_bits(uint32_t *lds, const uint32_t *__restrict__ words, uint32_t first_bit, uint32_t range, uint32_t nwords, uint32_t last_mask, uint32_t tid) {
    return stage_words<T>(lds, words, first_bit >> 5, 64u * (range >> 5) + kOver, 31u /* no padding */, nwords, last_mask, tid);
}
// value bits behind a codeword of `len` bits (bitbuffer.py:55-65): x with its top bit clear stands for x - (2^size - 1)
__device__ __forceinline__ int value_of(uint32_t pk, int len, int size) {
    if (size == 0) return 0;
    const uint32_t x = (pk << len) >> (32 - size);
    return (x >> (size - 1)) ? (int)x : (int)x - ((1 << size) - 1);
}

constexpr int kLongFirst = 0xff40, kLongCodes = 0x10000 - kLongFirst; // ac16 entries of the 11-bit prefixes 0x7fa..0x7ff
constexpr int kChainLds = 2048 + 4096 + 256; // DecLutsDev::mdc, mac, mlong (mlong's entries behind the 192 long codewords are zero: the slot of an index out of range)
// `n16` 16-byte pieces from memory to LDS with T threads: all of a lane's loads are in flight before its first LDS write
template <int T, int n16>
__device__ __forceinline__ void copy16_to_lds(void *lds, const void *__restrict__ src) {
    constexpr int kPer = (n16 + T - 1) / T;
    uint4 v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int j = (int)threadIdx.x + k * T;
        v[k] = reinterpret_cast<const uint4 *>(src)[j < n16 ? j : n16 - 1];
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int j = (int)threadIdx.x + k * T;
        if (j < n16) reinterpret_cast<uint4 *>(lds)[j] = v[k];
    }
}
// the same with the workgroup's size known at run time only (the measure kernel: one to four waves)
__device__ __forceinline__ void copy16_to_lds_rt(void *lds, const void *__restrict__ src, int n16) {
    for (int j0 = (int)threadIdx.x; j0 < n16; j0 += 4 * (int)blockDim.x) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = j0 + k * (int)blockDim.x;
            v[k] = reinterpret_cast<const uint4 *>(src)[j < n16 ? j : n16 - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = j0 + k * (int)blockDim.x;
            if (j < n16) reinterpret_cast<uint4 *>(lds)[j] = v[k];
        }
    }
}
static_assert(offsetof(DecLutsDev, dc11) % 16 == 0 && offsetof(DecLutsDev, ac11) % 16 == 0 && (offsetof(DecLutsDev, ac16) + 2 * kLongFirst) % 16 == 0 &&
                  (2 * kLongCodes) % 16 == 0,
              "the tables are copied in 16-byte pieces (the structure itself comes from hipMalloc)");
// ---- prefix sums inside the kernels that need them ---------------------------------------------------------------------------------
// The decoder has two: the ranges' block counts (-> index of each range's first block) and the blocks' DC differences (np.cumsum,
// codec.py:53).  Every workgroup sums its own elements, PUBLISHES the sum - one 8-byte agent-scope atomic store that carries the sum
// and the launch's epoch, so that nothing has to be zeroed between launches and no fence is needed: the word is the whole message -
// and then adds up the words of all workgroups in front of it, lane j reading workgroup j's, j + 64's ...  No workgroup waits for a
// RESULT of another, only for its publication, which every workgroup makes before it looks back; workgroups are dispatched in index
// order, so the ones a workgroup waits for are running or done.  (Rounds 2-3: three launches per scan - tile sums, a one-workgroup scan
// of the sums, apply; round 4: one launch per scan; now none - the sums are taken where their inputs are produced and their results used.)
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
// 24 bits of epoch, 40 bits of sum (two's complement): |sum| < 2^39 covers 2^32 blocks of +-2047 each... by a wide margin for every
// frame the 32-bit bit positions of this file admit
__device__ __forceinline__ unsigned long long scan_pack(uint32_t epoch, long long sum) {
    return ((unsigned long long)(epoch & 0xffffffu) << 40) | ((unsigned long long)sum & 0xffffffffffull);
}
__device__ __forceinline__ long long scan_sum_of(unsigned long long d) { return ((long long)(d << 24)) >> 24; }
__device__ __forceinline__ void scan_publish(unsigned long long *desc, uint32_t epoch, long long tot) {
    if (threadIdx.x == 0) __hip_atomic_store(&desc[blockIdx.x], scan_pack(epoch, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Sum of the workgroups in front of this one, taken by ONE WAVE (all 64 lanes call; `desc` = this sum's two word arrays, `half` words
// each: own sums, then inclusive sums).
// Launches of up to `flat_grid` workgroups (the caller's choice, 4,096; a 4096^2 stream: 1,024 and 1,350): lane l adds up the own sums of workgroups l, l + 64 ...,
// eight words requested before the first is looked at (one after the other they were 16 - 21 dependent trips to memory: 6 - 8 us at the
// end of a kernel); a word that does not carry this launch's epoch yet is read again until it does.
// Larger launches (16384^2: 16,384 and 21,600 workgroups - adding up all own sums would read 8.6 GB of them, quadratic in the
// workgroups): lane l looks at workgroup i - 1 - l, at its INCLUSIVE sum (everything up to and including it) if it has published one -
// the nearest such workgroup ends the look-back: the own sums of the workgroups between it and this one, waited for, plus its inclusive
// sum - and 64 workgroups further back when none of the 64 has one yet; then this workgroup's inclusive sum is published for the ones
// behind.  (Measured, profiles/r05_decoder.txt: the inclusive sums everywhere cost a 4096^2 stream 11 us - the workgroups of a launch's
// first round arrive here together, nobody has an inclusive sum yet, and every window is a trip to memory of its own; eight windows
// per trip: 19 us; without them a 16384^2 stream takes 1.15 ms instead of 0.95.)
__device__ __forceinline__ long long wave_lookback(unsigned long long *desc, uint32_t half, uint32_t flat_grid, uint32_t epoch, long long own_total, DecStatus *st,
                                                   uint32_t tile /* this wave's (or workgroup's) index among the `ntiles` that publish sums */, uint32_t ntiles) {
    const uint32_t lane = threadIdx.x & 63u, ep = epoch & 0xffffffu, nfront = tile;
    const unsigned long long *own = desc;
    // (an exit every wave reaches: a workgroup in front that never publishes - which the dispatch order rules out - ends the wait
    // after about a second, and the flag sends the whole stream to the host decoder)
    auto wait_for = [&](unsigned long long d, uint32_t j) {
        uint32_t spins = 0;
        while ((uint32_t)(d >> 40) != ep) {
            d = __hip_atomic_load(&own[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins == (1u << 20)) {
                atomicOr(&st->giveup, 128);
                d = scan_pack(epoch, 0);
            }
        }
        return scan_sum_of(d);
    };
    long long sum = 0;
    if (ntiles <= flat_grid) {
        for (uint32_t j0 = lane; j0 < nfront; j0 += 64u * 8u) {
            unsigned long long d[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t j = j0 + 64u * k;
                d[k] = __hip_atomic_load(&own[j < nfront ? j : nfront - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t j = j0 + 64u * k;
                if (j < nfront) sum += wait_for(d[k], j);
            }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) sum += __shfl_xor(sum, sh, 64);
        return sum;
    }
    unsigned long long *incl = desc + half;
    uint32_t j_hi = nfront; // workgroups [0, j_hi) are still to be accounted for
    while (j_hi > 0u) {
        const bool in = lane < j_hi;
        const uint32_t j = in ? j_hi - 1u - lane : 0u;
        const unsigned long long a = __hip_atomic_load(&own[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long c = __hip_atomic_load(&incl[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long found = __ballot(in && (uint32_t)(c >> 40) == ep);
        const uint32_t f = found ? (uint32_t)__builtin_ctzll(found) : 64u; // the nearest workgroup with an inclusive sum
        long long v = in && lane < f ? wait_for(a, j) : (lane == f ? scan_sum_of(c) : 0ll);
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh, 64);
        sum += v;
        if (f < 64u) break;
        j_hi = j_hi > 64u ? j_hi - 64u : 0u;
    }
    if (lane == 0u) __hip_atomic_store(&incl[tile], scan_pack(epoch, sum + own_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return sum;
}
// One step of a chain walk: the look-up for the 32 stream bits `pk` in the chain tables (mdc | mac | mlong, adjacent in LDS) - the next
// 11 bits with the DC category next (mode 0), the next 12 inside the AC symbols (mode 1), the next 16 - 0xff40 when the step before met
// the prefix of a long codeword (mode 2) - and what follows from the entry: the bits to advance, whether a block ended, the next mode.
// Integer state and arithmetic: with the state in two booleans and the index in nested conditions the compiler kept the booleans as lane
// masks in scalar registers, merged under the loop's exec mask at every step, and built three exec-mask regions out of the index - 75
// instructions per step where this takes 50.  (The kernel's time did not move, 38.8 us: the walk is 16.7 us of it - a timing build that
// walks twice - at one wave per SIMD; profiles/r05_decoder.txt.)
// A window without a codeword (of at most 11 bits) has an entry like any other: inside the AC symbols it is the prefix of a long codeword - no advance,
// mode 2; with the DC category next or in the long table it is no codeword at all - skip a bit (a walk that is out of step, or a damaged stream).
__device__ __forceinline__ void chain_step(const uint8_t *lutm, uint32_t pk, uint32_t &mode, uint32_t &adv, uint32_t &eob) {
    const uint32_t shift = mode == 2u ? 16u : 21u - mode;
    const uint32_t base = mode == 2u ? 6144u - (uint32_t)kLongFirst : mode << 11; // (wraps: a long codeword's 16 bits are >= 0xff40)
    const uint32_t idx = base + (pk >> shift);
    const uint32_t e = lutm[idx < 6144u + (uint32_t)kLongCodes ? idx : 6144u + (uint32_t)kLongCodes]; // (never out of range by the walk's own rules; the slot behind the tables holds a zero)
    eob = e & 1u;         // (bits to advance << 3) | (the next step's table << 1) | the chain ends with EOB: tic_entropy.cpp chain_entry - a window
    adv = e >> 3;         // without a codeword has an entry too (skip a bit; or, inside the AC symbols, the prefix of a long codeword: advance nothing,
    mode = (e >> 1) & 3u; // table 2 next), so the step derives nothing
}
// Diagnostic build only (make -C tools bin/libvar_700.so VARSRC=tic_entropy_dec_gpu.hip; tools/dec_stamps.py): the 100 MHz clock at the phase boundaries of the two
// kernels, taken by lane 0 of the first and of the last wave / workgroup of a launch - where a kernel that is one chain of trips to memory spends its time
#if defined(TIC_EXP) && TIC_EXP == 700
__device__ unsigned long long g_dec_stamps[4][16];
#define DEC_STAMP(row_first, is_first, is_last, k)                                                                               \
    do {                                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                               \
        if ((threadIdx.x & 63u) == 0u && ((is_first) || (is_last))) g_dec_stamps[(row_first) + ((is_first) ? 0 : 1)][k] = wall_clock64(); \
    } while (0)
#else
#define DEC_STAMP(row_first, is_first, is_last, k) do {} while (0)
#endif
// Measure and stitch, one kernel.  The lanes of a wave walk side by side, a look-up per step.  (Round 3's first version looped per
// block: the lanes of a wave then wait for each other at every block end - 630 symbol steps per wave where the longest lane has ~250
// symbols.)  Here the position inside the block (is the DC category next?) is lane state and a block end is just another step.
//
// A wave stands alone (a workgroup is one to four of them, sharing only the tables' trip from memory: 38.5 -> 36.0 us for a 4096^2 stream).
// Lanes kShadow..63 own 64 - kShadow consecutive ranges; lanes 0..kShadow-1 SHADOW the ranges in front of them - each walks its range exactly as
// its owner (a lane of the workgroup before) does, and lanes 1.. also stitch it as the owner does, writing nothing but the trace (the
// same words the owner writes) - so that every owner finds the exit of the range in front of its own in the lane next to it: the stitch
// needs no second launch (its start, tables and window staged again, was a third of it) and no workgroup waits for another.
// Round 6: FOUR shadows instead of one.  The true chain does not always enter a range where the walk of the range in front ended: a
// block longer than a range passes over whole ranges (sparse streams - smooth, flat, blocky content at 8-15 bits per block - hold blocks
// of 600 bits and more among thousands of 8: rounds 2-5 gave up on such a stream and decoded it again with the longest range, which is
// why streams below 32 bits per block were kept on the host decoder).  Now a range the chain passes over, or walks through without
// meeting its trace, hands its EXIT to the lane behind it and that lane stitches again from there (the fix-up loop below); a block is at
// most 1,728 bits long, a range at least 544: four ranges in front of the first owned one are enough for the true entry to reach it.
// (later in round 6: the count is the launch's - `shadows`, dec_shadows() - so that ranges below 544 bits get EIGHT: the same 2,300 bits in front of the
//  first owned range for the true entry to arrive through)
constexpr uint32_t kEveryStepTiles = 128; // launches of at most this many waves store a trace word at every step of the walk (below)
constexpr uint32_t kShadowMax = 16;        // ranges in front of its own that a wave shadows: at most (sizes the look-back words)
template <bool kStoreEveryStep>
__device__ __forceinline__ void measure_stitch_body(const uint32_t *__restrict__ gwords, uint32_t nwords, uint32_t last_mask, const DecLutsDev *__restrict__ L,
                                                    uint32_t fast_end, uint32_t stream_bits, uint32_t range, uint32_t nranges, uint16_t *__restrict__ starts,
                                                    uint16_t *__restrict__ hand, unsigned long long *__restrict__ desc, uint32_t desc_half, uint32_t flat_grid, uint32_t epoch,
                                                    unsigned long long nblocks, uint32_t *__restrict__ bpos, long long *__restrict__ grand_total,
                                                    uint32_t ntiles, DecStatus *__restrict__ st, const uint32_t tile /* this wave's index among the stream's `ntiles` waves */, const uint32_t stitch_rounds,
                                                    const uint32_t kShadow /* ranges in front of its own that the wave shadows */) {
    __shared__ __attribute__((aligned(16))) uint8_t lutm[kChainLds];   // the chain tables: the measure walk
    extern __shared__ uint32_t sbits_all[]; // stage_lds_words(range) per wave, the launch's dynamic LDS
    // A workgroup is one to four WAVES that share nothing but the chain tables (6.4 KB from memory once per workgroup instead of once
    // per wave) and the barrier behind the staging; `tile` is the wave's index among all waves of the launch - what rounds 4's
    // one-wave workgroups called blockIdx.x.
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t kOwned = 64u - kShadow; // ranges a wave owns
    uint32_t *sbits = sbits_all + wave * stage_lds_words(range);
#define MS_STAMP(k) DEC_STAMP(0, tile == 0u, tile + 1u == ntiles, k)
    MS_STAMP(0);
    if (tile == 0u && lane < 4u) st->head[lane] = gwords[lane]; // (a stream has its 16-byte header: nwords >= 4)
    const uint32_t t_first = tile ? tile * kOwned - kShadow : 0u; // the window's first range
    static_assert(offsetof(DecLutsDev, mac) == offsetof(DecLutsDev, mdc) + 2048 && offsetof(DecLutsDev, mlong) == offsetof(DecLutsDev, mac) + 4096 &&
                      offsetof(DecLutsDev, mdc) % 16 == 0,
                  "the chain tables are adjacent and copied in 16-byte pieces");
    copy16_to_lds_rt(lutm, L->mdc, kChainLds / 16); // (stage_bits below ends with the barrier)
    const Bits words = stage_bits<64>(sbits, gwords, 128u + t_first * range, range, nwords, last_mask, lane);
    if (tile >= ntiles) return; // (a wave behind the last range; behind the barrier)
    MS_STAMP(1);
    const bool shadow = lane < kShadow;
    const uint32_t t = tile * kOwned + lane - kShadow; // (lanes 0..kShadow-1 of tile 0: no such range)
    const bool walks = (tile != 0u || !shadow) && t < nranges;
    const bool mine = walks && !shadow;
    const uint32_t lo = 128u + (walks ? t : 0u) * range;
    const uint32_t hi = lo + range < fast_end ? lo + range : fast_end;
    // ---- measure.  One symbol per step, state updated by selects (the compiler's version of the same loop with if / else had ~25
    // branches per step: 40 vector + 40 scalar instructions).  Every read stays inside the staged window: a block that starts in
    // front of `hi` ends within 1,728 bits of it, a walk that goes on after an incident is cut at hi + 1,800, and the window reaches
    // 2,112 bits (kOver words) behind the workgroup's last range.
    //
    // The walk keeps NO record of incidents.  An invalid prefix skips a bit, a block of more than 63 coefficients simply goes on to its
    // EOB, and every EOB records the first bit of the block it ends - whether that block was well-formed or not.  A recorded position is
    // a bit at which this lane stood with a block about to start; from such a bit the walk is a function of the stream alone, so a
    // position the TRUE chain shares with the trace has the true chain's future behind it, incident or not.  Whether the blocks of the
    // true chain are well-formed and follow each other without a gap is checked where they are decoded (the fused kernel: an invalid
    // prefix, a 64th coefficient, or a block that does not end where the next one starts raises the give-up flag).  (Rounds 2-3 tracked
    // the scan position, a clean flag and the last incident per lane here: the larger half of 100 instructions per symbol.)
    const uint32_t stop = hi + 1800u < stream_bits ? hi + 1800u : stream_bits; // (behind the stream's end there is nothing to walk: with margin_bits = 0 the last
                                                                                // range's walk would go through 1,800 zero bits there, 600 steps with the whole launch waiting)
    const uint32_t cap = cap_of(range);
    uint16_t *tr = starts + (size_t)(walks ? t : 0u) * cap; // (a shadow writes the very words the range's owner writes: its own stitch reads them back)
    uint32_t pos = lo, bstart = lo, cnt = 0; // a block that STARTS in front of `hi` is measured to its end
    uint32_t mode = 0; // chain_step: the DC category comes next (a block starts here)
    bool live = walks && pos < hi;
    // The stream words under the read position sit in registers (wa, wb, wc); the word that becomes wc when a step crosses a word
    // boundary is requested together with the table entry, at the top of the step, so that one wait covers both and the only LDS
    // access on the lane's dependent chain is the table look-up.
    //
    // A step consumes a CHAIN of symbols: the walk needs the end of every block, not the values, and a symbol's length is known from its
    // codeword alone - so one look-up of the next 12 bits (11 with the DC category next) gives the bits of every symbol whose codeword
    // lies inside them, up to and including an EOB (DecLutsDev::mdc / mac, tic_entropy.cpp build_chain: symbol by symbol the very steps
    // of a one-symbol walk).  2.3 symbols per step on noise at q = 50: the walk, one lane's dependent chain, is that much shorter.
    //
    // One look-up per step, always: a lane whose AC prefix resolves in the long table (one symbol in a hundred at q = 50) spends a
    // second STEP on it instead of a second look-up inside the step.  With 64 lanes side by side every second step had such a lane,
    // and the whole wave went through a branch, a second LDS round trip and a second wait for it.
    const uint32_t *wp = sbits - words.wbase; // wp[w] = word w of the stream (inside the window)
    uint32_t wi = pos >> 5;                   // the word `pos` lies in
    uint32_t wa = wp[wi], wb = wp[wi + 1u], wc = wp[wi + 2u];
    while (live) {
        const uint32_t wn = wp[wi + 3u];
        asm volatile("" ::: "memory"); // (keeps the request in front of the table look-up: it has returned when the entry has)
        const uint32_t pk = (uint32_t)(((((unsigned long long)wa) << 32) | wb) << (pos & 31u) >> 32); // 32 stream bits from `pos`
        uint32_t adv, eob;
        chain_step(lutm, pk, mode, adv, eob);
        pos += adv;
        { // a step consumes at most 27 bits: at most one word boundary is crossed
            const uint32_t now = pos >> 5;
            const bool crossed = now != wi;
            wa = crossed ? wb : wa;
            wb = crossed ? wc : wb;
            wc = crossed ? wn : wc;
            wi = now;
        }
        // A store at every step - into the trace's last place, which no range fills (a block has at least 6 bits: at most cap - 2 block starts), when no
        // block ends - is six instructions less than the branch around a store per block: a 512^2 stream's launch 22 us instead of 24.  A 4096^2 stream's
        // takes 64 instead of 33 that way - five million two-byte stores - so only launches of up to kEveryStepTiles waves do it (profiles/r06_decoder.txt (11)).
        if (kStoreEveryStep) tr[eob && cnt < cap ? cnt : cap - 1u] = (uint16_t)(bstart - lo);
        else if (eob && cnt < cap) tr[cnt] = (uint16_t)(bstart - lo); // (a lane is in this loop only if it walks)
        cnt += eob;
        bstart = eob ? pos : bstart;
        live = pos < (eob ? hi : stop); // a block that STARTS in front of `hi` is walked to its end (hi <= stop)
    }
    if (mine && cnt > cap) atomicOr(&st->giveup, 2); // (cannot happen: a block has at least 6 bits)
    MS_STAMP(2);
    // ---- stitch: does the true chain, entering where the walk of the range in front ended, meet this range's trace?
    const uint32_t wend = pos; // where this range's own walk ended: the true chain's exit IF the chain met the trace inside the range
    const uint32_t n_rec = cnt;
    const uint32_t n = n_rec < cap ? n_rec : cap;
    // what the stitch finds out about this range: its blocks on the true chain (nb): `by_hand` of them measured below, from where the
    // chain entered, the others the range's own trace from entry `a` on; and where the chain LEAVES the range (exit_pos)
    uint32_t nb = 0, by_hand = 0, a = n, exit_pos = wend;
    // the head of the trace, back from memory in one trip (this lane's own stores); entry 0 is the range's first bit
    const uint32_t tr1 = tr[1], tr2 = tr[2]; // (entries behind n are never looked at; cap >= 3)
    auto trace_at = [&](uint32_t k) { return k >= n ? 0xffffffffu : (k == 0u ? 0u : (k == 1u ? tr1 : (k == 2u ? tr2 : (uint32_t)tr[k]))); };
    auto stitch = [&](uint32_t entry) {
        nb = 0u, by_hand = 0u, a = n, exit_pos = entry;
        if (t == 0u) { // the first range starts on a true block start: its trace is the chain
            nb = n_rec;
            a = 0u;
            exit_pos = wend;
            return;
        }
        pos = entry; // hypothesis: where the true chain enters this range
        if (pos >= fast_end) return; // the chain left the fast part of the stream in front of this range
        // the trace is sorted and the walk only moves forward: ONE pointer into the trace, advanced past the entries in front of the
        // walk (rounds 2-3 searched the trace from scratch at every block: eight dependent loads from memory where this takes one or two)
        uint32_t ap = 0;
        uint32_t ta = trace_at(0u);
        for (;;) {
            // the chain is past the range: it entered behind it (a block longer than the range passes over it: no block of the chain
            // starts here) or it was walked by hand all the way through without meeting the trace - either way the chain's exit is `pos`,
            // not the end of this range's own walk, and the lane behind stitches again from there (rounds 2-5 gave up here: bit 4)
            if (pos >= hi || pos + 6u > stream_bits) { // (... or fewer bits than a block has are left: the chain has arrived at the stream's end, its padding)
                nb = by_hand;
                exit_pos = pos;
                return;
            }
            // is `pos` a block start this range's trace recorded?
            const uint32_t want = pos - lo;
            while (ta < want) {
                ap++;
                ta = trace_at(ap);
            }
            if (ta == want) {
                // from here on the trace walked the true chain (whether its blocks are well-formed is checked where they are decoded)
                nb = by_hand + (n_rec - ap);
                a = ap;
                exit_pos = wend;
                return;
            }
            if (mine && by_hand < cap) hand[(size_t)t * cap + by_hand] = (uint16_t)want; // first bit of the by-hand block (pos >= lo: the walk enters behind the range before)
            // one block by hand: the same chain walk, from `pos` with the DC category next, to its EOB.  (Rounds 2-3 decoded the block
            // symbol by symbol with the one-symbol tables and checked it - a second set of tables in LDS, and three times the steps; whether
            // a block of the true chain is well-formed is checked where it is decoded, in the fused kernel, as for the trace's blocks.)
            {
                uint32_t p = pos;
                const uint32_t limit = p + 1800u < stream_bits ? p + 1800u : stream_bits; // (a block has at most 1,728 bits)
                uint32_t md = 0; // (the DC category next)
                bool open = true;
                uint32_t bi = p >> 5;
                uint32_t ba = wp[bi], bb = wp[bi + 1u], bc = wp[bi + 2u];
                while (open && p < limit) {
                    const uint32_t bn = wp[bi + 3u];
                    asm volatile("" ::: "memory");
                    const uint32_t pk = (uint32_t)(((((unsigned long long)ba) << 32) | bb) << (p & 31u) >> 32);
                    uint32_t adv, eob;
                    chain_step(lutm, pk, md, adv, eob);
                    p += adv;
                    const uint32_t now = p >> 5;
                    const bool crossed = now != bi;
                    ba = crossed ? bb : ba;
                    bb = crossed ? bc : bb;
                    bc = crossed ? bn : bc;
                    bi = now;
                    open = eob == 0u;
                }
                if (open) { // no EOB within a block's length, or the stream ended first
                    // ... at the stream's end that is how the chain of a whole stream ends: in the padding bits behind its last block (fewer than
                    // 8 zeros, no block), and a cut stream's chain ends the same way in front of its open block - the host continues from it
                    // (whatever the ranges behind still add to the chain is caught where it is decoded: those blocks do not follow each other)
                    if (mine && pos + 2048u <= fast_end) atomicOr(&st->giveup, 16);
                    nb = by_hand;
                    exit_pos = fast_end; // (the chain ends here: the ranges behind hold none of its blocks)
                    return;
                }
                pos = p;
            }
            by_hand++;
        }
    };
    // Every lane stitches on the hypothesis that the chain enters its range where the walk of the range in front ended; a lane whose
    // neighbour reports another exit (the chain passed over that range, or went through it without meeting its trace) stitches again from
    // that exit, and so on down the wave: as many rounds as ranges in a row are passed over (a 1,728-bit block: three of 544 bits).  Lane 0
    // has no neighbour: its exit is taken to be its walk's end, the three shadows behind it put that right where it is not.
    {
        const bool can = walks && (t == 0u || lane != 0u);
        uint32_t entry = (uint32_t)__shfl_up((int)wend, 1, 64); // (all 64 lanes are here: nobody has returned)
        MS_STAMP(3);
        if (can) stitch(entry);
        MS_STAMP(4);
        bool need = false;
        for (uint32_t round = 0; round < stitch_rounds; round++) {
            const uint32_t pe = (uint32_t)__shfl_up((int)exit_pos, 1, 64);
            need = can && t != 0u && pe != entry;
            if (__ballot(need) == 0ull) break;
            if (need) {
                entry = pe;
                stitch(entry);
            }
        }
        if (need && mine) atomicOr(&st->giveup, 4); // more ranges in a row than rounds without a synchronisation point (a block covers three at most: a walk that stays out of step - periodic content)
    }
    MS_STAMP(5);
    if (!mine) nb = 0u; // a shadow's blocks are counted by the range's owner
    // ---- index of every range's first true block, and with it the first bit of every block of the true chain - HERE, in the same launch
    // (rounds 2-4: a launch of its own, scan_counts_bpos_kernel, over arrays this kernel wrote): the wave sums its 63 counts, PUBLISHES the
    // sum (one 8-byte word that carries the launch's epoch, scan_pack below) and adds up the words of the workgroups in front of it - they were
    // dispatched earlier and publish without waiting for anybody.  Then every lane writes its blocks' positions: the ones it walked by
    // hand, then its trace from the entry on.
    {
        long long inc = (long long)nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += o;
        }
        const long long tot = __shfl(inc, 63, 64);
        if (lane == 63u) __hip_atomic_store(&desc[tile], scan_pack(epoch, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long part = wave_lookback(desc, desc_half, flat_grid, epoch, tot, st, tile, ntiles);
        if (tile == ntiles - 1u && lane == 63u) *grand_total = part + tot;
        MS_STAMP(6);
        if (!mine) return;
        const unsigned long long first = (unsigned long long)(part + inc - (long long)nb);
        const uint32_t from_trace = n - a < nb ? n - a : nb, hand_n = nb - from_trace < cap ? nb - from_trace : cap; // (a <= n)
        // a range has five blocks or so: four offsets are requested before the first position is written
        auto emit = [&](const uint16_t *__restrict__ src, uint32_t cnt_, unsigned long long dst0) {
            for (uint32_t j0 = 0; j0 < cnt_; j0 += 4u) {
                uint32_t x[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) x[k] = (uint32_t)src[j0 + k < cnt_ ? j0 + k : cnt_ - 1u];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++)
                    if (j0 + k < cnt_ && dst0 + j0 + k < nblocks) bpos[dst0 + j0 + k] = lo + x[k];
            }
        };
        emit(hand + (size_t)t * cap, hand_n, first);
        emit(tr + a, from_trace, first + (nb - from_trace));
        MS_STAMP(7);
    }
#undef MS_STAMP
}

// One stream per launch: a wave's tile is its index in the grid.
template <bool kStoreEveryStep>
__global__ __launch_bounds__(256) void dec_measure_stitch_kernel(const uint32_t *__restrict__ gwords, uint32_t nwords, uint32_t last_mask, const DecLutsDev *__restrict__ L,
                                                                uint32_t fast_end, uint32_t stream_bits, uint32_t range, uint32_t nranges, uint16_t *__restrict__ starts,
                                                                uint16_t *__restrict__ hand, unsigned long long *__restrict__ desc, uint32_t desc_half, uint32_t flat_grid, uint32_t epoch,
                                                                unsigned long long nblocks, uint32_t *__restrict__ bpos, long long *__restrict__ grand_total,
                                                                uint32_t ntiles, DecStatus *__restrict__ st, uint32_t stitch_rounds, uint32_t shadows) {
    measure_stitch_body<kStoreEveryStep>(gwords, nwords, last_mask, L, fast_end, stream_bits, range, nranges, starts, hand, desc, desc_half, flat_grid, epoch, nblocks, bpos, grand_total, ntiles, st,
                        blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), stitch_rounds, shadows);
}
// A BATCH of streams per launch (tic_decompress_batch: the reference's benchmark loop decodes 49 streams of 512 x 512 one after the other,
// tests/benchmark.py:12-23 - two launches per stream are launch latency and little else).  Wave `g` of the grid works on stream
// tile_frame[g], as that stream's wave g - tile0: its words, ranges, traces, sums, positions and status are the frame's own slices of the
// batch's arrays (DecFrame), and nothing crosses a frame: the count of blocks in front of a range is taken over the frame's own waves.
template <bool kStoreEveryStep>
__global__ __launch_bounds__(256) void dec_measure_stitch_batch_kernel(const uint32_t *__restrict__ words_all, const DecFrame *__restrict__ frames, const uint32_t *__restrict__ tile_frame,
                                                                      uint32_t total_tiles, const DecLutsDev *__restrict__ L, uint32_t range, uint16_t *__restrict__ starts_all,
                                                                      uint16_t *__restrict__ hand_all, unsigned long long *__restrict__ desc, uint32_t desc_half, uint32_t flat_grid,
                                                                      uint32_t epoch, uint32_t *__restrict__ bpos_all, long long *__restrict__ totals, DecStatus *__restrict__ status,
                                                                      uint32_t stitch_rounds, uint32_t shadows) {
    const uint32_t g = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const bool beyond = g >= total_tiles; // (a wave behind the last stream's last range: it goes through the staging's barrier as that stream's wave `ntiles` and returns)
    const uint32_t f = tile_frame[beyond ? total_tiles - 1u : g];
    const DecFrame &F = frames[f];
    const size_t cap = cap_of(range);
    measure_stitch_body<kStoreEveryStep>(words_all + F.word0, F.nwords, F.last_mask, L, F.fast_end, F.stream_bits, range, F.nranges, starts_all + (size_t)F.range0 * cap,
                        hand_all + (size_t)F.range0 * cap, desc + F.tile0, desc_half, flat_grid, epoch, (unsigned long long)F.nblocks, bpos_all + F.blk0, totals + f, F.ntiles, status + f,
                        beyond ? F.ntiles : g - F.tile0, stitch_rounds, shadows);
}

// ---- decode + inverse transform, fused --------------------------------------------------------------------------------------------
// A workgroup per kDecodeWG consecutive blocks (one contiguous piece of the stream, staged in LDS: up to kBlkWin words, 250 or 500 bits
// per block on average; what lies behind is read from memory), a LANE PER BLOCK through both phases.  Phase 1: lane b decodes the block
// at bpos[b] into its LDS image - 64 int16 in NATURAL order (the zig-zag walk is undone by the store address), entry 0 = the integrated
// DC, saturated as the host decoder saturates it.  Phase 2: the same lane reads its image back, two columns at a time, dequantises (coef
// x div, one rounding; the scaled_dct branch's three), runs the exact DCT-III down the columns and along the rows on 64 registers, + 128,
// clip, truncating cast (codec.py:46-70, utils.py:40-45): idct_kernel's arithmetic with the passes' 1/4 folded into the conversion.
// LDS: pair table 9 KB + window 8.7 (17.2) KB + images 33.8 KB (132 B apart) + constants = 52 (60) KB: three (two) workgroups per CU;
// 138 VGPRs: three waves per SIMD.  (Rounds 2-4: phase 2 with 8 lanes per block and the 8x8 float64 matrix transposed through LDS, a
// workgroup barrier between the phases; what made round 3's separate decode kernel slow were its 45 two-byte global stores per lane.)
constexpr int kDecodeWG = 256;
constexpr int kPairLutDw = 2048 + kLongCodes + 4; // DecLutsDev::ac2 + long32 (the DC symbol is looked up in memory, once per lane: no dc11 here)
static_assert(offsetof(DecLutsDev, long32) == offsetof(DecLutsDev, ac2) + 8192 && offsetof(DecLutsDev, ac2) % 16 == 0 && (kPairLutDw * 4) % 16 == 0 &&
                  sizeof(DecLutsDev) >= offsetof(DecLutsDev, ac2) + kPairLutDw * 4,
              "the pair table and the long codewords are adjacent and copied in 16-byte pieces");
constexpr int kImgStrideB = 132;  // bytes between the images of two blocks: 33 dwords (a lane per image: a dword of every image in one access, no bank conflict)
__constant__ int kAnnScalesDec[64] = { // ANNSCALES of the reference's scaled_dct branch (constants.py; utils.py:59-62), as integers x 2048
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
    21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,
    8867,  12299, 11585, 10426, 8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

// kWinWords: stream words of the workgroup's window: 2048 (+ kOver) hold 256 blocks of up to 256 bits on average - three workgroups
// per CU - 4096 of up to 512 - two; what lies behind the window is read from memory
template <uint32_t kWinWords, bool kScaled>
__device__ __forceinline__ void decode_idct_body(const uint32_t *__restrict__ gwords, uint32_t nwords, uint32_t last_mask, const DecLutsDev *__restrict__ L,
                                                 const uint32_t *__restrict__ bpos, unsigned long long *__restrict__ desc, uint32_t desc_half, uint32_t flat_grid, uint32_t epoch,
                                                 const long long *__restrict__ total_blocks, unsigned long long n_want, uint32_t stream_bits,
                                                 const DecIdctArgs &a, DecStatus *__restrict__ st, const uint32_t wg /* this workgroup among the stream's `nwgs` */, const uint32_t nwgs) {
    // tables + stream window
    constexpr uint32_t kBlkWin = kWinWords + kOver;
    constexpr uint32_t kBlkLds = kBlkWin + kBlkWin / 32 + 2;
    constexpr int kLutDw = kPairLutDw;
    constexpr int kScratchDw = kLutDw + (int)kBlkLds;
    __shared__ __attribute__((aligned(16))) uint32_t scratch[kScratchDw];
    __shared__ __attribute__((aligned(16))) unsigned char img[kDecodeWG * kImgStrideB];
    __shared__ uint8_t zznat[64];
    __shared__ __attribute__((aligned(16))) double dq[64]; // the dequantisation constants, natural order (every lane reads the same entry: a broadcast)
    __shared__ long long scan_lds[16];
    uint32_t *lut = scratch;
    uint32_t *sbits = scratch + kLutDw;
#define FD_STAMP(k) DEC_STAMP(2, wg == 0u && threadIdx.x == 0, wg + 1u == nwgs && threadIdx.x == 0, k)
    FD_STAMP(0);
    const unsigned long long total = (unsigned long long)*total_blocks;
    const unsigned long long m = total < n_want ? total : n_want; // blocks produced here
    const unsigned long long b0 = (unsigned long long)wg * kDecodeWG, b = b0 + threadIdx.x;
    if (b0 >= m) return; // (the whole workgroup)
    // geometry and quality came from a.head - possibly a guess (tic_decompress_dev): pixels are written only under the header the stream
    // really has (four words, the same for every lane: scalar loads; the measure kernel echoes them to the host, which decodes again)
    if (gwords[0] != a.head[0] || gwords[1] != a.head[1] || gwords[2] != a.head[2] || gwords[3] != a.head[3]) return;
    if (threadIdx.x < 64) {
        zznat[threadIdx.x] = a.consts->zznat[threadIdx.x];
        dq[threadIdx.x] = a.consts->div[threadIdx.x];
    }
    { // the images start as zeros: the decoder writes the non-zero coefficients only
        uint4 *z = reinterpret_cast<uint4 *>(img);
        for (int k = threadIdx.x; k < kDecodeWG * kImgStrideB / 16; k += kDecodeWG) z[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    // this lane's block: position and the position of the block behind it - requested here, in front of the staging, so that they
    // arrive while the tables and the window do (behind the barrier each would be a trip to memory of its own)
    const uint32_t my_pos = b < m ? bpos[b] : 0u, next_pos = b + 1 < m ? bpos[b + 1] : 0u;
    copy16_to_lds<kDecodeWG, kPairLutDw / 4>(lut, L->ac2);
    const unsigned long long last = b0 + kDecodeWG - 1 < m - 1 ? b0 + kDecodeWG - 1 : m - 1;
    const uint32_t w0 = bpos[b0] >> 5, w1 = bpos[last] >> 5;
    const uint32_t want = w1 >= w0 ? w1 - w0 + kOver : kOver;
    const Bits words = stage_words<kDecodeWG>(sbits, gwords, w0, want < kBlkWin ? want : kBlkWin, 5u, nwords, last_mask, threadIdx.x); // (ends with a barrier: tables, window, zeros, zznat)
    // ---- phase 1: a lane per block, one SYMBOL per step (the values are needed here; the measure walk takes chains): the stream words under the read position sit
    // in registers (wa, wb) and the word behind them (wc) is fetched a step ahead, so that the table look-up is the only LDS access
    // on the lane's dependent chain; the coefficient's store (its address comes through the zig-zag table) is off that chain.
    //
    // np.cumsum of the DC differences (codec.py:53) happens HERE (rounds 2-4: a launch of its own in front of this one, a lane per block
    // as well): the lane decodes its block's DC symbol first, the workgroup sums its 256 differences and PUBLISHES the sum (scan_publish),
    // then the lanes decode their AC symbols - and only behind them every wave adds up the sums of the workgroups in front (published
    // long since: nobody spins), because the integrated DC is needed for one store at the block's end and nothing else.
    FD_STAMP(1);
    int16_t *c = reinterpret_cast<int16_t *>(img + (size_t)threadIdx.x * kImgStrideB);
    uint32_t pos = my_pos;
    uint32_t wi = pos >> 5;
    uint32_t wa = 0, wb = 0, wc = 0;
    auto advance = [&](uint32_t bits, uint32_t wn) { // a symbol consumes at most 27 bits: at most one word boundary is crossed
        pos += bits;
        const bool crossed = (pos >> 5) != wi;
        wa = crossed ? wb : wa;
        wb = crossed ? wc : wb;
        wc = crossed ? wn : wc;
        wi += crossed ? 1u : 0u;
    };
    long long dc_diff = 0;
    if (b < m) {
        wa = word_be(words, wi), wb = word_be(words, wi + 1u), wc = word_be(words, wi + 2u);
        const uint32_t wn = word_be(words, wi + 3u);
        const uint32_t pk = (uint32_t)(((((unsigned long long)wa) << 32) | wb) << (pos & 31u) >> 32);
        const uint32_t e = L->dc11[pk >> 21]; // (one look-up per lane: from memory, no room in LDS for this table)
        if (!e) atomicOr(&st->giveup, 32);    // (measure or stitch walked this block: a DC codeword is there)
        dc_diff = (long long)value_of(pk, (int)(e >> 8), (int)(e & 15u));
        advance((e >> 8) + (e & 15u), wn);
    }
    FD_STAMP(2);
    long long dc_tile;
    const long long dc_inc = wg_inclusive_scan(dc_diff, scan_lds, dc_tile);
    if (threadIdx.x == 0) __hip_atomic_store(&desc[wg], scan_pack(epoch, dc_tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (scan_publish, with the stream's own index)
    if (b < m) {
        bool ok = true;
        {
            int k = 1;
            bool live = true, in_long = false;
            // One table look-up per step, as in the measure walk (a long codeword takes a second step, not a second look-up) - and the look-up
            // settles TWO symbols when the second one's codeword lies inside the same 11 bits (dec_pair_luts_fill): 25 steps per block
            // instead of 46 on noise at q = 50, and the step is no longer - the lane's dependent chain is look-up -> bits consumed -> next
            // window; the two values and their stores hang off it.
            while (live) {
                const uint32_t wn = word_be(words, wi + 3u);
                asm volatile("" ::: "memory"); // (the request stays in front of the table look-up)
                const uint32_t pk = (uint32_t)(((((unsigned long long)wa) << 32) | wb) << (pos & 31u) >> 32);
                const uint32_t li = (pk >> 16) - (uint32_t)kLongFirst;
                const uint32_t e = lut[in_long ? 2048u + (li < (uint32_t)kLongCodes ? li : (uint32_t)kLongCodes) : pk >> 21];
                const bool none = e == 0u;
                const bool esc = none && !in_long;  // the prefix of a long codeword: the next step resolves it
                const bool nocode = none && in_long; // no codeword at all
                const bool eob1 = !none && (e & 0xffu) == 0u;
                const int len1 = (int)((e >> 8) & 31u), size1 = (int)(e & 15u);
                const int k1 = k + (int)((e >> 4) & 15u);
                const bool bad1 = nocode || (!none && !eob1 && k1 > 63);
                if (!none && !eob1 && !bad1) c[zznat[k1]] = (int16_t)value_of(pk, len1, size1);
                const bool has2 = ((e >> 13) & 1u) != 0u; // (never behind an EOB, never in a long codeword's step)
                const uint32_t e2 = e >> 14;
                const bool eob2 = has2 && (e2 & 0xffu) == 0u;
                const int len2 = (int)((e2 >> 8) & 31u), size2 = (int)(e2 & 15u);
                const int k2 = k1 + 1 + (int)((e2 >> 4) & 15u);
                const bool bad2 = has2 && !bad1 && !eob2 && k2 > 63;
                if (has2 && !eob2 && !bad1 && !bad2) c[zznat[k2]] = (int16_t)value_of(pk << (len1 + size1), len2, size2); // (at most 11 + 10 bits in front of and in it)
                advance(none ? 0u : e >> 27, wn); // (at most 26 bits)
                k = none ? k : (has2 ? k2 + 1 : k1 + 1);
                in_long = esc;
                ok = ok && !bad1 && !bad2;
                live = !eob1 && !eob2 && !bad1 && !bad2;
            }
        }
        // the block is well-formed, and the next block of the chain starts where this one ends (the measure kernel vouches for
        // neither: its walk goes on through incidents)
        if (!ok || (b + 1 < m && next_pos != pos)) atomicOr(&st->giveup, 32);
        // ... and it ends inside the stream: the words behind the stream's end read as zeros, and a codeword's last bits may have been
        // such zeros (only with margin_bits = 0: a block that starts 2,048 bits in front of the end cannot reach it)
        if (pos > stream_bits) atomicOr(&st->giveup, 256);
        if (b == m - 1) {
            st->pos_out = pos;
            st->m = m;
        }
    }
    FD_STAMP(3);
    { // the sums of the workgroups in front, wave by wave (no barrier: a wave whose blocks were short goes on)
        const long long part = wave_lookback(desc, desc_half, flat_grid, epoch, dc_tile, st, wg, nwgs); // (every wave of the workgroup publishes the same inclusive sum: whichever is first)
        if (b < m) {
            const long long dc = part + dc_inc; // sum of the differences of blocks 0..b
            const int32_t dc32 = (int32_t)dc;   // (the host decoder's long long, narrowed where it is used)
            c[0] = (int16_t)(dc32 < -32768 ? -32768 : (dc32 > 32767 ? 32767 : dc32)); // the host decoder's sat16
            if (b == m - 1) st->dc_out = (int)dc32; // running DC behind the last block produced here (the host's tail continues from it)
        }
    }
    // ---- phase 2: the lane that decoded a block transforms it - all 64 coefficients in registers, no transposition, no barrier between
    // the phases (a wave whose blocks were short goes on while the others still decode), nothing but the lane's own image read back.
    // (Rounds 2-4: 8 lanes per block, 8 blocks per wave and round, the 8x8 float64 matrix transposed through LDS between the passes:
    // the same arithmetic, plus two LDS round trips and a workgroup barrier per round; profiles/r05_decoder.txt.)
    FD_STAMP(4);
    if (b >= m) return;
    asm volatile("" ::: "memory"); // (the image's two-byte stores above are read back as 16-byte pieces)
    double x[64]; // x[u * 8 + v]: natural order
    // axis -2, two columns at a time: their sixteen coefficients come back as eight dwords (the lanes' images are an odd number of dwords
    // apart: no bank conflict), the constants as eight 16-byte broadcasts; nothing of the columns behind is in registers yet (the whole
    // block at once, packed, plus its constants would be 160 registers on top of the 128 of x: the third wave per SIMD)
    const uint32_t *im32 = reinterpret_cast<const uint32_t *>(img + (size_t)threadIdx.x * kImgStrideB);
    const double2 *dq2 = reinterpret_cast<const double2 *>(dq);
#pragma unroll
    for (int vp = 0; vp < 4; vp++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t w2 = im32[u * 4 + vp];
            const double2 d2 = dq2[u * 4 + vp];
            const int lo = (int)(int16_t)(w2 & 0xffffu), hi = (int)w2 >> 16;
            if (kScaled) { // codec.py:60-62: (coeffs / ANNSCALES) * 2**quality, then the inverse quantiser of quality 50: three roundings
                x[u * 8 + 2 * vp] = (((double)lo / ((double)kAnnScalesDec[u * 8 + 2 * vp] / 2048.0)) * a.pow2) * d2.x;
                x[u * 8 + 2 * vp + 1] = (((double)hi / ((double)kAnnScalesDec[u * 8 + 2 * vp + 1] / 2048.0)) * a.pow2) * d2.y;
            } else { // coeffs * (Q*factor/100)
                x[u * 8 + 2 * vp] = (double)lo * d2.x;
                x[u * 8 + 2 * vp + 1] = (double)hi * d2.y;
            }
        }
#pragma unroll
        for (int v = 2 * vp; v < 2 * vp + 2; v++) {
            idct8_exact_impl<false>(x[v], x[8 + v], x[16 + v], x[24 + v], x[32 + v], x[40 + v], x[48 + v], x[56 + v]); // (four times the column's values)
            // the column's results exist HERE: element (u, v) feeds pixel row u only, whose store sits behind a condition, and the compiler
            // otherwise sinks the tail of every column's arithmetic into those conditions - twice as many values in flight, 250 registers
            asm volatile("" : "+v"(x[v]), "+v"(x[8 + v]), "+v"(x[16 + v]), "+v"(x[24 + v]), "+v"(x[32 + v]), "+v"(x[40 + v]), "+v"(x[48 + v]), "+v"(x[56 + v]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    FD_STAMP(5);
    const uint32_t blk = (uint32_t)b; // (bit positions are 32-bit: fewer than 2^32 / 6 blocks)
    const uint32_t by = blk / (uint32_t)a.bw, bx = blk - by * (uint32_t)a.bw;
    const int x0 = (int)bx * 8;
    const bool whole = a.aligned8 && x0 + 8 <= a.w;
    uint8_t *p = a.out + (long)by * 8 * a.stride + x0;
    const int rows_here = a.h - (int)by * 8; // (>= 1: the block exists)
#pragma unroll
    for (int u = 0; u < 8; u++) { // axis -1, a pixel row at a time
        idct8_exact_impl<false>(x[u * 8], x[u * 8 + 1], x[u * 8 + 2], x[u * 8 + 3], x[u * 8 + 4], x[u * 8 + 5], x[u * 8 + 6], x[u * 8 + 7]); // (sixteen times the row's)
        uint32_t px[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // + 128, clip, truncation toward zero as astype(np.uint8) on a clipped value - on sixteen times the value (idct8_exact_impl: the
            // passes above leave out their 1/4): fl(16 r + 2048) = 16 fl(r + 128); truncation first and the clip on the integer (the
            // bounds are integers, truncation is monotonic, the sum is far inside the int range: one v_med3_i32 instead of a float64
            // maximum and minimum), and floor(y / 16) of a non-negative integer y is y >> 4.
            // (the scaled_dct branch multiplies by 2**quality, quality up to 62: there the sum is clamped as a double first - a conversion that
            // overflows the int is undefined; the clamp's bounds are integers, so the result is the same)
            double s16 = x[u * 8 + k] + 2048.0;
            if (kScaled) s16 = fmin(fmax(s16, 0.0), 4080.0);
            const int y = (int)s16;
            px[k] = (uint32_t)(y < 0 ? 0 : (y > 4080 ? 4080 : y)) >> 4;
        }
        uint2 o;
        o.x = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        o.y = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
        asm volatile("" : "+v"(o.x), "+v"(o.y)); // (as above: the row's arithmetic stays in front of the condition)
        if (u < rows_here) {
            if (whole) {
                *reinterpret_cast<uint2 *>(p) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (x0 + k < a.w) p[k] = (uint8_t)((k < 4 ? o.x : o.y) >> (8 * (k & 3)));
            }
        }
        p += a.stride;
        __builtin_amdgcn_sched_barrier(0);
    }
    FD_STAMP(6);
#undef FD_STAMP
}

template <uint32_t kWinWords, bool kScaled>
__global__ __launch_bounds__(kDecodeWG, kWinWords <= 2048u ? 3 : 2) void dec_decode_idct_kernel(const uint32_t *__restrict__ gwords, uint32_t nwords, uint32_t last_mask, const DecLutsDev *__restrict__ L,
                                                                    const uint32_t *__restrict__ bpos, unsigned long long *__restrict__ desc, uint32_t desc_half, uint32_t flat_grid, uint32_t epoch,
                                                                    const long long *__restrict__ total_blocks, unsigned long long n_want, uint32_t stream_bits,
                                                                    DecIdctArgs a, DecStatus *__restrict__ st) {
    decode_idct_body<kWinWords, kScaled>(gwords, nwords, last_mask, L, bpos, desc, desc_half, flat_grid, epoch, total_blocks, n_want, stream_bits, a, st, blockIdx.x, gridDim.x);
}
// The batch form: workgroup g decodes 256 blocks of stream wg_frame[g] (a stream's blocks start a workgroup of their own: the DC sum
// starts over with every frame, codec.py:53) and writes that frame's pixels through the frame's own DecIdctArgs.
template <uint32_t kWinWords>
__global__ __launch_bounds__(kDecodeWG, kWinWords <= 2048u ? 3 : 2) void dec_decode_idct_batch_kernel(const uint32_t *__restrict__ words_all, const DecFrame *__restrict__ frames, const uint32_t *__restrict__ wg_frame,
                                                                          const DecLutsDev *__restrict__ L, const uint32_t *__restrict__ bpos_all, unsigned long long *__restrict__ desc, uint32_t desc_half,
                                                                          uint32_t flat_grid, uint32_t epoch, const long long *__restrict__ totals, DecStatus *__restrict__ status) {
    const uint32_t f = wg_frame[blockIdx.x];
    const DecFrame &F = frames[f];
    decode_idct_body<kWinWords, false>(words_all + F.word0, F.nwords, F.last_mask, L, bpos_all + F.blk0, desc + F.wg0, desc_half, flat_grid, epoch, totals + f, (unsigned long long)F.nblocks,
                                       F.stream_bits, F.idct, status + f, blockIdx.x - F.wg0, F.nwgs);
}

} // namespace

// rounds of the stitch's hand-over loop (a hook for measurements: TIC_DECODE_ROUNDS)
static uint32_t stitch_rounds() {
    if (const char *e = test_hook("TIC_DECODE_ROUNDS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return (uint32_t)v;
    }
    return 8u;
}

// ranges in front of its own that a wave of the measure kernel shadows: the true entry into the wave's first own range has to arrive through them - a
// block of the chain is up to 1,728 bits long and a walk needs a few blocks to fall in step (hook for measurements: TIC_DECODE_SHADOWS)
static uint32_t dec_shadows(int range_bits) {
    if (const char *e = test_hook("TIC_DECODE_SHADOWS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= (int)kShadowMax) return (uint32_t)v;
    }
    return range_bits < 544 ? 8u : 4u;
}

bool entropy_decode_gpu_range_ok(int range_bits) {
    return range_bits >= kRangeMin && range_bits <= kRangeMax && range_bits % 64 == 32; // an odd number of 32-bit words
}

size_t entropy_decode_gpu_work_bytes(size_t stream_bytes, size_t nblocks) {
    const size_t nbits = stream_bytes * 8;
    const size_t nranges = nbits / kRangeMin + 2; // (the smallest range: most ranges, and the most room per stream bit)
    return nranges * ((size_t)cap_of(kRangeMin) * 2 * 2) + nblocks * 4 + 16384; // (two traces per range, a 4-byte position per block; every piece is rounded up to 256 B)
}

size_t entropy_decode_gpu_desc_words(size_t stream_bytes, size_t nblocks) { // look-back words of the two sums (own and inclusive sums: a quarter of the array each)
    const size_t nranges = stream_bytes * 8 / kRangeMin + 2;
    const size_t tr = nranges / (64 - kShadowMax) + 2, tb = nblocks / kDecodeWG + 2; // (workgroups of the measure kernel and of the fused kernel)
    return 4 * (tr > tb ? tr : tb);
}

hipError_t entropy_decode_idct_gpu(const void *d_stream_words, size_t stream_bytes, size_t nblocks, const DecLutsDev *d_luts, void *d_work,
                                   size_t work_bytes, unsigned long long *d_desc, size_t desc_words, uint32_t epoch, const DecIdctArgs &idct,
                                   DecStatus *d_status, int range_bits, int margin_bits, hipStream_t stream, int flat_grid) {
    const size_t nbits = stream_bytes * 8;
    if (!entropy_decode_gpu_range_ok(range_bits)) return hipErrorInvalidValue;
    const uint32_t range = (uint32_t)range_bits;
    const uint32_t kCap = cap_of(range);
    const int kRange = range_bits;
    if (nbits < 128 + 2048 + 2048 || nbits + 8192 >= (1ull << 32) || nblocks == 0) return hipErrorInvalidValue; // (a walk stands up to 1,827 bits behind the end)
    if (margin_bits != 0 && margin_bits != 2048) return hipErrorInvalidValue;
    if (flat_grid < 0) return hipErrorInvalidValue;
    if (work_bytes < entropy_decode_gpu_work_bytes(stream_bytes, nblocks)) return hipErrorInvalidValue;
    // a block may START up to here.  margin_bits = 2048 (the host decoder's rule, rounds 2-3): every block of the chain lies inside the
    // stream whatever it holds, and the blocks that start behind are the caller's (bit-serial, on the host).  margin_bits = 0: the chain
    // runs to the stream's end - nothing is left for the host when the stream is whole - and a block that reaches behind the end, where the
    // words read as zeros, raises giveup bit 256 (a cut stream: the caller comes back with the margin).
    const uint32_t fast_end = (uint32_t)(nbits - (size_t)margin_bits);
    const uint32_t nranges = (uint32_t)((fast_end - 128 + kRange - 1) / kRange);
    const uint32_t shadows = dec_shadows(range_bits), owned = 64u - shadows;
    const unsigned measure_wgs = (nranges + owned - 1u) / owned;
    const size_t ntiles_r = measure_wgs, ntiles_b = (nblocks + kDecodeWG - 1) / kDecodeWG; // (the workgroups of the two kernels are the tiles of the two sums)
    // The look-back words of the two sums live in an array of their own that holds nothing else, ever: a word there either is zero
    // (since allocation) or carries the epoch of the launch that wrote it, and the caller never reuses an epoch on it.  (Inside the
    // workspace the pieces move with the sizes of the call: a stale trace entry could pass for a published sum.)
    if (!d_desc || desc_words < 4 * (ntiles_r > ntiles_b ? ntiles_r : ntiles_b) || epoch == 0 || epoch >= (1u << 22)) return hipErrorInvalidValue;
    const uint32_t desc_half = (uint32_t)(desc_words / 4); // per sum: own sums, then inclusive sums
    unsigned long long *desc_r = d_desc, *desc_b = d_desc + 2 * (size_t)desc_half;
    // workspace carve-up
    char *w = (char *)d_work;
    auto take = [&](size_t bytes) { char *p = w; w += (bytes + 255) / 256 * 256; return (void *)p; };
    long long *totals = (long long *)take(16);
    uint16_t *starts = (uint16_t *)take((size_t)nranges * kCap * 2), *hand = (uint16_t *)take((size_t)nranges * kCap * 2);
    uint32_t *bpos = (uint32_t *)take(nblocks * 4);
    if ((size_t)(w - (char *)d_work) > work_bytes) return hipErrorInvalidValue;
    const uint32_t *words = (const uint32_t *)d_stream_words;
    const uint32_t nwords = (uint32_t)((stream_bytes + 3) / 4);
    const uint32_t last_mask = (stream_bytes & 3) ? 0xffffffffu << (8u * (4u - (uint32_t)(stream_bytes & 3))) : 0xffffffffu; // (big-endian: the stream's bytes are the word's high bytes)
    // (*d_status is zeroed by the caller: it is host-mapped memory)
    // waves per workgroup of the measure kernel: four while their windows fit 64 KB of LDS together with the tables, else two, else one
    const unsigned win_lds = stage_lds_words(range) * 4u;
    const unsigned wpw = (unsigned)kChainLds + 4u * win_lds <= 60000u ? 4u : ((unsigned)kChainLds + 2u * win_lds <= 60000u ? 2u : 1u);
    auto measure = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((measure_wgs + wpw - 1u) / wpw), dim3(64u * wpw), wpw * win_lds, stream, words, nwords, last_mask, d_luts, fast_end, (uint32_t)nbits, range,
                           nranges, starts, hand, desc_r, desc_half, (uint32_t)flat_grid, 2u * epoch, (unsigned long long)nblocks, bpos, totals, (uint32_t)measure_wgs, d_status,
                           stitch_rounds(), shadows);
    };
    measure_wgs <= kEveryStepTiles ? measure(dec_measure_stitch_kernel<true>) : measure(dec_measure_stitch_kernel<false>);
    const dim3 dgrid((unsigned)((nblocks + kDecodeWG - 1) / kDecodeWG));
    auto fused = [&](auto kern) {
        hipLaunchKernelGGL(kern, dgrid, dim3(kDecodeWG), 0, stream, words, nwords, last_mask, d_luts, (const uint32_t *)bpos, desc_b, desc_half, (uint32_t)flat_grid, 2u * epoch + 1u,
                           (const long long *)totals, (unsigned long long)nblocks, (uint32_t)nbits, idct, d_status);
    };
    const bool small_win = nbits / nblocks <= 240; // sparse enough for the small window: one workgroup more per CU
    if (idct.scaled) small_win ? fused(dec_decode_idct_kernel<2048, true>) : fused(dec_decode_idct_kernel<4096, true>);
    else small_win ? fused(dec_decode_idct_kernel<2048, false>) : fused(dec_decode_idct_kernel<4096, false>);
    return hipGetLastError();
}


size_t entropy_decode_batch_work_bytes(size_t total_ranges_288, size_t total_blocks, size_t nframes) {
    return total_ranges_288 * ((size_t)cap_of(kRangeMin) * 2 * 2) + total_blocks * 4 + nframes * 8 + 16384;
}
uint32_t entropy_decode_batch_tiles(uint32_t nranges, int range_bits) { const uint32_t owned = 64u - dec_shadows(range_bits); return (nranges + owned - 1u) / owned; }
uint32_t entropy_decode_batch_wgs(size_t nblocks) { return (uint32_t)((nblocks + kDecodeWG - 1) / kDecodeWG); }
uint32_t entropy_decode_batch_ranges(size_t stream_bytes, int range_bits) { return (uint32_t)((stream_bytes * 8 - 128 + (size_t)range_bits - 1) / (size_t)range_bits); }

hipError_t entropy_decode_idct_gpu_batch(const void *d_words_all, const DecFrame *d_frames, const uint32_t *d_tile_frame, const uint32_t *d_wg_frame, uint32_t nframes,
                                         uint32_t total_tiles, uint32_t total_wgs, uint32_t total_ranges, size_t total_blocks, bool small_win, const DecLutsDev *d_luts, void *d_work,
                                         size_t work_bytes, unsigned long long *d_desc, size_t desc_words, uint32_t epoch, DecStatus *d_status, int range_bits, hipStream_t stream,
                                         int flat_grid) {
    if (!entropy_decode_gpu_range_ok(range_bits) || nframes == 0 || total_tiles == 0 || total_wgs == 0) return hipErrorInvalidValue;
    const uint32_t range = (uint32_t)range_bits, kCap = cap_of(range);
    if (!d_desc || desc_words < 4 * (size_t)(total_tiles > total_wgs ? total_tiles : total_wgs) || epoch == 0 || epoch >= (1u << 22) || flat_grid < 0) return hipErrorInvalidValue;
    const uint32_t desc_half = (uint32_t)(desc_words / 4);
    unsigned long long *desc_r = d_desc, *desc_b = d_desc + 2 * (size_t)desc_half;
    char *w = (char *)d_work;
    auto take = [&](size_t bytes) { char *p = w; w += (bytes + 255) / 256 * 256; return (void *)p; };
    long long *totals = (long long *)take((size_t)nframes * 8);
    uint16_t *starts = (uint16_t *)take((size_t)total_ranges * kCap * 2), *hand = (uint16_t *)take((size_t)total_ranges * kCap * 2);
    uint32_t *bpos = (uint32_t *)take(total_blocks * 4);
    if ((size_t)(w - (char *)d_work) > work_bytes) return hipErrorInvalidValue;
    const unsigned win_lds = stage_lds_words(range) * 4u;
    const unsigned wpw = (unsigned)kChainLds + 4u * win_lds <= 60000u ? 4u : ((unsigned)kChainLds + 2u * win_lds <= 60000u ? 2u : 1u);
    auto measure = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((total_tiles + wpw - 1u) / wpw), dim3(64u * wpw), wpw * win_lds, stream, (const uint32_t *)d_words_all, d_frames, d_tile_frame, total_tiles, d_luts,
                           range, starts, hand, desc_r, desc_half, (uint32_t)flat_grid, 2u * epoch, bpos, totals, d_status, stitch_rounds(), dec_shadows(range_bits));
    };
    total_tiles <= kEveryStepTiles ? measure(dec_measure_stitch_batch_kernel<true>) : measure(dec_measure_stitch_batch_kernel<false>);
    if (small_win)
        hipLaunchKernelGGL(dec_decode_idct_batch_kernel<2048>, dim3(total_wgs), dim3(kDecodeWG), 0, stream, (const uint32_t *)d_words_all, d_frames, d_wg_frame, d_luts, (const uint32_t *)bpos, desc_b,
                           desc_half, (uint32_t)flat_grid, 2u * epoch + 1u, (const long long *)totals, d_status);
    else
        hipLaunchKernelGGL(dec_decode_idct_batch_kernel<4096>, dim3(total_wgs), dim3(kDecodeWG), 0, stream, (const uint32_t *)d_words_all, d_frames, d_wg_frame, d_luts, (const uint32_t *)bpos, desc_b,
                           desc_half, (uint32_t)flat_grid, 2u * epoch + 1u, (const long long *)totals, d_status);
    return hipGetLastError();
}

} // namespace tic

#if defined(TIC_EXP) && TIC_EXP == 700
extern "C" int tic_debug_dec_stamps(unsigned long long *out /* [4][16]: measure first / last wave, fused first / last workgroup */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tic::g_dec_stamps), sizeof(tic::g_dec_stamps));
}
#endif
