// tic_entropy_gpu.hip - entropy stage on the GPU: run-length + Huffman symbol generation and
// parallel bit packing, producing the reference's stream byte for byte.
//
// Replaces, on the device, the per-block Python loops of compress() (codec.py:142-162 of the reference):
// DC DPCM (codec.py:34-35), encode_run_length (huffman.py:12-33), encode_huffman (huffman.py:41-63) and the
// MSB-first BitBuffer (bitbuffer.py).  The host coder in tic_entropy.cpp remains (it needs no GPU and is what the
// batch pipeline's worker threads run); this path keeps a whole frame on the device: only the finished stream
// (about 1/7 of the coefficient bytes on noise, far less on natural images) crosses PCIe.
//
// Decomposition: 8 lanes per block, lane k owns zig-zag entries 8k..8k+7 (one 16-byte load).  The only cross-lane
// dependency of the symbol stream is the run of zeros carried into a lane, an associative "carry-through" scan over
// the 8 lanes.  A wave owns a PARTITION of 8 consecutive blocks of one frame, i.e. one contiguous bit range of the stream.
//
// Two launches (three for frames beyond 8192^2), one walk over the symbols, no zero fill, no atomics on the stream:
//   entropy_pack_kernel     a wave walks the symbols of its 8 blocks ONCE, every lane packing its bits into a private string
//                           in LDS (at most 245 bits) and counting them; a prefix over the lanes places the strings in the
//                           wave's LDS image of its bit range, which leaves as coalesced words into the wave's slot of a
//                           staging buffer - every partition starts at bit 0 of its own slot, nothing is shared.  The
//                           bit count goes to nbits[partition], the workgroup's 16 counts summed to gsum[group].
//   entropy_tilesum_kernel  only for frames of more than 8192 groups: sums of 256 group sums, a second level of offsets.
//   entropy_place_kernel    a workgroup per 32 partitions: its stream offset is the sum of the group sums before it (one
//                           batch of independent loads), its partitions' offsets a prefix over 35 bit counts; every
//                           OUTPUT word is then written once, complete - words inside one partition by a funnel shift
//                           of two staged words, the words in which partitions meet piece by piece.  It also writes
//                           header, length and the caller's status.
// History (profiles/r02_entropy_*.txt).  Round 1 ran five launches per frame - count (walks the symbols), a rocPRIM scan
// (two kernels), a zero fill, emit (walks the symbols again and ORs into the zeroed stream; edge words by global atomics):
// 109-129 us for a 4096^2 frame against a 10 us transform.  Explored and dropped this round: a single pass with a decoupled
// look-back over the waves (byte-exact, 127 us: 5,120 resident 8-block partitions start together and each sums up to 5,000
// predecessor descriptors, 64 per memory round trip); group sums by device-scope atomics from the packing waves (pack
// 68 us instead of 29: 64 same-address atomics from 8 XCDs per group); a one-workgroup scan kernel between pack and place
// (59 us of serial load latency); a branch-free symbol walk (+2 us: for noise every wave walks every position anyway, and
// the uniform bit sink costs more than the branch it saves).  What helped after the structure stood (pack 29 -> 25 us, place
// 9.5 -> 8.5): every cross-lane scan and reduction by DPP instead of LDS shuffles, write-through stores, the coefficient
// load and the zero-run scan in front of the barrier that publishes the tables.  DESIGN.md section 5.4 has the numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tic_entropy_gpu.h"
#include "tic_hooks.h"
#include "tic_tables.h"

namespace tic {

void build_huff_dev(HuffDev *t) {
    memset(t, 0, sizeof(*t));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < kDcBits[l - 1]; i++) {
            const int sz = kDcVals[k++]; // the symbol is the size category
            t->dc_sym[sz] = code++ << sz;
            t->dc_bits[sz] = (unsigned)(l + sz);
        }
        code <<= 1;
    }
    code = 0;
    k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < kAcBits[l - 1]; i++) {
            const int rs = kAcVals[k++], sz = rs & 15; // (run << 4) | size
            t->ac_sym[rs] = code++ << sz;
            t->ac_bits[rs] = (unsigned)(l + sz);
        }
        code <<= 1;
    }
    for (int i = 0; i < 256; i++) t->ac_pack[i] = t->ac_bits[i] ? ((t->ac_bits[i] << 27) | t->ac_sym[i]) : 0u; // <= 26 bits each
    for (int i = 0; i < 16; i++) t->dc_pack[i] = (t->dc_bits[i] && i <= 11) ? ((t->dc_bits[i] << 27) | t->dc_sym[i]) : 0u;
}

namespace {

// Size category of a coefficient: its bit length, 0 for 0.  The exponent of the float is exact for |v| <= 32768.
__device__ __forceinline__ int size_category(int v) {
    int e;
    (void)frexpf((float)v, &e); // v_cvt_f32_i32 + v_frexp_exp_i32_f32; frexp(0) gives exponent 0
    return e;
}

// Value bits of a coefficient of size category sz: v for v > 0, v - 1 for v < 0, low sz bits (huffman.py:51-56).
__device__ __forceinline__ uint32_t value_bits(int v, int sz) { return __builtin_amdgcn_ubfe((uint32_t)(v + (v >> 31)), 0u, (uint32_t)sz); }

// The symbols of one lane (8 consecutive scan positions of one block) into its sink; returns their total bits.
// huffman.py:12-33 (run/size symbols, ZRL, EOB) and :41-63 (codeword + value bits).
template <bool EMIT, typename Sink>
__device__ __forceinline__ int walk_lane(const int16_t c[8], int k, int carry_run, int dc_diff, const uint2 *ac_tab,
                                         const uint2 *dc_tab, Sink *sink, int *err) {
    int bits = 0;
    int run = carry_run;
    if (k == 0) { // DC: category code + value bits (huffman.py:41-63 with dc_ac = DC)
        const int v = dc_diff;
        const int sz = size_category(v);
        const uint2 e = dc_tab[sz & 15]; // |difference| <= 65535: sz <= 16, and 16 wraps to the (valid) entry 0 ...
        if (e.y == 0u || sz > 11) {      // ... hence the explicit bound
            *err = 1;
        } else {
            bits += (int)e.y;
            if (EMIT) sink->put(e.x | value_bits(v, sz), e.y);
        }
    }
    const uint2 zrl = ac_tab[0xF0];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (k == 0 && j == 0) continue;
        const int v = c[j];
        if (v == 0) {
            run++;
            continue;
        }
        while (run >= 16) { // ZRL = (15,0), huffman.py:26-28
            bits += (int)zrl.y;
            if (EMIT) sink->put(zrl.x, zrl.y);
            run -= 16;
        }
        const int sz = size_category(v); // 1..16 for an int16
        const uint2 e = ac_tab[(run << 4) | (sz > 15 ? 15 : sz)];
        if (e.y == 0u) { // sizes above 10 have no code (the reference raises KeyError)
            *err = 1;
        } else {
            bits += (int)e.y;
            if (EMIT) sink->put(e.x | value_bits(v, sz), e.y);
        }
        run = 0;
    }
    if (k == 7) { // EOB = (0,0) always closes the block (huffman.py:33)
        const uint2 e = ac_tab[0];
        bits += (int)e.y;
        if (EMIT) sink->put(e.x, e.y);
    }
    return bits;
}

// The walk of the packing kernel.  Against walk_lane above: the DC position of lane 0 is an ordinary zero (its run starts at
// -1); the zero-run escapes are at most two conditional puts (ZRL ZRL packed into one 22-bit symbol), and compiled in only
// where the WAVE has a lane that needs one (ZRL = false otherwise: the kernel decides per wave, see need_zrl there); a
// coefficient without a code becomes a symbol of 0 bits and an error flag, without a branch.  (The two tests inside the loop
// cost 1.6 us of 25 on a 4096^2 noise frame, for paths that noise never takes.)
template <bool ZRL, typename Sink>
__device__ __forceinline__ void walk_lane_lean(const int16_t c[8], int k, int carry_run, int dc_diff, const uint2 *ac_tab,
                                               const uint2 *dc_tab, Sink *sink, int *err) {
    int run = carry_run;
    const uint2 zrl = ac_tab[0xF0];
    const uint32_t zrl2 = (zrl.x << zrl.y) | zrl.x, zrl2_bits = 2u * zrl.y; // 22 bits
    if (k == 0) {
        const int v = dc_diff;
        const int sz = size_category(v);
        const uint2 e = dc_tab[sz & 15];
        const bool bad = e.y == 0u || sz > 11; // (|difference| up to 65535: size 16 wraps to the valid entry 0, hence the bound)
        *err |= bad ? 1 : 0;
        sink->put(bad ? 0u : (e.x | value_bits(v, sz)), bad ? 0u : e.y);
        run = -1;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int v = (j == 0 && k == 0) ? 0 : (int)c[j];
        if (v == 0) {
            run++;
            continue;
        }
        if (ZRL && run >= 16) { // at most three ZRL = (15,0), huffman.py:26-28
            if (run >= 32) {
                sink->put(zrl2, zrl2_bits);
                run -= 32;
            }
            if (run >= 16) {
                sink->put(zrl.x, zrl.y);
                run -= 16;
            }
        }
        const int sz = size_category(v); // 1..16 for an int16
        const uint2 e = ac_tab[((run & 15) << 4) | (sz > 15 ? 15 : sz)];
        const bool bad = e.y == 0u; // sizes above 10 have no code (the reference raises KeyError)
        *err |= bad ? 1 : 0;
        sink->put(bad ? 0u : (e.x | value_bits(v, sz)), e.y);
        run = 0;
    }
    if (k == 7) { // EOB = (0,0) always closes the block (huffman.py:33)
        const uint2 e = ac_tab[0];
        sink->put(e.x, e.y);
    }
}

constexpr int kWaveImageWords = 432; // 8 blocks x at most 64 x 27 bits, plus word alignment: 13,855 bits
constexpr int kLaneWords = 8;                    // a lane emits at most 3 ZRL + 8 x 26 + EOB = 245 bits
constexpr int kStageWords = kWaveImageWords + 2; // staging slot of a partition (8 blocks), 32-bit words
constexpr int kGroup = 16;                       // partitions (waves) per workgroup of the packing kernel = per group sum

// Bit sink into the lane's private string (bit 0 = MSB of word 0), branch-free: every symbol rewrites the word under
// construction in LDS (word i of the lane's string is str[i * 64], so the 64 lanes of a wave hit 64 different banks), a
// symbol of 0 bits is a no-op.  The last, partial word stays in `cur`.
struct LaneSink {
    uint32_t *str;
    uint32_t cur;  // word under construction (MSB = first bit)
    uint32_t sh;   // bits used in it, 0..31
    uint32_t full; // complete words behind it
    __device__ __forceinline__ void put(uint32_t v, uint32_t n) { // n <= 27, v < 2^n
        const unsigned long long w64 = ((((unsigned long long)v) << 32) << (32u - n)) >> sh; // the symbol at bit `sh` of 64
        const uint32_t hi = cur | (uint32_t)(w64 >> 32), lo = (uint32_t)w64;
        const uint32_t nsh = sh + n;
        const bool done = nsh >= 32u;
        str[full * 64u] = hi;
        cur = done ? lo : hi;
        sh = nsh & 31u;
        full += done ? 1u : 0u;
    }
    __device__ __forceinline__ uint32_t bits() const { return full * 32u + sh; }
};

// The same sink with a branch: the word under construction reaches LDS only when it is complete.
struct LaneSinkB {
    uint32_t *str;
    uint32_t cur, sh, full;
    __device__ __forceinline__ void put(uint32_t v, uint32_t n) { // n <= 27, v < 2^n (n = 0, v = 0: nothing happens)
        const uint32_t avail = 32u - sh;
        if (n < avail) {
            cur |= v << ((avail - n) & 31u);
            sh += n;
        } else {
            const uint32_t rest = n - avail;
            cur |= v >> rest;
            str[full * 64u] = cur;
            full++;
            sh = rest;
            cur = rest ? (v << (32u - rest)) : 0u;
        }
    }
    __device__ __forceinline__ uint32_t bits() const { return full * 32u + sh; }
};

// The same walk in uniform control flow (a zero coefficient is a symbol of 0 bits): timing-only builds, see the history above.
__device__ __forceinline__ void walk_pack(const int16_t c[8], int k, int carry_run, int dc_diff, const uint2 *ac_tab,
                                          const uint2 *dc_tab, LaneSink &sink, int &err) {
    int run = carry_run;
    {
        const int v = dc_diff;
        const int sz = size_category(v);
        const uint2 e = dc_tab[sz & 15];
        const bool ok = k == 0 && sz <= 11;
        err |= (k == 0 && sz > 11) ? 1 : 0;
        sink.put(ok ? (e.x | value_bits(v, sz)) : 0u, ok ? e.y : 0u);
    }
    const uint2 zrl = ac_tab[0xF0], eob = ac_tab[0];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int v = c[j];
        const bool dcpos = k == 0 && j == 0;
        const bool nz = v != 0 && !dcpos;
        while (__any(nz && run >= 16)) {
            const bool z = nz && run >= 16;
            sink.put(z ? zrl.x : 0u, z ? zrl.y : 0u);
            run -= z ? 16 : 0;
        }
        const int sz = size_category(v);
        const uint2 e = ac_tab[((run << 4) | (sz > 15 ? 15 : sz)) & 255];
        const bool ok = nz && e.y != 0u;
        err |= (nz && e.y == 0u) ? 1 : 0;
        sink.put(ok ? (e.x | value_bits(v, sz)) : 0u, ok ? e.y : 0u);
        run = nz ? 0 : run + (dcpos ? 0 : 1);
    }
    sink.put(k == 7 ? eob.x : 0u, k == 7 ? eob.y : 0u);
}

// Write-through store (sc1): the launch does not end with a write-back of dirty L2 lines (as in the transform kernel).
__device__ __forceinline__ void store_u32_wt(uint32_t *p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

// Cross-lane moves by DPP (one VALU instruction, no LDS): row_shr:n shifts inside rows of 16 lanes, row_bcast:15 / :31 hand the
// last lane of a row / of the lower half-wave to the rows behind it.  Lanes without a source (or outside row_mask) get 0.
template <int N>
__device__ __forceinline__ int dpp_row_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, false);
}
// Inclusive prefix sum over the 64 lanes of the wave: 4 steps inside the rows, 2 across them.
__device__ __forceinline__ uint32_t wave_prefix_sum_u32(uint32_t x) {
    int v = (int)x;
    v += dpp_row_shr<1>(v);
    v += dpp_row_shr<2>(v);
    v += dpp_row_shr<4>(v);
    v += dpp_row_shr<8>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false); // rows 1, 3 += total of the row before
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false); // rows 2, 3 += total of rows 0-1
    return (uint32_t)v;
}

template <int ABL> // ABL != 0: timing-only builds (tools/), wrong output
__global__ __launch_bounds__(kGroup * 64) void entropy_pack_kernel(const int16_t *__restrict__ zz, const HuffDev *__restrict__ tab,
                                                            unsigned long long blocks_per_frame, unsigned long long parts_per_frame,
                                                            unsigned long long groups_per_frame, uint32_t *__restrict__ stage,
                                                            uint32_t *__restrict__ nbits, uint32_t *__restrict__ gsum,
                                                            int *__restrict__ err_flag) {
    __shared__ uint2 ac_tab[256];
    __shared__ uint2 dc_tab[16];
    __shared__ uint32_t wbits[kGroup];
    __shared__ __attribute__((aligned(16))) uint32_t image_all[kGroup][512]; // kStageWords used, 512 so that two 16-byte stores per lane zero it
    __shared__ uint32_t str_all[kGroup][64 * kLaneWords];
    if (threadIdx.x < 256) ac_tab[threadIdx.x] = make_uint2(tab->ac_sym[threadIdx.x], tab->ac_bits[threadIdx.x]);
    if (threadIdx.x < 16) dc_tab[threadIdx.x] = make_uint2(tab->dc_sym[threadIdx.x], tab->dc_bits[threadIdx.x]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *image = image_all[wave];
    uint32_t *str = str_all[wave] + lane;
    static_assert(kStageWords <= 512, "image rows are 512 words");
    reinterpret_cast<uint4 *>(image)[lane] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4 *>(image)[64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    // one workgroup per (frame, group of kGroup partitions); a wave per partition = 8 blocks.  The coefficient load and the
    // zero-run scan need no table: they run in front of the barrier that publishes the tables, not behind it.
    const unsigned long long frame = blockIdx.x / groups_per_frame, g = blockIdx.x - frame * groups_per_frame;
    const unsigned long long pif = g * (unsigned long long)kGroup + (unsigned long long)wave;
    const unsigned long long part = frame * parts_per_frame + pif;
    const bool active = pif < parts_per_frame;
    uint32_t wave_bits = 0;
    const unsigned long long first_in_frame = pif * 8ull;
    const unsigned long long bif = first_in_frame + (unsigned long long)(lane >> 3); // block index inside the frame
    const int k = lane & 7;
    const bool valid = active && bif < blocks_per_frame;
    const unsigned long long frame_first = frame * blocks_per_frame;
    const unsigned long long blk = frame_first + (valid ? bif : blocks_per_frame - 1);
    int16_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dc_diff = 0;
    if (valid) {
        uint4 v;
        if (ABL & 256) { // timing only: no coefficient load (values from the lane id)
            const uint32_t hsh = (uint32_t)(blk * 2654435761u) ^ (uint32_t)(k * 40503u);
            v = make_uint4(hsh & 0x000f001fu, (hsh >> 3) & 0x0007000fu, (hsh >> 7) & 0x00030007u, k < 5 ? (hsh >> 11) & 0x00010003u : 0u);
        } else {
            v = *reinterpret_cast<const uint4 *>(zz + blk * 64 + k * 8);
        }
        c[0] = (int16_t)(v.x & 0xffff); c[1] = (int16_t)(v.x >> 16); c[2] = (int16_t)(v.y & 0xffff); c[3] = (int16_t)(v.y >> 16);
        c[4] = (int16_t)(v.z & 0xffff); c[5] = (int16_t)(v.z >> 16); c[6] = (int16_t)(v.w & 0xffff); c[7] = (int16_t)(v.w >> 16);
    }
    {
        // codec.py:34-35: DPCM over the blocks of one frame in raster order, the first block raw.  The previous block's DC
        // sits in lane - 8 of this wave, except for the wave's first block (one 2-byte load per wave).
        const int prev_in_wave = __shfl_up((int)c[0], 8, 64);
        if (k == 0 && valid) {
            int prev = prev_in_wave;
            if (lane == 0) prev = bif ? (int)zz[(blk - 1) * 64] : 0;
            dc_diff = bif ? (int)c[0] - prev : (int)c[0];
        }
    }
    int nz_mask = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) nz_mask |= (c[j] != 0 && !(k == 0 && j == 0)) ? (1 << j) : 0;
    const int cnt = (k == 0) ? 7 : 8;
    int az = nz_mask == 0;
    int tz = az ? cnt : (__clz(nz_mask) - 24);
    // (a lane only uses what comes from inside its own group of 8: k >= d, so the 16-lane rows of row_shr are wide enough)
#define TIC_CARRY_STEP(D)                                          \
    {                                                              \
        const int pa = dpp_row_shr<D>(az), pt = dpp_row_shr<D>(tz); \
        if (k >= D) {                                              \
            tz = az ? pt + tz : tz;                                \
            az = az & pa;                                          \
        }                                                          \
    }
    TIC_CARRY_STEP(1)
    TIC_CARRY_STEP(2)
    TIC_CARRY_STEP(4)
#undef TIC_CARRY_STEP
    int carry = dpp_row_shr<1>(tz);
    if (k == 0) carry = 0;
    __syncthreads();
    if (active) {
    // ---- the one walk: symbols -> the lane's private bit string ---------------------------------------------------------
    uint32_t my_bits = 0, last_word = 0;
    if (ABL & 4) {
        int err = 0;
        if (valid) my_bits = (uint32_t)walk_lane<false>(c, k, carry, dc_diff, ac_tab, dc_tab, (LaneSink *)nullptr, &err);
    } else if (!(ABL & 56)) { // the product
        LaneSinkB sink;
        sink.str = str;
        sink.cur = 0u;
        sink.sh = 0u;
        sink.full = 0u;
        int err = 0;
        // a zero run of 16 or more can only reach a lane's first non-zero entry (a lane holds 8): decided per wave
        const int lead = nz_mask ? __builtin_ctz((unsigned)nz_mask) - (k == 0 ? 1 : 0) : 0;
        const bool need_zrl = valid && nz_mask != 0 && carry + lead >= 16;
        if (__any(need_zrl)) {
            if (valid) walk_lane_lean<true>(c, k, carry, dc_diff, ac_tab, dc_tab, &sink, &err);
        } else {
            if (valid) walk_lane_lean<false>(c, k, carry, dc_diff, ac_tab, dc_tab, &sink, &err);
        }
        my_bits = valid ? sink.bits() : 0u;
        last_word = sink.cur;
        if (err && valid) atomicMax(err_flag, 1);
    } else if (ABL & 32) { // experiment: the first walk (walk_lane: DC special case, ZRL loop, tests as branches), branching sink
        LaneSinkB sink;
        sink.str = str;
        sink.cur = 0u;
        sink.sh = 0u;
        sink.full = 0u;
        int err = 0;
        if (valid) walk_lane<true>(c, k, carry, dc_diff, ac_tab, dc_tab, &sink, &err);
        my_bits = valid ? sink.bits() : 0u;
        last_word = sink.cur;
        if (err && valid) atomicMax(err_flag, 1);
    } else if (ABL & 8) { // branching walk, branch-free sink
        LaneSink sink;
        sink.str = str;
        sink.cur = 0u;
        sink.sh = 0u;
        sink.full = 0u;
        int err = 0;
        if (valid) walk_lane<true>(c, k, carry, dc_diff, ac_tab, dc_tab, &sink, &err);
        my_bits = valid ? sink.bits() : 0u;
        last_word = sink.cur;
        if (err && valid) atomicMax(err_flag, 1);
    } else { // uniform control flow throughout
        LaneSink sink;
        sink.str = str;
        sink.cur = 0u;
        sink.sh = 0u;
        sink.full = 0u;
        int err = 0;
        walk_pack(c, k, carry, dc_diff, ac_tab, dc_tab, sink, err); // (lanes past the frame's last block walk zeros)
        my_bits = valid ? sink.bits() : 0u;
        last_word = sink.cur;
        if (err && valid) atomicMax(err_flag, 1);
    }
    // prefix of the lanes' bit counts over the wave (lanes are in stream order: block, then scan position)
    const uint32_t incl = wave_prefix_sum_u32(my_bits);
    wave_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // ---- lane strings -> the wave's image (bit 0 of the partition = MSB of word 0) ---------------------------------------
    if (!(ABL & 64) && !(ABL & 2)) { // every string word ORed into its two image words (LDS atomics into the zeroed image)
        const uint32_t lane_pos = incl - my_bits;
        const uint32_t w0 = lane_pos >> 5, sh = lane_pos & 31u;
        const int nw = (int)((my_bits + 31u) >> 5);
        for (int w = 0; w < nw; w++) {
            const uint32_t v = (w == nw - 1 && (my_bits & 31u)) ? last_word : str[w * 64]; // the partial word never left the lane
            atomicOr(image + w0 + w, v >> sh);
            if (sh) atomicOr(image + w0 + w + 1, v << (32u - sh));
        }
    } else if ((ABL & 64) && my_bits != 0u) {
        // Experiment (pack 29.2 us against 28.0): one LDS operation per IMAGE word the lane's bits touch: the word takes the low bits of string word j - 1 and the high
        // bits of string word j.  Words that lie inside the lane's bit range belong to it alone (plain store); only its first
        // and last word can be shared with the neighbouring lanes (atomic OR into the zeroed image).
        const uint32_t lane_pos = incl - my_bits;
        const uint32_t w0 = lane_pos >> 5, sh = lane_pos & 31u;
        const uint32_t nw = (my_bits + 31u) >> 5;        // string words
        const uint32_t nwi = (sh + my_bits + 31u) >> 5;  // image words
        const bool partial = (my_bits & 31u) != 0u;      // the partial last string word never left the lane's register
        uint32_t carry = 0u;
        for (uint32_t j = 0; j < nwi; j++) {
            uint32_t v = 0u;
            if (j < nw) v = (j == nw - 1u && partial) ? last_word : str[j * 64u];
            const uint32_t out = carry | (v >> sh);
            carry = sh ? v << (32u - sh) : 0u;
            const bool inside = (j != 0u || sh == 0u) && (32u * j + 32u - sh <= my_bits);
            if (inside) image[w0 + j] = out;
            else atomicOr(image + w0 + j, out);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t nwords = (wave_bits + 31u) >> 5;
    uint32_t *slot = stage + part * (unsigned long long)kStageWords;
    if (!(ABL & 1))
        for (uint32_t i = (uint32_t)lane; i < nwords; i += 64u) store_u32_wt(slot + i, image[i]);
    if (lane == 0) nbits[part] = wave_bits;
    }
    if (lane == 0) wbits[wave] = wave_bits;
    __syncthreads();
    if (threadIdx.x == 0) { // bits of the group: the coarse level of the stream offsets, one plain store
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < kGroup; i++) t += wbits[i];
        gsum[blockIdx.x] = t;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The packing kernel, round-3 form: a LANE per block, a wave per partition of 64 blocks.
//
// Round 2's kernel above (8 lanes per block, 8 scan positions per lane) is instruction-bound at 471 vector instructions per
// wave of 8 blocks (59 per block) whatever the content: the lane that owns scan positions 0..7 is never idle, so a wave walks all
// eight of its positions at full cost even when the other 56 coefficients of every block are zero (Lenna 23.7 us against
// 24.9 us for noise, 4096^2), and 46 % of its LDS cycles are bank conflicts of the 8-byte table look-ups and the merge's atomics.
// Here a lane owns a whole block: its 64 coefficients sit in 32 registers (coalesced 16-byte loads, transposed through LDS with
// a conflict-free 144-byte block stride), the walk over the scan positions is unrolled, a position costs ~26 vector
// instructions only where some lane of the wave has a non-zero coefficient there (exec-masked body, skipped by the hardware
// when no lane is active), nothing where none has, and the walk ends at the wave's last non-zero group of 8 positions.  No
// zero-run scan across lanes (the run is position minus last non-zero position), one 4-byte table word per symbol.
// Every lane packs into its own bit string in LDS (word i of lane l at str[i * 64 + l]: 64 lanes, 64 banks), at most W words;
// a prefix over the lanes then places the strings in the wave's image of its bit range exactly as above.  A block that needs
// more than W words (32 * W bits; noise at quality >= ~85) raises error 4 and the stage is run again with the kernel above,
// which takes any block the format allows.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kPB = 64;          // blocks per partition = per wave
constexpr int kGroupL = 4;       // partitions (waves) per workgroup = per group sum
constexpr int kTStrideB = 144;   // bytes per block in the transpose buffer: 128 + 16, so that the lanes' 16-byte reads (stride
                                 // 144 B = 9 bank groups of 16 B, odd) fall on 16 different bank groups in every group of 16 lanes
template <int W>
struct PackL {
    static constexpr int kImageWords = kPB * W + 4;                       // the wave's bit range, whole 16-byte pieces
    static constexpr int kStrWords = (W + 1) * 64;                        // lane strings (+ one row that takes overflowing words)
    static constexpr int kWaveWords = (kStrWords + kImageWords) > (kPB * kTStrideB / 4) ? (kStrWords + kImageWords) : (kPB * kTStrideB / 4);
    static constexpr int kStageWords = kImageWords;                        // staging slot of a partition
};

template <int W, int ABL = 0> // ABL != 0: timing-only instantiations (1: no walk, 2: no coefficient loads, 4: no slot store, 8: no merge); the library instantiates <W, 0> only
__global__ __launch_bounds__(kGroupL * 64) void entropy_pack_lane_kernel(const int16_t *__restrict__ zz, const HuffDev *__restrict__ tab,
                                                                     unsigned long long blocks_per_frame, unsigned long long parts_per_frame,
                                                                     unsigned long long groups_per_frame, uint32_t *__restrict__ stage,
                                                                     uint32_t *__restrict__ nbits, uint32_t *__restrict__ gsum,
                                                                     int *__restrict__ err_flag) {
    typedef PackL<W> P;
    __shared__ uint32_t ac_tab[256];
    __shared__ uint32_t dc_tab[16];
    __shared__ uint32_t wbits[kGroupL];
    __shared__ __attribute__((aligned(16))) uint32_t buf_all[kGroupL][P::kWaveWords];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ac_tab[threadIdx.x] = tab->ac_pack[threadIdx.x]; // 256 threads, 256 entries
    if (threadIdx.x < 16) dc_tab[threadIdx.x] = tab->dc_pack[threadIdx.x];
    uint32_t *buf = buf_all[wave];
    const unsigned long long frame = blockIdx.x / groups_per_frame, g = blockIdx.x - frame * groups_per_frame;
    const unsigned long long pif = g * (unsigned long long)kGroupL + (unsigned long long)wave; // partition inside the frame
    const unsigned long long part = frame * parts_per_frame + pif;
    const bool active = pif < parts_per_frame;
    const unsigned long long first_in_frame = pif * (unsigned long long)kPB;
    const unsigned long long nblk_part = !active ? 0ull : (blocks_per_frame - first_in_frame < (unsigned long long)kPB ? blocks_per_frame - first_in_frame : (unsigned long long)kPB);
    const bool valid = (unsigned long long)lane < nblk_part; // this lane's block exists
    uint32_t wave_bits = 0;
    if (active) {
        // ---- coefficients: 8 KiB per wave in eight coalesced 1 KiB loads, transposed through LDS: lane l ends with block l --------
        const int16_t *src = zz + (frame * blocks_per_frame + first_in_frame) * 64ull;
        uint4 ld[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned long long bl = (unsigned long long)(i * 8 + (lane >> 3)); // block of this 16-byte piece
            if (ABL & 2) ld[i] = make_uint4((uint32_t)(lane * 2654435761u + i) & 0x00030007u, (uint32_t)(bl * 40503u) & 0x00010003u, 0u, i < 2 ? 0x00010000u : 0u);
            else ld[i] = bl < nblk_part ? *reinterpret_cast<const uint4 *>(src + bl * 64ull + (unsigned long long)(lane & 7) * 8ull) : make_uint4(0u, 0u, 0u, 0u);
        }
        int prev_dc = 0; // DC of the block in front of the wave's first (codec.py:34-35: DPCM over the blocks of one frame, the first raw)
        if (lane == 0 && first_in_frame != 0ull) prev_dc = (int)src[-64];
        char *tb = reinterpret_cast<char *>(buf);
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<uint4 *>(tb + (i * 8 + (lane >> 3)) * kTStrideB + (lane & 7) * 16) = ld[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t c[32]; // c[j] = scan positions 2j (low half) and 2j + 1 (high half) of the lane's block
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tb + lane * kTStrideB + i * 16);
            c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (the buffer is about to be reused for the strings and the image)
        const int dc = (int)(int16_t)(c[0] & 0xffffu);
        int dc_prev = __shfl_up(dc, 1, 64);
        if (lane == 0) dc_prev = prev_dc;
        const int dc_diff = dc - dc_prev; // (frame's first block: prev_dc = 0)
        // groups of 8 scan positions in which some lane of the wave has a non-zero AC coefficient: where the walk may stop
        uint32_t wave_groups = 0;
#pragma unroll
        for (int gq = 0; gq < 8; gq++) {
            const uint32_t any = (gq == 0 ? (c[0] & 0xffff0000u) : c[4 * gq]) | c[4 * gq + 1] | c[4 * gq + 2] | c[4 * gq + 3];
            if (__ballot(any != 0u && valid) != 0ull) wave_groups |= 1u << gq;
        }
        if (ABL & 1) wave_groups = 0u;
        __syncthreads(); // the tables are in LDS
        // ---- the walk: symbols -> the lane's private bit string -------------------------------------------------------------
        uint32_t *str = buf + lane;                        // word i of this lane: str[i * 64]
        uint32_t *image = buf + P::kStrWords;              // the wave's bit range (bit 0 = MSB of word 0)
        uint32_t cur = 0u, sh = 0u, full = 0u;             // word under construction (MSB first), bits used in it, complete words behind it
        int err = 0;
        // branch-free append of the n-bit symbol v (1 <= n <= 26; n = 0, v = 0 is a no-op): the word under construction goes to
        // LDS whenever it is complete (row W takes whatever overflows the lane's W words and is never read)
        auto put = [&](uint32_t v, uint32_t n) {
            const uint32_t x = v << ((32u - n) & 31u);                          // the symbol left-aligned
            const uint32_t hi = cur | (x >> sh);
            const uint32_t lo = __builtin_amdgcn_alignbit(x, 0u, sh);            // the bits that did not fit (0 for sh = 0)
            const uint32_t nsh = sh + n;
            const bool done = nsh >= 32u;
            str[(full < (uint32_t)W ? full : (uint32_t)W) * 64u] = hi;
            cur = done ? lo : hi;
            sh = nsh & 31u;
            full += done ? 1u : 0u;
        };
        const uint32_t zrl = ac_tab[0xF0], eob = ac_tab[0];
        if (valid) { // DC: category code + value bits (huffman.py:41-63 with dc_ac = DC)
            const int sz = size_category(dc_diff);
            const uint32_t e = dc_tab[sz & 15];
            const bool bad = e == 0u || sz > 11; // (|difference| up to 65535: size 16 wraps to the valid entry 0, hence the bound)
            err |= bad ? 1 : 0;
            put(bad ? 0u : ((e & 0x7ffffffu) | value_bits(dc_diff, sz)), bad ? 0u : (e >> 27));
        }
        // The AC walk, a group of 8 scan positions at a time (groups in which no lane of the wave has a non-zero coefficient are
        // skipped; the walk ends behind the wave's last such group).  Phase 1 of a group is eight INDEPENDENT chains - coefficient ->
        // size category -> table word -> symbol and length (length 0 for a zero coefficient) - which the hardware overlaps; phase 2
        // appends the eight symbols in order (the only serial part: ~4 dependent instructions per symbol).  (The first form of this
        // kernel ran `if (v != 0) { look up; append }` position by position: one dependent chain of ~30 instructions and an LDS round
        // trip per position, 26 us for 4096^2 noise at 4 waves per SIMD - the same as the 8-lane kernel it was to replace.)
        uint32_t last6 = 0u; // (position of the last non-zero coefficient, DC counting as one) << 6
#pragma unroll
        for (int gq = 0; gq < 8; gq++) {
            if ((wave_groups >> gq) == 0u) break;               // no lane has anything from here on
            if (((wave_groups >> gq) & 1u) == 0u) continue;     // nothing in this group (zero runs are position differences: nothing to count)
            uint32_t sym[8], len[8], trun[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p = gq * 8 + j;
                if (p == 0) { sym[j] = 0u; len[j] = 0u; trun[j] = 0u; continue; } // (the DC went first)
                const int v = (p & 1) ? ((int)c[p >> 1] >> 16) : ((int)(c[p >> 1] << 16) >> 16);
                const bool nz = v != 0 && valid;
                const uint32_t t = (uint32_t)((p - 1) << 6) - last6; // zero run in front of this coefficient, << 6
                int sz = size_category(v);                           // 1..16 for an int16; sizes above 10 have no code (KeyError in the reference)
                sz = sz > 11 ? 11 : sz;
                const uint32_t e = ac_tab[((t >> 2) & 0xf0u) + (uint32_t)sz]; // ((run mod 16) << 4) | size: the ZRLs below take the multiples of 16
                err |= (nz && e == 0u) ? 1 : 0;
                sym[j] = nz ? ((e & 0x7ffffffu) | (e ? value_bits(v, sz) : 0u)) : 0u;
                len[j] = nz ? (e >> 27) : 0u;
                trun[j] = nz ? t : 0u;
                last6 = nz ? ((uint32_t)p << 6) : last6;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p = gq * 8 + j;
                if (p >= 17 && __builtin_expect(__ballot(trun[j] >= (16u << 6)) != 0ull, 0)) { // ZRL = (15,0) for every 16 zeros, huffman.py:26-28
                    for (uint32_t t = trun[j]; t >= (16u << 6); t -= 16u << 6) put(zrl & 0x7ffffffu, zrl >> 27);
                }
                put(sym[j], len[j]);
            }
        }
        if (valid) put(eob & 0x7ffffffu, eob >> 27); // EOB = (0,0) always closes the block (huffman.py:33)
        const uint32_t my_bits = valid ? full * 32u + sh : 0u;
        if (my_bits > 32u * (uint32_t)W) err |= 4; // the block does not fit a lane string: the caller falls back to the 8-lane kernel
        if (err) atomicMax(err_flag, (err & 1) ? 1 : 4);
        // ---- lane strings -> the wave's image ---------------------------------------------------------------------------------
        const uint32_t incl = wave_prefix_sum_u32(my_bits);
        wave_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const bool fits = __ballot(my_bits > 32u * (uint32_t)W) == 0ull;
        if (!fits) wave_bits = 0; // (error 4 is raised: this run's output is discarded; keep every index in range)
        const uint32_t nwords = (wave_bits + 31u) >> 5;
        for (uint32_t i = (uint32_t)lane * 4u; i < nwords + 1u; i += 256u) *reinterpret_cast<uint4 *>(image + i) = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (fits && !(ABL & 8)) {
            const uint32_t lane_pos = incl - my_bits;
            const uint32_t w0 = lane_pos >> 5, s0 = lane_pos & 31u;
            const int nw = (int)((my_bits + 31u) >> 5);
            for (int wq = 0; wq < nw; wq++) {
                const uint32_t v = (wq == nw - 1 && (my_bits & 31u)) ? cur : str[wq * 64]; // the partial word never left the lane
                atomicOr(image + w0 + wq, v >> s0);
                if (s0) atomicOr(image + w0 + wq + 1, v << (32u - s0));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t *slot = stage + part * (unsigned long long)P::kStageWords;
        for (uint32_t i = (uint32_t)lane * 4u; i < nwords && !(ABL & 4); i += 256u) { // (whole 16-byte pieces: the slot and the image are padded)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const uint4 q = *reinterpret_cast<const uint4 *>(image + i);
            const u32x4 d = {q.x, q.y, q.z, q.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(slot + i), "v"(d) : "memory");
        }
        if (lane == 0) nbits[part] = wave_bits;
    } else {
        __syncthreads();
    }
    if (lane == 0) wbits[wave] = wave_bits;
    __syncthreads();
    if (threadIdx.x == 0) gsum[blockIdx.x] = wbits[0] + wbits[1] + wbits[2] + wbits[3]; // bits of the group: the coarse level of the stream offsets
}

constexpr int kTileGroups = 256; // groups per tile sum: the second level of the offsets, only for frames of many groups

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (unsigned long long)__shfl_xor((long long)v, d, 64);
    return v;
}

// Bits per tile of kTileGroups groups.  Only launched for frames of more than `direct` groups (beyond 8192^2 pixels), where a
// placing workgroup summing every group sum before it would re-read too much.
__global__ __launch_bounds__(256) void entropy_tilesum_kernel(const uint32_t *__restrict__ gsum, unsigned long long groups_per_frame,
                                                              unsigned long long tiles_per_frame, unsigned long long *__restrict__ tile_sum) {
    __shared__ unsigned long long ws[4];
    const unsigned long long frame = blockIdx.x / tiles_per_frame, tile = blockIdx.x - frame * tiles_per_frame;
    const unsigned long long j = tile * (unsigned long long)kTileGroups + threadIdx.x;
    unsigned long long sum = (threadIdx.x < kTileGroups && j < groups_per_frame) ? (unsigned long long)gsum[frame * groups_per_frame + j] : 0ull;
    sum = wave_sum_u64(sum);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

constexpr int kSlotLds = 64; // words of every partition's slot a workgroup fetches into LDS up front (8-block partitions: typically 55 are used)

// Places kPlace partitions in the frame's stream.  One workgroup per (frame, kPlace partitions); it owns the output words
// whose first bit lies in its partitions' bit range and assembles each from the partitions that meet in it.  The kernel is
// latency-bound (4 waves per workgroup, a few words per thread), so everything it needs from memory - group sums, bit
// counts, the head of every slot - is requested at once, before the first dependent step.
// kPlace: partitions per workgroup (a multiple of kGrp, the partitions per packing workgroup = per group sum); kSlotWords: words
// per staging slot; kHeads: the head of every slot is fetched into LDS up front (partitions of 8 blocks: a slot seldom has more than
// 64 words); partitions of 64 blocks are read from memory, every word by the lane that assembles it (coalesced).
template <int ABL, int kPlace, int kGrp, int kSlotWords, bool kHeads> // ABL != 0: timing-only builds (tools/), wrong output
__global__ __launch_bounds__(256) void entropy_place_kernel(const uint32_t *__restrict__ stage, const uint32_t *__restrict__ nbits,
                                                            const uint32_t *__restrict__ gsum, const unsigned long long *__restrict__ tile_sum,
                                                            unsigned long long parts_per_frame, unsigned long long groups_per_frame,
                                                            unsigned long long places_per_frame, unsigned long long tiles_per_frame,
                                                            unsigned char *__restrict__ out, unsigned long long out_frame_stride,
                                                            unsigned long long cap_words, int h, int w, int quality,
                                                            unsigned long long *__restrict__ lens, int *__restrict__ err_flag,
                                                            int *__restrict__ err_next, unsigned long long *__restrict__ status) {
    constexpr int kN = kPlace + 3;             // its own partitions and the three behind them (the last word may run into those)
    constexpr int kPer = (kN + 3) / 4;         // slots a wave fetches
    __shared__ uint32_t roff[kN + 1]; // first bit of partition p0 + i, counted from bit 0 of word base_word (empty past the frame's end)
    __shared__ unsigned long long base_word; // stream word that holds the workgroup's first bit
    __shared__ unsigned long long ws[2][4];
    __shared__ int long_slot;
    __shared__ uint32_t lds[kHeads ? kN * kSlotLds : 1];
    const unsigned long long frame = blockIdx.x / places_per_frame, gp = blockIdx.x - frame * places_per_frame;
    const unsigned long long p0 = gp * (unsigned long long)kPlace;
    const unsigned long long g = gp * (unsigned long long)(kPlace / kGrp); // first packing group of this workgroup
    const uint32_t *fn = nbits + frame * parts_per_frame;
    const uint32_t *fstage = stage + (frame * parts_per_frame + p0) * (unsigned long long)kSlotWords;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- every load of the prologue, back to back --------------------------------------------------------------------------
    const unsigned long long pl = p0 + (unsigned long long)lane;
    const unsigned long long nb = (wave == 0 && lane < kN && pl < parts_per_frame) ? (unsigned long long)fn[pl] : 0ull;
    unsigned long long before = 0, total = 0;
    {   // first bit of partition p0 = the group sums before group g (through the tile sums when the frame has them); the
        // frame's first workgroup also needs the frame's total
        const uint32_t *fg = gsum + frame * groups_per_frame;
        unsigned long long first = 0; // first group summed directly
        if (tiles_per_frame) {
            const unsigned long long *fts = tile_sum + frame * tiles_per_frame;
            const unsigned long long tile = g / (unsigned long long)kTileGroups;
            first = tile * (unsigned long long)kTileGroups;
            const unsigned long long upto = gp == 0 ? tiles_per_frame : tile;
            for (unsigned long long j = threadIdx.x; j < upto; j += 256ull) {
                const unsigned long long v = fts[j];
                total += v;
                if (j < tile) before += v;
            }
        }
        const unsigned long long upto = (ABL & 4) ? 0ull : ((gp == 0 && !tiles_per_frame) ? groups_per_frame : g);
        for (unsigned long long j0 = first; j0 < upto; j0 += 8ull * 256ull) { // 8 independent loads in flight per thread
            uint32_t v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const unsigned long long j = j0 + (unsigned long long)(r * 256) + threadIdx.x;
                v[r] = j < upto ? fg[j] : 0u;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const unsigned long long j = j0 + (unsigned long long)(r * 256) + threadIdx.x;
                if (!tiles_per_frame) total += v[r];
                if (j < g) before += v[r];
            }
        }
    }
    uint32_t head[kPer];
#pragma unroll
    for (int r = 0; r < kPer; r++) {
        const int i = wave + 4 * r;
        head[r] = (kHeads && !(ABL & 2) && i < kN && p0 + (unsigned long long)i < parts_per_frame) ? fstage[(unsigned long long)i * (unsigned long long)kSlotWords + lane] : 0u;
    }
    // ---- offsets ---------------------------------------------------------------------------------------------------------------
    if (tiles_per_frame) { // frames beyond 8192^2: 64-bit sums
        before = wave_sum_u64(before);
        total = wave_sum_u64(total);
    } else { // a frame of at most 8,192 groups holds fewer than 2^32 bits: 32-bit sums by DPP (six instructions, no LDS round trips)
        before = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_sum_u32((uint32_t)before), 63);
        total = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_sum_u32((uint32_t)total), 63);
    }
    if (lane == 0) {
        ws[0][wave] = before;
        ws[1][wave] = total;
    }
#pragma unroll
    for (int r = 0; r < kPer; r++) {
        const int i = wave + 4 * r;
        if (kHeads && i < kN) lds[i * kSlotLds + lane] = head[r];
    }
    __syncthreads();
    const unsigned long long frame_bits = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
    if (wave == 0) { // bit offsets of the kN partitions: a prefix sum over the lanes
        const unsigned long long base = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
        // positions relative to the word that holds the workgroup's first bit: 35 partitions of at most 13.9 Kbit fit 32 bits
        const uint32_t incl = wave_prefix_sum_u32((uint32_t)nb);
        const uint32_t a = (uint32_t)(base & 31ull);
        if (lane < kN) roff[lane] = a + (incl - (uint32_t)nb);
        if (lane == kN - 1) roff[kN] = a + incl;
        const int any_long = !kHeads || __any(nb > (unsigned long long)(kSlotLds * 32)); // a slot longer than its LDS copy: read the slots directly
        if (lane == 0) {
            long_slot = any_long;
            base_word = base >> 5;
        }
    }
    __syncthreads();
    const bool in_lds = long_slot == 0;
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + frame * out_frame_stride + 16) + base_word;
    const unsigned long long room_words = cap_words > base_word ? cap_words - base_word : 0ull; // words of the caller's buffer from base_word on
    auto assemble = [&](auto fetch) {
        // words that lie inside one partition: a funnel shift of two staged words, the shift the same for the whole partition
        for (int i = wave; i < kPlace; i += 4) {
            const uint32_t b = roff[i], e = roff[i + 1];
            const uint32_t wfirst = (b + 31u) >> 5, wend = e >> 5;
            const uint32_t s = (32u - (b & 31u)) & 31u; // position of the first such word inside the partition
            for (uint32_t wr = wfirst + (uint32_t)lane; wr < wend; wr += 64u) {
                const uint32_t x = ((wr << 5) - b) >> 5;
                uint32_t v = fetch(i, x);
                if (s) v = (v << s) | (fetch(i, x + 1u) >> (32u - s));
                if (wr < room_words) store_u32_wt(dst + wr, __builtin_bswap32(v)); // the stream is MSB-first bytes
            }
        }
        // words in which partitions meet: one per partition end that is not word-aligned, assembled piece by piece.  The
        // workgroup owns a word if its first bit lies in the workgroup's range (the others belong to the workgroup before).
        if (threadIdx.x < kPlace) {
            int i = (int)threadIdx.x;
            const uint32_t e = roff[i + 1];
            if ((e & 31u) && e > roff[i]) {
                const uint32_t wr = e >> 5, b0 = wr << 5;
                while (i > 0 && roff[i] > b0) i--;
                if (roff[i] <= b0) {
                    uint32_t word = 0u;
                    uint32_t filled = 0u;
                    while (filled < 32u && i < kN) {
                        const uint32_t pbeg = roff[i], pend = roff[i + 1], pos = b0 + filled;
                        if (pend > pos) {
                            const uint32_t lb = pos - pbeg, room = pend - pos;
                            const uint32_t take = room < 32u - filled ? room : 32u - filled;
                            const uint32_t w0 = lb >> 5, sh = lb & 31u;
                            uint32_t v = fetch(i, w0) << sh;
                            if (sh && 32u - sh < take) v |= fetch(i, w0 + 1u) >> (32u - sh);
                            v = take == 32u ? v : (v >> (32u - take)) << (32u - take); // keep the top `take` bits
                            word |= v >> filled;
                            filled += take;
                        }
                        i++;
                    }
                    if (wr < room_words) store_u32_wt(dst + wr, __builtin_bswap32(word));
                }
            }
        }
    };
    if (ABL & 1) {
    } else if (in_lds)
        assemble([&](int i, uint32_t w0) { return lds[i * kSlotLds + (int)w0]; });
    else
        assemble([&](int i, uint32_t w0) { return fstage[(unsigned long long)i * (unsigned long long)kSlotWords + w0]; });
    if (gp == 0 && threadIdx.x == 0) { // make_header (codec.py:102-114), the frame's length, the caller's status
        uint32_t *hdr = reinterpret_cast<uint32_t *>(out + frame * out_frame_stride);
        hdr[0] = (uint32_t)h; // struct.pack("III") little-endian == native order here
        hdr[1] = (uint32_t)w;
        hdr[2] = (uint32_t)quality;
        hdr[3] = 0u;
        if (lens) lens[frame] = 16ull + (frame_bits + 7ull) / 8ull;
        const bool over = ((frame_bits + 31ull) >> 5) > cap_words;
        if (over) atomicMax(err_flag, 2);
        if (frame == 0ull) {
            *err_next = 0; // the flag the NEXT call uses (two flags in turn: no memset between calls)
            if (status) { // single-frame calls: total and error straight into the caller's (host-mapped) status block
                status[0] = frame_bits;
                const int e = *err_flag; // set by the packing kernel, complete before this launch
                status[1] = (unsigned long long)(over ? (e > 2 ? e : 2) : e);
            }
        }
    }
}

} // namespace

// Workspace layout: [tile sums u64 x cap | group sums u32 x cap | bits per partition u32 x cap | staging slots]; cap = partitions
// of the 8-block form (the 64-block form has an eighth of them and slots eight times as long, within a few words: the staging area
// of the one covers the other).  Both functions below derive cap from the same formula, so a workspace sized by the first always
// passes the check of the second.
constexpr int kLaneW = 16;                                   // words per block of the lane-per-block packing kernel: 512 bits
static constexpr size_t kPerPart = 8 + 4 + 4 + (size_t)kStageWords * 4;
static_assert((size_t)PackL<kLaneW>::kStageWords <= 8 * (size_t)kStageWords, "a 64-block slot fits the room of eight 8-block slots");

size_t entropy_fused_work_bytes(size_t nblocks_total) {
    const size_t npart = (nblocks_total + 7) / 8 + 8 * 9; // (+ a partition of 64 blocks rounded up per frame end, generously)
    return npart * kPerPart + 256;
}

hipError_t entropy_gpu_fused(const int16_t *d_zz, size_t blocks_per_frame, int nframes, const HuffDev *d_tab, void *d_work,
                             size_t work_bytes, void *d_out, size_t out_frame_stride, size_t cap_words, int h, int w, int quality,
                             unsigned long long *d_lens, unsigned long long *d_status, int *d_err, int *d_err_next, int mode,
                             hipStream_t stream, hipStream_t place_stream, hipEvent_t pack_done) {
    if (blocks_per_frame == 0 || nframes <= 0) return hipSuccess;
    const bool lane_form = mode == kEntropyLanePerBlock;
    const size_t part_blocks = lane_form ? (size_t)kPB : 8, grp = lane_form ? (size_t)kGroupL : (size_t)kGroup;
    const size_t place = lane_form ? 8 : 32; // partitions per placing workgroup
    const size_t parts_per_frame = (blocks_per_frame + part_blocks - 1) / part_blocks;
    const size_t groups_per_frame = (parts_per_frame + grp - 1) / grp;
    const size_t npart = parts_per_frame * (size_t)nframes, ngroup = groups_per_frame * (size_t)nframes;
    if (work_bytes < 256 + kPerPart) return hipErrorInvalidValue;
    const size_t cap_parts = (work_bytes - 256) / kPerPart; // in 8-block partitions
    // frames of more than `direct` groups take the offsets in two levels (tile sums); TIC_ENT_DIRECT_GROUPS moves the switch
    size_t direct = 8192;
    if (const char *e = test_hook("TIC_ENT_DIRECT_GROUPS")) direct = (size_t)strtoull(e, nullptr, 10); // (tic_hooks.h: off unless TIC_TEST_HOOKS=1)
    if (direct > 8192) direct = 8192; // the placing kernel's 32-bit sums rely on it
    const size_t tiles_per_frame = groups_per_frame > direct ? (groups_per_frame + kTileGroups - 1) / kTileGroups : 0;
    const size_t ntiles = tiles_per_frame * (size_t)nframes;
    const size_t slot_words = lane_form ? (size_t)PackL<kLaneW>::kStageWords : (size_t)kStageWords;
    if (npart > cap_parts || ngroup > cap_parts || ngroup > 0x7fffffffull || npart * slot_words > cap_parts * (size_t)kStageWords) return hipErrorInvalidValue;
    unsigned long long *tile_sum = (unsigned long long *)d_work; // (never more tiles than groups, never more groups than partitions)
    uint32_t *gsum = (uint32_t *)(tile_sum + cap_parts);
    uint32_t *nbits = gsum + cap_parts;
    uint32_t *stage = nbits + cap_parts;
    const dim3 pack_grid((unsigned)ngroup);
    if (lane_form) {
#define TIC_PACKL(A)                                                                                                                             \
    hipLaunchKernelGGL((entropy_pack_lane_kernel<kLaneW, A>), pack_grid, dim3(kGroupL * 64), 0, stream, d_zz, d_tab, (unsigned long long)blocks_per_frame, \
                       (unsigned long long)parts_per_frame, (unsigned long long)groups_per_frame, stage, nbits, gsum, d_err)
#ifdef TIC_ABLATION
        static const int labl = getenv("TIC_ENT_ABL") ? atoi(getenv("TIC_ENT_ABL")) : 0;
        switch (labl) {
        case 1: TIC_PACKL(1); break;
        case 2: TIC_PACKL(2); break;
        case 3: TIC_PACKL(3); break;
        case 4: TIC_PACKL(4); break;
        case 7: TIC_PACKL(7); break;
        case 8: TIC_PACKL(8); break;
        case 15: TIC_PACKL(15); break;
        default: TIC_PACKL(0); break;
        }
#else
        TIC_PACKL(0);
#endif
#undef TIC_PACKL
    } else {
#ifdef TIC_ABLATION
        static const int abl = getenv("TIC_ENT_ABL") ? atoi(getenv("TIC_ENT_ABL")) : 0;
#define TIC_PACK(A)                                                                                                            \
    hipLaunchKernelGGL(entropy_pack_kernel<A>, pack_grid, dim3(kGroup * 64), 0, stream, d_zz, d_tab, (unsigned long long)blocks_per_frame, \
                       (unsigned long long)parts_per_frame, (unsigned long long)groups_per_frame, stage, nbits, gsum, d_err)
        switch (abl) {
        case 1: TIC_PACK(1); break;
        case 2: TIC_PACK(2); break;
        case 3: TIC_PACK(3); break;
        case 4: TIC_PACK(4); break;
        case 7: TIC_PACK(7); break;
        case 8: TIC_PACK(8); break;
        case 16: TIC_PACK(16); break;
        case 32: TIC_PACK(32); break;
        case 64: TIC_PACK(64); break;
        case 256: TIC_PACK(256); break;
        default: TIC_PACK(0); break;
        }
#undef TIC_PACK
#else
        hipLaunchKernelGGL(entropy_pack_kernel<0>, pack_grid, dim3(kGroup * 64), 0, stream, d_zz, d_tab, (unsigned long long)blocks_per_frame,
                           (unsigned long long)parts_per_frame, (unsigned long long)groups_per_frame, stage, nbits, gsum, d_err);
#endif
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ntiles) {
        hipLaunchKernelGGL(entropy_tilesum_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, gsum, (unsigned long long)groups_per_frame,
                           (unsigned long long)tiles_per_frame, tile_sum);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const size_t places_per_frame = (parts_per_frame + place - 1) / place;
    if (place_stream != nullptr && place_stream != stream) { // the placing kernel on a stream of its own, behind the packing (and the tile sums)
        if (!pack_done) return hipErrorInvalidValue;
        if ((e = hipEventRecord(pack_done, stream)) != hipSuccess || (e = hipStreamWaitEvent(place_stream, pack_done, 0)) != hipSuccess) return e;
        stream = place_stream;
    }
#define TIC_PLACE_ARGS                                                                                                                     \
    dim3((unsigned)(places_per_frame * (size_t)nframes)), dim3(256), 0, stream, stage, nbits, gsum, tile_sum, (unsigned long long)parts_per_frame, \
        (unsigned long long)groups_per_frame, (unsigned long long)places_per_frame, (unsigned long long)tiles_per_frame, (unsigned char *)d_out,   \
        (unsigned long long)out_frame_stride, (unsigned long long)cap_words, h, w, quality, d_lens, d_err, d_err_next, d_status
    if (lane_form) {
        hipLaunchKernelGGL((entropy_place_kernel<0, 8, kGroupL, PackL<kLaneW>::kStageWords, false>), TIC_PLACE_ARGS);
    } else {
#ifdef TIC_ABLATION
        static const int pabl = getenv("TIC_PLACE_ABL") ? atoi(getenv("TIC_PLACE_ABL")) : 0;
#define TIC_PLACE(A) hipLaunchKernelGGL((entropy_place_kernel<A, 32, kGroup, kStageWords, true>), TIC_PLACE_ARGS)
        switch (pabl) {
        case 1: TIC_PLACE(1); break;
        case 2: TIC_PLACE(2); break;
        case 3: TIC_PLACE(3); break;
        case 4: TIC_PLACE(4); break;
        case 7: TIC_PLACE(7); break;
        default: TIC_PLACE(0); break;
        }
#undef TIC_PLACE
#else
        hipLaunchKernelGGL((entropy_place_kernel<0, 32, kGroup, kStageWords, true>), TIC_PLACE_ARGS);
#endif
    }
#undef TIC_PLACE_ARGS
    return hipGetLastError();
}

} // namespace tic
