// tic_entropy_dec_gpu.h - Huffman + run-length decode of a long stream on the device (see tic_entropy_dec_gpu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tic {

// The host decoder's look-up tables (tic_entropy.cpp EncTables::Dec): the next 11 / 16 stream bits -> (code length << 8) | symbol,
// 0 when no codeword (of at most 11 bits, for the short tables) is a prefix of them.
struct DecLutsDev {
    uint16_t dc11[2048];
    uint16_t ac11[2048];
    uint16_t ac16[65536];
    // the measure walk's tables: one look-up consumes a CHAIN of symbols (it needs block ends, not values): entry = (bits consumed << 3)
    // | (table of the next step: 0 mdc, 1 mac, 2 mlong) << 1 | (the chain ended with EOB); a window without a codeword has an entry too (tic_entropy.cpp chain_entry).  mdc: the next 11 stream bits with the DC category next;
    // mac: the next 12 bits inside the AC symbols; mlong: the AC codewords of 12..16 bits (index: the next 16 bits - 0xff40), one symbol.
    uint8_t mdc[2048];
    uint8_t mac[4096];
    uint8_t mlong[256];
    // the fused kernel's table: one look-up of the next 11 stream bits gives up to TWO AC symbols with their values' places
    // (tic_entropy.cpp dec_pair_luts_fill), then the long codewords one by one, then a zero entry (the slot of an index out of range)
    uint32_t ac2[2048];
    uint32_t long32[192 + 4];
};

struct DecStatus {
    int giveup;                 // != 0: something unusual on the true chain - the caller decodes the whole stream on the host
    int dc_out;                 // running DC behind the last block produced here
    unsigned long long m;       // blocks produced: the pixels of blocks [0, m) have been written
    unsigned long long pos_out; // first stream bit behind block m - 1
    uint32_t head[4];           // the stream's first 16 bytes as the kernels saw them (a caller that launched on a GUESS of the header compares)
};

// Where the pixels go and how the coefficients become pixels (the inverse stage of decode(), codec.py:46-70): the arguments of
// idct_kernel, minus the coefficient array - the fused kernel never writes one.
struct DecIdctArgs {
    uint8_t *out;      // device, uint8 [h][stride]
    int h, w;
    long stride;
    int bw;            // blocks per row
    int aligned8;      // out and stride are multiples of 8: whole 8-byte row stores
    const struct DctqConsts *consts; // constants of the stream's quality (quality 50 on the scaled_dct branch)
    int scaled;        // decode()'s scaled_dct branch (codec.py:59-62)
    double pow2;       // 2 ** (quality field of the stream) on that branch
    uint32_t head[4];  // the 16-byte header h, w, stride and the constants above were derived from: a workgroup of the fused kernel whose
                       // stream starts with other bytes writes NO pixel (a launch on a guessed header must not touch memory outside the
                       // real image - a caller decoding into a window of a larger surface keeps its neighbours)
};

// One stream of a batch (entropy_decode_idct_gpu_batch): where its words, ranges, tiles, blocks and workgroups sit in the batch's arrays.
// The stream starts at word `word0` of the batch's stream buffer (every stream 4-byte aligned there); the same margin-free run as
// entropy_decode_idct_gpu with margin_bits = 0 (fast_end = stream_bits).
struct DecFrame {
    uint32_t word0, nwords, last_mask;  // the stream's words
    uint32_t fast_end, stream_bits;
    uint32_t nranges, range0;           // its ranges; index of its first range among the batch's (traces: range0 * cap entries in)
    uint32_t tile0, ntiles;             // its waves in the measure kernel's grid
    uint32_t blk0, nblocks;             // its blocks; index of its first block among the batch's (positions)
    uint32_t wg0, nwgs;                 // its workgroups in the fused kernel's grid
    uint32_t pad_;
    DecIdctArgs idct;                   // where its pixels go, its geometry, constants and header
};

size_t entropy_decode_gpu_work_bytes(size_t stream_bytes, size_t nblocks);
// d_stream_words: the whole stream (header included) in device memory, 4-byte aligned; the 4-byte word that holds its last byte is
// read whole (the bytes behind the stream's end are masked off), nothing behind that word is touched.  range_bits: stream bits per lane, a value entropy_decode_gpu_range_ok() accepts - an odd
// number of 32-bit words from 288 to 2,016 bits (shorter = shorter chains = faster, but every range must
// hold a block start of the true chain: giveup has bit 4 set when one did not - try a longer range).  d_work: zeroed when it was allocated;
// `epoch`: a number never used before on this workspace (the single-launch scans recognise this call's words by it), != 0.
// margin_bits: 2048 = blocks that start in the stream's last 2,048 bits are left to the caller (the host decoder's rule: whatever the
// stream holds, no block of the chain reaches its end); 0 = the chain runs to the end, m = all blocks of a whole stream, and giveup bit 256
// says that a block reached behind the end (a cut stream: call again with 2048).
// flat_grid: launches of up to this many workgroups add up all sums in front of a workgroup; larger ones use inclusive sums (wave_lookback).
// Writes the PIXELS of blocks [0, m) through `idct`; *d_status (zeroed by the caller) says how many that is and where the stream
// and the running DC stand behind them.  Asynchronous on `stream`; *d_status is complete when the stream has drained.
size_t entropy_decode_gpu_desc_words(size_t stream_bytes, size_t nblocks);
bool entropy_decode_gpu_range_ok(int range_bits);
hipError_t entropy_decode_idct_gpu(const void *d_stream_words, size_t stream_bytes, size_t nblocks, const DecLutsDev *d_luts, void *d_work,
                                   size_t work_bytes, unsigned long long *d_desc, size_t desc_words, uint32_t epoch, const DecIdctArgs &idct,
                                   DecStatus *d_status, int range_bits, int margin_bits, hipStream_t stream, int flat_grid = 4096);

// The batch form: `nframes` whole streams in two launches.  d_frames / d_tile_frame / d_wg_frame: device copies of the descriptors, the frame of
// every wave of the measure grid and of every workgroup of the fused grid; d_status: nframes entries, zeroed by the caller; the sums'
// look-back words and the epoch as above; small_win: every stream has at most 240 bits per block on average.  A frame whose status comes back
// with giveup != 0 or m != its block count is the caller's to decode again on its own.
size_t entropy_decode_batch_work_bytes(size_t total_ranges_288, size_t total_blocks, size_t nframes);
uint32_t entropy_decode_batch_tiles(uint32_t nranges, int range_bits);
uint32_t entropy_decode_batch_wgs(size_t nblocks);
uint32_t entropy_decode_batch_ranges(size_t stream_bytes, int range_bits);
hipError_t entropy_decode_idct_gpu_batch(const void *d_words_all, const DecFrame *d_frames, const uint32_t *d_tile_frame, const uint32_t *d_wg_frame, uint32_t nframes,
                                         uint32_t total_tiles, uint32_t total_wgs, uint32_t total_ranges, size_t total_blocks, bool small_win, const DecLutsDev *d_luts, void *d_work,
                                         size_t work_bytes, unsigned long long *d_desc, size_t desc_words, uint32_t epoch, DecStatus *d_status, int range_bits, hipStream_t stream,
                                         int flat_grid = 4096);

} // namespace tic
