// tic_entropy_gpu.h - device-side entropy stage (see tic_entropy_gpu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tic {

struct HuffDev {
    uint32_t ac[256]; // (codeword << 8) | length, index (run << 4) | size; 0 = no code
    uint32_t dc[16];  // index = size category
};
void build_huff_dev(HuffDev *t);

size_t entropy_gpu_scan_temp_bytes(size_t nblocks);
// nblocks = blocks of all frames (blocks_per_frame each; DPCM restarts at every frame).  nbits / bitoff hold
// nblocks entries; bitoff = exclusive scan of nbits over the whole batch, in bits.  d_lanebits (8 bytes per block) receives
// the bits of each of the 8 lanes of a block for the emit kernel.
hipError_t entropy_gpu_count(const int16_t *d_zz, size_t nblocks, size_t blocks_per_frame, const HuffDev *d_tab,
                             uint32_t *d_nbits, uint8_t *d_lanebits, unsigned long long *d_bitoff, void *d_temp,
                             size_t temp_bytes, int *d_err, hipStream_t stream);
// d_payload_words = first payload word of frame 0 (16 bytes after its buffer start); frame f's buffer starts
// out_frame_stride bytes further and holds cap_words payload words.  The payload words must be zero on entry.
// *d_err becomes 2 if a frame's payload does not fit.
hipError_t entropy_gpu_emit(const int16_t *d_zz, size_t nblocks, size_t blocks_per_frame, const HuffDev *d_tab,
                            const unsigned long long *d_bitoff, const uint8_t *d_lanebits, uint32_t *d_payload_words,
                            size_t out_frame_stride, size_t cap_words, int *d_err, hipStream_t stream);
// One frame, no host round trip: publishes the payload size in bits (d_nbits/d_bitoff of entropy_gpu_count), writes
// the 16-byte header in front of the payload and zeroes the payload words (at most cap_words; *d_err becomes 2 when
// the payload needs more).  cap_words*4 must be a
// multiple of 16 or the buffer must extend to the next 16-byte boundary.
hipError_t entropy_gpu_zero_payload(const uint32_t *d_nbits, const unsigned long long *d_bitoff, size_t nblocks,
                                    uint32_t *d_payload_words, size_t cap_words, unsigned long long *d_total_bits, int *d_err,
                                    int h, int w, int quality, hipStream_t stream);
// Writes each frame's header at the start of its buffer and its stream length (bytes) into d_lens[f].
hipError_t entropy_gpu_finish_frames(const uint32_t *d_nbits, const unsigned long long *d_bitoff, size_t blocks_per_frame,
                                     int nframes, int h, int w, int quality, void *d_out, size_t out_frame_stride,
                                     unsigned long long *d_lens, hipStream_t stream);

// Round 2: the whole stage in one pass + a finishing kernel (tic_entropy_gpu.hip).  d_work: entropy_fused_work_bytes() bytes,
// zeroed once when allocated; `parity` must alternate between consecutive calls on the same workspace (the finishing kernel
// re-arms the descriptor array the next call uses).  Frame f's stream (16-byte header + payload) starts at d_out +
// f * out_frame_stride and may hold cap_words payload words; d_lens[f] (may be null) receives its length in bytes;
// d_status[0] (may be null) the payload bits of frame 0.  *d_err: 1 = a coefficient without a Huffman code, 2 = a stream does
// not fit, 3 = internal (look-back gave up); it must be zero on entry, and the finishing kernel zeroes *d_err_next (the flag of the
// next call: two flags used in turn need no memset between calls).  The stream area need not be zeroed.
size_t entropy_fused_work_bytes(size_t nblocks_total);
hipError_t entropy_gpu_fused(const int16_t *d_zz, size_t blocks_per_frame, int nframes, const HuffDev *d_tab, void *d_work,
                             size_t work_bytes, int parity, void *d_out, size_t out_frame_stride, size_t cap_words, int h, int w,
                             int quality, unsigned long long *d_lens, unsigned long long *d_status, int *d_err, int *d_err_next,
                             hipStream_t stream);

} // namespace tic
