// tic_entropy_gpu.h - device-side entropy stage (see tic_entropy_gpu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tic {

// Huffman tables as the kernels use them: the codeword already shifted left by the size category (the value bits are
// ORed in below it) and the symbol's total length.  Index (run << 4) | size for AC, size for DC; bits == 0: no code.
struct HuffDev {
    uint32_t ac_sym[256], ac_bits[256];
    uint32_t dc_sym[16], dc_bits[16];
    // the same in one word per symbol (lane-per-block packing kernel): (total bits << 27) | (codeword << size); 0: no code
    uint32_t ac_pack[256], dc_pack[16];
};
void build_huff_dev(HuffDev *t);

// The device entropy stage (tic_entropy_gpu.hip): pack -> (tile sums) -> place.  d_work: entropy_fused_work_bytes() bytes,
// never initialised by the caller.  Frame f's stream (16-byte header + payload) starts at d_out + f * out_frame_stride and
// may hold cap_words payload words (the stream area need not be zeroed; nothing past it is written); d_lens[f] (may be
// null) receives its length in bytes.  d_status (may be null, may be host-mapped; single-frame calls) receives {payload bits
// of frame 0, error}.  *d_err: 1 = a coefficient without a Huffman code, 2 = a stream does not fit; it must be zero on
// entry, and the placing kernel zeroes *d_err_next, the flag of the next call (two flags used in turn need no memset
// between calls).
// mode: kEntropyLanePerBlock - the packing kernel with a lane per block (a wave = 64 blocks, up to 512 bits per block);
//       *d_err = 4 when a block needs more: the caller runs the stage again with kEntropyEightLanes, the packing kernel with
//       8 lanes per block, which has no such limit (any block the format allows).  Both feed the same placing kernel.
enum { kEntropyLanePerBlock = 0, kEntropyEightLanes = 1 };
size_t entropy_fused_work_bytes(size_t nblocks_total);
hipError_t entropy_gpu_fused(const int16_t *d_zz, size_t blocks_per_frame, int nframes, const HuffDev *d_tab, void *d_work,
                             size_t work_bytes, void *d_out, size_t out_frame_stride, size_t cap_words, int h, int w, int quality,
                             unsigned long long *d_lens, unsigned long long *d_status, int *d_err, int *d_err_next, int mode,
                             hipStream_t stream, hipStream_t place_stream = nullptr, hipEvent_t pack_done = nullptr);
// place_stream (with pack_done, an event of the caller's): the placing kernel - the stage's only writer of d_out, d_lens, d_status and
// *d_err_next - is queued on place_stream behind the packing on `stream`; the packing of the NEXT frame may then run on `stream` beside it.

} // namespace tic
