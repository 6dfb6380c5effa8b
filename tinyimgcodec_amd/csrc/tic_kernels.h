// tic_kernels.h - launch interface between the C-ABI layer (tic_api.hip) and the kernels (tic_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tic_math.h"

namespace tic {

struct DctqArgs {
    const uint8_t *img;  // device, uint8 [h][stride]
    int h, w;
    long stride;         // bytes between rows
    int bw;              // blocks per row  = ceil(w/8)
    int tiles_x;         // 8-block strips per block row = ceil(bw/8)
    int ntiles;          // bh * tiles_x
    int aligned8;        // img and stride are multiples of 8 -> 8-byte row loads
    const DctqConsts *consts; // device
    int16_t *out;        // device, int16 [N][64] zig-zag
    unsigned long long *fallback_count; // device counter of blocks redone on the exact path (may be null)
    // persistent-wave schedule of the hybrid kernel: wave g handles strips g, g + nwaves, g + 2*nwaves, ...
    int nwaves;          // waves in the grid
    // generalised walk: workgroup w starts at strip w*wg_stride (+ wave index), a wave advances by tstep strips and
    // stops at min(nfast, w*wg_stride + wg_span).  Strided schedule: wg_stride 4, tstep nwaves, wg_span = all;
    // chunked schedule (large frames): wg_stride = wg_span = 4*S, tstep 4 - a workgroup streams S*4 adjacent strips
    int wg_stride, tstep, wg_span;
    // round-interleaved schedule (round_wgs > 0): workgroups [r*round_wgs, (r+1)*round_wgs) share round r, a dense range
    // of round_wgs*wg_span strips which they walk together with stride round_wgs*4 (wg_stride = 4 inside the round)
    int round_wgs;
    // team schedule (team_count > 0; grids that fit the chip at once): workgroups w, w + team_count, w + 2*team_count ...
    // form a team (in practice: the workgroups resident on one CU, dispatched one round after the other).  The frame is
    // cut into rows of team_count*4 strips; round r of every team takes rows [split[r], split[r+1]) - later rounds start
    // later (their prologues compete with running waves) and get fewer rows, so that all waves end together.
    int team_count;
    int split[9];
    int step_ty, step_tx; // nwaves / fast_tx and nwaves % fast_tx (strip coordinates advance without a division)
    // the hybrid kernel covers the rectangle of complete, 8-byte aligned strips [0,fast_ty) x [0,fast_tx);
    // the exact kernel (rem_mode = 1) covers the rest: right-hand partial strips and the bottom partial block row
    int fast_ty, fast_tx;
    int rem_mode;
    // the hybrid kernel's strip walk, precomputed on the host so the loop needs only 32-bit scalar adds: advancing by
    // tstep strips adds in_step32 bytes to the pixel offset and oblk_step to the block index; when the strip column
    // wraps past fast_tx the *_wrap terms are added as well (frames handled by this kernel are < 4 GiB)
    uint32_t in_step32, in_wrap32, oblk_step, oblk_wrap;
    // batch of equally sized frames in one launch: blockIdx.y = frame; byte strides between frames
    int nframes;
    long frame_stride_in, frame_stride_out;
    unsigned long long *dbg; // diagnostic builds only: per-wave s_memtime stamps (8 per wave), else null
    // strip kernel (round 2): division-free prologue.  q = mulhi(n, magic) with magic = floor(2^32 / d) + 1 is n / d for
    // n * d < 2^32 (checked by the launcher).  The team schedule runs on a 2-D grid (x = team, y = round) and takes the
    // rows of round r from byte r of split_lo (byte 8: split_hi).
    uint32_t magic_fast_tx, magic_tstep;
    unsigned long long split_lo, split_hi;
};

struct WideArgs {          // dctq_exact_wide_kernel: integer images outside 0..255
    const int32_t *img;    // device, int32 [h][stride]
    int32_t *out;          // device, int32 [N][64] zig-zag
    int h, w;
    long stride;           // elements between rows
    int bw, tiles_x, ntiles;
    const DctqConsts *consts;
};

struct IdctArgs {
    const int16_t *coeffs; // device, int16 [N][64] zig-zag, DC integrated
    uint8_t *out;          // device, uint8 [h][stride]
    int h, w;
    long stride;
    int bw, tiles_x, ntiles;
    int aligned8;
    const DctqConsts *consts;
    int scaled = 0;    // decode()'s scaled_dct branch (codec.py:59-62): coefficients / ANNSCALES * pow2, consts = quality 50
    double pow2 = 1.0; // 2 ** (quality field of the stream)
    // Block-range form (first_block >= 0): the kernel transforms blocks [first_block, first_block + nblocks_sel) in raster order,
    // 8 per wave, instead of the frame's strips - the few blocks the host decodes at the end of a long stream (their coefficients
    // sit at their places in `coeffs`; every other block's pixels come from the fused decode kernel).
    long first_block = -1;
    long nblocks_sel = 0;
};

// C-ABI kernel selector (TIC_KERNEL_AUTO / _EXACT / _HYBRID) -> launch_dctq's variant (1 exact, 2 strip kernel), -1 for anything else.
int dctq_kernel_id(int abi_variant);
// ev_start / ev_stop (both or neither): bound to the strip kernel's own dispatch packet (hipExtLaunchKernelGGL) - its start and end
// time stamps, without a marker packet in the queue; ignored for the exact kernel and for banded frames
hipError_t launch_dctq(DctqArgs a, int variant, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_dctq_wide(const WideArgs &a, hipStream_t stream);
hipError_t launch_idct(const IdctArgs &a, hipStream_t stream);
hipError_t launch_selftest_transpose(const void *in, void *out_dpp, void *out_ref, int nthreads, hipStream_t stream);

} // namespace tic
