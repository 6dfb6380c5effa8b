// tic_hooks.h - the one gate in front of every environment variable that changes which code path the library takes.
//
// The SHIPPED library (libtinyimgcodec_hip.so) never consults the environment: test_hook() is a constant nullptr there and every
// branch behind it folds away.  The same sources compiled with -DTIC_TEST_HOOKS (libtinyimgcodec_hip_hooks.so, built beside the
// product by csrc/Makefile; tinyimgcodec_amd/_native.py loads it when TIC_TEST_HOOKS=1 is set, which tests/conftest.py and the
// measurement scripts under tools/ do) read
//   TIC_ENT_DIRECT_GROUPS   device entropy stage: group count above which stream offsets are summed in two levels
//   TIC_DECODE_SERIAL       Huffman decoder: always the host's serial decoder
//   TIC_DECODE_RANGE        device Huffman decoder: stream bits per lane (an odd number of 32-bit words, 288 ... 2016) instead of the choice by block length
//   TIC_DECODE_RULE         device Huffman decoder: "<average blocks per range>,<least words per range>" instead of 2,9 (measurements of the rule)
//   TIC_DECODE_SHADOWS      device Huffman decoder: ranges in front of its own that a wave of the measure kernel shadows (1 ... 16; default 4, 8 below 544 bits per range)
//   TIC_DECODE_ROUNDS       device Huffman decoder: hand-over rounds of the stitch (default 8)
//   TIC_DECODE_NO_HOSTPIX   tic_decompress of small images through the device image buffer and a DMA copy, as large ones (not through host-mapped memory)
//   TIC_BATCH_CHUNK         tic_compress_batch: frames per chunk instead of the choice by frame size
//   TIC_DECODE_NO_GUESS     tic_decompress_dev always reads the header first (no launch on a guess of it)
//   TIC_NO_SMALL_PATH       tic_compress of small frames through the device stream buffer and a DMA copy, as large ones (not through host-mapped memory)
//   TIC_DECODE_MIN_BLOCKS, TIC_DECODE_MIN_BITS, TIC_DECODE_MIN_DENSITY   the shortest stream the device Huffman decoder takes (defaults 1024 blocks, 8192 bits, any density)
//   TIC_DECODE_FLAT_GRID    device Huffman decoder: launches of up to this many workgroups sum all words in front (default 4096; 0: inclusive sums always)
//   TIC_DECODE_MARGIN       device Huffman decoder: first run with the 2,048-bit margin and the host's tail (rounds 2-3's only mode; now the second run's)
//   TIC_DECODE_TRACE        device Huffman decoder: one line per run on stderr (range, margin, give-up bits, blocks produced)
//   TIC_DECODE_HOST         Huffman decoder: never the device decoder (host parallel / serial as the stream's length says)
//   TIC_DECODE_THREADS      host Huffman decoder: threads of the parallel decoder
//   TIC_COMM_FORCE_RCCL     a single rank goes through RCCL too (the only way to exercise tic_comm.hip on a one-GPU box)
//   TIC_TUNE, TIC_SPLIT, TIC_SCHED, TIC_CHUNK, TIC_MAX_WGS   schedule knobs of the strip kernel's launcher
//   TIC_ORDER               pass order of the strip kernel (0 rows first, 1 columns first) instead of the choice by grid
//   TIC_BAND_BYTES          size from which a frame is transformed in bands of block rows (4 GiB in production)
// at every call (tests flip them inside one process).  tic_build_has_test_hooks() tells which build a process has loaded;
// tests/test_gpu_parity.py::test_shipped_library_in_a_fresh_process runs the parity core on the product build.
#pragma once
#include <stdlib.h>

namespace tic {
#ifdef TIC_TEST_HOOKS
inline bool test_hooks_enabled() { return true; }
inline const char *test_hook(const char *name) { return getenv(name); }
#else
inline constexpr bool test_hooks_enabled() { return false; }
inline constexpr const char *test_hook(const char *) { return nullptr; }
#endif
} // namespace tic
