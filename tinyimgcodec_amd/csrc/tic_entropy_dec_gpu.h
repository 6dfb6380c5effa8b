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
};

struct DecStatus {
    int giveup;                 // != 0: something unusual on the true chain - the caller decodes the whole stream on the host
    int dc_out;                 // running DC behind the last block produced here
    unsigned long long m;       // blocks produced: [0, m) of the int16 [N][64] array are complete (entry 0 = integrated DC)
    unsigned long long pos_out; // first stream bit behind block m - 1
};

size_t entropy_decode_gpu_work_bytes(size_t stream_bytes, size_t nblocks);
// d_stream_words: the whole stream (header included) in device memory, 4-byte aligned, readable for 8 bytes past its end.
// range_bits: 512, 1024 or 2048 stream bits per lane (shorter = more lanes = faster, but every range must hold a block start of
// the true chain: giveup has bit 4 set when one did not - try 2048).  Asynchronous on `stream`; *d_status is complete when the
// stream has drained.
hipError_t entropy_decode_gpu(const void *d_stream_words, size_t stream_bytes, size_t nblocks, const DecLutsDev *d_luts, void *d_work,
                              size_t work_bytes, int16_t *d_zz, DecStatus *d_status, int range_bits, hipStream_t stream);

} // namespace tic
