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
// nbits must hold nblocks entries, bitoff nblocks entries (exclusive scan of nbits, in bits).
hipError_t entropy_gpu_count(const int16_t *d_zz, size_t nblocks, const HuffDev *d_tab, uint32_t *d_nbits,
                             unsigned long long *d_bitoff, void *d_temp, size_t temp_bytes, int *d_err, hipStream_t stream);
// payload words must be zero on entry (whole 32-bit words covering the payload bits).
hipError_t entropy_gpu_emit(const int16_t *d_zz, size_t nblocks, const HuffDev *d_tab, const unsigned long long *d_bitoff,
                            uint32_t *d_payload_words, int *d_err, hipStream_t stream);

} // namespace tic
