// tic_entropy.h - host entropy stage (see tic_entropy.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tic {
size_t num_blocks(int h, int w);
size_t compress_bound(int h, int w);
void write_header(uint8_t *out, int h, int w, int quality);
int entropy_encode(const int16_t *zz, int h, int w, int quality, uint8_t *out, size_t cap, size_t *out_len);
int parse_header(const uint8_t *data, size_t len, int *h, int *w, int *quality, uint32_t *flag);
// Huffman + run-length decode into int16 [N][64] zig-zag with the DC already integrated (np.cumsum).
int entropy_decode(const uint8_t *data, size_t len, int h, int w, int16_t *zz);
} // namespace tic
