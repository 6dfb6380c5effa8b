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
// What the device decoder (tic_entropy_dec_gpu.hip) leaves to the host: blocks [first_block, N) from read position pos_bits with
// running DC running_dc, into zz_tail (N - first_block blocks); and the decoder's look-up tables for the device.
int entropy_decode_tail(const uint8_t *data, size_t len, int h, int w, size_t first_block, size_t pos_bits, int running_dc, int16_t *zz_tail);
void dec_luts_fill(uint16_t *dc11, uint16_t *ac11, uint16_t *ac16);
// the device decoder's chain tables (DecLutsDev::mdc / mac / mlong, tic_entropy_dec_gpu.h)
void dec_chain_luts_fill(uint8_t *mdc /*[2048]*/, uint8_t *mac /*[4096]*/, uint8_t *mlong /*[256]*/);
void dec_pair_luts_fill(uint32_t *ac2 /*[2048]*/, uint32_t *long32 /*[192]*/);
} // namespace tic
