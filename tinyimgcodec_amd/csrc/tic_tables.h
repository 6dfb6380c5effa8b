// tic_tables.h - codec constants shared by the device kernels and the host entropy stage.
//
// Values restate the reference's tables: constants.py:9-20 (JPEG Annex K luminance quantisation table),
// constants.py:23-34 (zig-zag scan), constants.py:53-242 (Annex K.3 luminance Huffman tables, held here as the
// standard's BITS/HUFFVAL lists and expanded canonically at start-up).
#pragma once
#include <stdint.h>

namespace tic {

static constexpr int kQTable[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

// kZigzag[k] = natural index u*8+v of scan position k.
static constexpr uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                        12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                        58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static constexpr uint8_t kDcBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static constexpr uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static constexpr uint8_t kAcBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static constexpr uint8_t kAcVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// ---- float64 constants of scipy.fftpack's (pocketfft) N=8 DCT-II/III, SURVEY.md Appendix A -----------------
static constexpr double kWR = 0x1.6a09e667f3bccp-1;  // radix-2 twiddle real part
static constexpr double kWI = 0x1.6a09e667f3bcdp-1;  // radix-2 twiddle imag part (1 ulp above kWR)
static constexpr double kTW0 = 0x1.f6297cff75cb0p-1; // cos(1*pi/16)
static constexpr double kTW1 = 0x1.d906bcf328d46p-1;
static constexpr double kTW2 = 0x1.a9b66290ea1a3p-1;
static constexpr double kTW3 = 0x1.6a09e667f3bccp-1; // cos(4*pi/16) as pocketfft computes it
static constexpr double kTW4 = 0x1.1c73b39ae68c8p-1;
static constexpr double kTW5 = 0x1.87de2a6aea963p-2;
static constexpr double kTW6 = 0x1.8f8b83c69a60ap-3;
static constexpr double kSqrt2 = 0x1.6a09e667f3bcdp+0;
static constexpr double kSq2h = 0x1.6a09e667f3bcdp-1; // sqrt2 * 0.5 (exact scaling)

} // namespace tic
