/*
 * tic_oracle.c - CPU restatement of the tinyimgcodec Python codec (TEST INFRASTRUCTURE, see tic_oracle.h).
 *
 * Plain scalar C, one function per reference function, written for clarity not speed.  Must be compiled
 * with -ffp-contract=off -fno-fast-math (the Makefile does): every + - * / below is meant to be exactly one
 * IEEE-754 binary64 operation, in the order scipy's pocketfft performs them.
 *
 * file:line citations are into /root/reference/.
 */
#include "tic_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------
 * Tables.  constants.py:9-20 (JPEG Annex K luminance quantisation table), constants.py:23-34 (zig-zag).
 * ---------------------------------------------------------------------------------------------------- */
static const int32_t QTABLE[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
    69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
    81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

/* constants.py:37-51: ANNSCALES = <this integer table> / 2048, the scale the reference's C encoder leaves in its coefficients
 * (8 * a_u * a_v of the AAN factorisation in 14-bit fixed point); used by decode()'s scaled_dct branch only. */
static const int32_t ANNSCALES_INT[64] = {
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
    21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,
    8867,  12299, 11585, 10426, 8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

/* ZIGZAG[k] = natural index (u*8+v) of the k-th coefficient in scan order. */
static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* constants.py:53-242 is the JPEG Annex K.3 luminance Huffman table pair written out as bit strings.  Here it
 * is rebuilt from the standard's BITS/HUFFVAL lists by the canonical code assignment (ITU-T T.81 Annex C);
 * tests check a digest of the resulting (symbol -> codeword) map against the reference's dictionaries. */
static const uint8_t DC_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint16_t code[256];
    uint8_t len[256]; /* 0 = symbol has no code (reference: KeyError) */
} hufftab;

static hufftab HT_DC, HT_AC;
static int tables_ready = 0;

static void build_table(const uint8_t bits[16], const uint8_t *vals, hufftab *t) {
    memset(t, 0, sizeof(*t));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++) {
            t->code[vals[k]] = (uint16_t)code;
            t->len[vals[k]] = (uint8_t)l;
            code++;
            k++;
        }
        code <<= 1;
    }
}

static void ensure_tables(void) {
    if (!tables_ready) {
        build_table(DC_BITS, DC_VALS, &HT_DC);
        build_table(AC_BITS, AC_VALS, &HT_AC);
        tables_ready = 1;
    }
}

/* Exposed for the table-digest test: writes "D|A,run,size,bitstring\n" lines; returns length. */
size_t tico_dump_tables(char *out, size_t cap) {
    ensure_tables();
    size_t n = 0;
    for (int pass = 0; pass < 2; pass++) {
        const hufftab *t = pass ? &HT_AC : &HT_DC;
        for (int s = 0; s < 256; s++) {
            if (!t->len[s]) continue;
            char line[64];
            int m = 0;
            line[m++] = pass ? 'A' : 'D';
            line[m++] = ',';
            int run = pass ? (s >> 4) : 0, size = pass ? (s & 15) : s;
            m += sprintf(line + m, "%d,%d,", run, size);
            for (int b = t->len[s] - 1; b >= 0; b--) line[m++] = ((t->code[s] >> b) & 1) ? '1' : '0';
            line[m++] = '\n';
            if (n + (size_t)m > cap) return 0;
            memcpy(out + n, line, (size_t)m);
            n += (size_t)m;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------------
 * scipy.fftpack.dct(x, norm="ortho"), N = 8: pocketfft T_dcst23<double>::exec, type 2, via the backward real
 * FFT radb2(ido=4,l1=1) -> radb4(ido=1,l1=2).  Operation order per SURVEY.md Appendix A.  Called from
 * utils.py:32-37.  (pocketfft is a scipy dependency, not under /root/reference; scipy 1.15.3 in the build
 * container - pinned by tests/golden/dct_blocks.npz.)
 * ---------------------------------------------------------------------------------------------------- */
static const double W_R = 0x1.6a09e667f3bccp-1; /* radix-2 twiddle, real part  (sin(pi/4) as pocketfft computes it) */
static const double W_I = 0x1.6a09e667f3bcdp-1; /* radix-2 twiddle, imag part  (cos(pi/4))                           */
static const double TW[7] = {
    0x1.f6297cff75cb0p-1, /* cos(1*pi/16) */
    0x1.d906bcf328d46p-1, /* cos(2*pi/16) */
    0x1.a9b66290ea1a3p-1, /* cos(3*pi/16) */
    0x1.6a09e667f3bccp-1, /* cos(4*pi/16), computed via sin -> ...bcc */
    0x1.1c73b39ae68c8p-1, /* cos(5*pi/16) */
    0x1.87de2a6aea963p-2, /* cos(6*pi/16) */
    0x1.8f8b83c69a60ap-3, /* cos(7*pi/16) */
};
static const double SQRT2 = 0x1.6a09e667f3bcdp+0; /* (double)1.41421356237309504880L */
static const double FCT = 0.25;                    /* 1/sqrt(2N) */

void tico_dct8(double c[8]) {
    double h[8], o[8];
    /* T_dcst23 type 2 pre-processing */
    c[0] = c[0] * 2.0;
    c[7] = c[7] * 2.0;
    for (int k = 1; k < 7; k += 2) { /* MPINPLACE(c[k+1], c[k]) */
        double t = c[k + 1];
        c[k + 1] = t - c[k];
        c[k] = t + c[k];
    }
    /* radb2, ido=4, l1=1 */
    h[0] = c[0] + c[7];
    h[4] = c[0] - c[7];
    h[3] = 2.0 * c[3];
    h[7] = -2.0 * c[4];
    {
        double tr2, ti2;
        h[1] = c[1] + c[5];
        tr2 = c[1] - c[5];
        ti2 = c[2] + c[6];
        h[2] = c[2] - c[6];
        double a = W_R * ti2, b = W_I * tr2;
        h[6] = a + b;
        double d = W_R * tr2, e = W_I * ti2;
        h[5] = d - e;
    }
    /* radb4, ido=1, l1=2 */
    for (int k = 0; k < 2; k++) {
        double tr2 = h[4 * k] + h[4 * k + 3];
        double tr1 = h[4 * k] - h[4 * k + 3];
        double tr3 = 2.0 * h[4 * k + 1];
        double tr4 = 2.0 * h[4 * k + 2];
        o[k] = tr2 + tr3;
        o[k + 4] = tr2 - tr3;
        o[k + 6] = tr1 + tr4;
        o[k + 2] = tr1 - tr4;
    }
    for (int i = 0; i < 8; i++) c[i] = o[i] * FCT;
    /* T_dcst23 type 2 post-processing */
    for (int k = 1, kc = 7; k < 4; k++, kc--) {
        double p1 = TW[k - 1] * c[kc], p2 = TW[kc - 1] * c[k];
        double t1 = p1 + p2;
        double p3 = TW[k - 1] * c[k], p4 = TW[kc - 1] * c[kc];
        double t2 = p3 - p4;
        c[k] = 0.5 * (t1 + t2);
        c[kc] = 0.5 * (t1 - t2);
    }
    c[4] = c[4] * TW[3];
    c[0] = c[0] * (SQRT2 * 0.5);
}

/* scipy.fftpack.idct(x, norm="ortho") = DCT-III: pocketfft T_dcst23 type 3 via the forward real FFT
 * radf4(ido=1,l1=2) -> radf2(ido=4,l1=1).  Called from utils.py:40-45. */
void tico_idct8(double c[8]) {
    double a[8], r[8];
    c[0] = c[0] * SQRT2;
    for (int k = 1, kc = 7; k < 4; k++, kc--) {
        double t1 = c[k] + c[kc], t2 = c[k] - c[kc];
        double p1 = TW[k - 1] * t2, p2 = TW[kc - 1] * t1;
        double p3 = TW[k - 1] * t1, p4 = TW[kc - 1] * t2;
        c[k] = p1 + p2;
        c[kc] = p3 - p4;
    }
    c[4] = c[4] * (2.0 * TW[3]);
    /* radf4, ido=1, l1=2: cc[b + 2c] -> ch[b + 4k] */
    for (int k = 0; k < 2; k++) {
        double tr1 = c[k + 6] + c[k + 2];
        a[4 * k + 2] = c[k + 6] - c[k + 2];
        double tr2 = c[k] + c[k + 4];
        a[4 * k + 1] = c[k] - c[k + 4];
        a[4 * k] = tr2 + tr1;
        a[4 * k + 3] = tr2 - tr1;
    }
    /* radf2, ido=4, l1=1 */
    r[0] = a[0] + a[4];
    r[7] = a[0] - a[4];
    r[4] = -a[7];
    r[3] = a[3];
    {
        double m1 = W_R * a[5], m2 = W_I * a[6];
        double tr2 = m1 + m2;
        double m3 = W_R * a[6], m4 = W_I * a[5];
        double ti2 = m3 - m4;
        r[1] = a[1] + tr2;
        r[5] = a[1] - tr2;
        r[2] = ti2 + a[2];
        r[6] = ti2 - a[2];
    }
    for (int i = 0; i < 8; i++) c[i] = r[i] * FCT;
    for (int k = 1; k < 7; k += 2) { /* MPINPLACE(c[k], c[k+1]) */
        double t = c[k];
        c[k] = t - c[k + 1];
        c[k + 1] = t + c[k + 1];
    }
}

/* utils.py:32-37: dct along axis -2 (for every column, over the row index) and then along axis -1. */
void tico_block_dct(const int32_t in[64], double out[64]) {
    double col[8];
    for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 8; i++) col[i] = (double)in[i * 8 + j];
        tico_dct8(col);
        for (int i = 0; i < 8; i++) out[i * 8 + j] = col[i];
    }
    for (int i = 0; i < 8; i++) tico_dct8(out + i * 8);
}

/* utils.py:40-45 */
void tico_block_idct(const double in[64], double out[64]) {
    double col[8];
    for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 8; i++) col[i] = in[i * 8 + j];
        tico_idct8(col);
        for (int i = 0; i < 8; i++) out[i * 8 + j] = col[i];
    }
    for (int i = 0; i < 8; i++) tico_idct8(out + i * 8);
}

/* utils.py:50-53: factor = 5000/q (float) if q < 50 else 200 - 2q (int); divisor = (Q*factor)/100. */
int tico_divisors(int quality, double div[64]) {
    if (quality < 1 || quality > 99) return TICO_E_QUALITY;
    if (quality < 50) {
        double factor = 5000.0 / (double)quality;
        for (int i = 0; i < 64; i++) {
            double p = (double)QTABLE[i] * factor;
            div[i] = p / 100.0;
        }
    } else {
        int factor = 200 - 2 * quality;
        for (int i = 0; i < 64; i++) div[i] = (double)(QTABLE[i] * factor) / 100.0;
    }
    return TICO_OK;
}
/* The same for a quality that is not an integer (utils.py:50-53 computes with whatever number it is given: a float quality makes
 * `200 - 2 * quality` a float too, and `table * factor / 100` is evaluated left to right in float64). */
int tico_divisors_f(double quality, double div[64]) {
    if (!(quality >= 1.0 && quality <= 99.0)) return TICO_E_QUALITY;
    double factor = quality < 50.0 ? 5000.0 / quality : 200.0 - 2.0 * quality;
    for (int i = 0; i < 64; i++) {
        double p = (double)QTABLE[i] * factor;
        div[i] = p / 100.0;
    }
    return TICO_OK;
}

/* utils.py:56-61: np.pad(..., "reflect") index map for right/bottom padding (edge sample not repeated;
 * a length-1 axis degenerates to edge replication, as numpy does). */
static int reflect_index(int i, int n) {
    if (n == 1) return 0;
    int p = 2 * (n - 1);
    int j = i % p;
    return j < n ? j : p - j;
}

/* Transform stage up to and including zig-zag; zz: int32 [N][64], zz[.][0] = un-differenced DC.
 * Pixels come as uint8 (image) or, for integer images outside 0..255, as int32 (image32; codec.py:29 casts with astype(int32)
 * and transforms whatever it finds); stride in elements. */
static int transform_zz_div(const uint8_t *image, const int32_t *image32, int h, int w, ptrdiff_t stride, const double div[64], int32_t *zz);
static int transform_zz_any(const uint8_t *image, const int32_t *image32, int h, int w, ptrdiff_t stride, int quality, int32_t *zz) {
    double div[64];
    int rc = tico_divisors(quality, div);
    if (rc) return rc;
    return transform_zz_div(image, image32, h, w, stride, div, zz);
}
static int transform_zz_div(const uint8_t *image, const int32_t *image32, int h, int w, ptrdiff_t stride, const double div[64], int32_t *zz) {
    int bh = (h + 7) / 8, bw = (w + 7) / 8;
    if (h == 0 || w == 0) return TICO_OK;
    for (int by = 0; by < bh; by++) {
        for (int bx = 0; bx < bw; bx++) {
            int32_t px[64];
            double X[64];
            for (int i = 0; i < 8; i++) {
                int y = reflect_index(by * 8 + i, h);
                for (int j = 0; j < 8; j++) {
                    int x = reflect_index(bx * 8 + j, w);
                    px[i * 8 + j] = (image ? (int32_t)image[(ptrdiff_t)y * stride + x] : image32[(ptrdiff_t)y * stride + x]) - 128; /* codec.py:29 */
                }
            }
            tico_block_dct(px, X);
            int32_t q[64];
            for (int i = 0; i < 64; i++) {
                double t = X[i] / div[i];
                q[i] = (int32_t)nearbyint(t); /* np.round: half to even (default FE_TONEAREST) */
            }
            int32_t *o = zz + ((size_t)by * bw + bx) * 64;
            for (int k = 0; k < 64; k++) o[k] = q[ZIGZAG[k]]; /* codec.py:32-33 */
        }
    }
    return TICO_OK;
}

static int transform_zz(const uint8_t *image, int h, int w, ptrdiff_t stride, int quality, int32_t *zz) {
    return transform_zz_any(image, NULL, h, w, stride, quality, zz);
}

static int encode_any(const uint8_t *image, const int32_t *image32, int h, int w, ptrdiff_t stride, int quality, int32_t *dc, int32_t *ac) {
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    if (h == 0 || w == 0) n = 0;
    int32_t *zz = (int32_t *)malloc((n ? n : 1) * 64 * sizeof(int32_t));
    if (!zz) return TICO_E_SPACE;
    int rc = transform_zz_any(image, image32, h, w, stride, quality, zz);
    if (rc == TICO_OK) {
        int32_t prev = 0;
        for (size_t b = 0; b < n; b++) { /* codec.py:34-36: DPCM over all blocks in raster order */
            int32_t cur = zz[b * 64];
            dc[b] = b ? cur - prev : cur;
            prev = cur;
            memcpy(ac + b * 63, zz + b * 64 + 1, 63 * sizeof(int32_t));
        }
    }
    free(zz);
    return rc;
}

/* encode() with a non-integral quality (codec.py:26-43 with utils.py:50-53 on a float): dc (DPCM'd), ac as tico_encode */
int tico_encode_f(const uint8_t *image, int h, int w, ptrdiff_t stride, double quality, int32_t *dc, int32_t *ac) {
    double div[64];
    int rc = tico_divisors_f(quality, div);
    if (rc) return rc;
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    if (h == 0 || w == 0) n = 0;
    int32_t *zz = (int32_t *)malloc((n ? n : 1) * 64 * sizeof(int32_t));
    if (!zz) return TICO_E_SPACE;
    rc = transform_zz_div(image, NULL, h, w, stride, div, zz);
    if (rc == TICO_OK) {
        int32_t prev = 0;
        for (size_t b = 0; b < n; b++) { /* codec.py:34-36 */
            int32_t cur = zz[b * 64];
            dc[b] = b ? cur - prev : cur;
            prev = cur;
            memcpy(ac + b * 63, zz + b * 64 + 1, 63 * sizeof(int32_t));
        }
    }
    free(zz);
    return rc;
}

int tico_encode(const uint8_t *image, int h, int w, ptrdiff_t stride, int quality, int32_t *dc, int32_t *ac) {
    return encode_any(image, NULL, h, w, stride, quality, dc, ac);
}

/* encode() for integer images outside 0..255: int32 pixels, stride in elements (codec.py:29: astype(int32) - 128). */
int tico_encode_i32(const int32_t *image, int h, int w, ptrdiff_t stride, int quality, int32_t *dc, int32_t *ac) {
    return encode_any(NULL, image, h, w, stride, quality, dc, ac);
}

int tico_encode_zz16(const uint8_t *image, int h, int w, ptrdiff_t stride, int quality, int16_t *out) {
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    if (h == 0 || w == 0) n = 0;
    int32_t *zz = (int32_t *)malloc((n ? n : 1) * 64 * sizeof(int32_t));
    if (!zz) return TICO_E_SPACE;
    int rc = transform_zz(image, h, w, stride, quality, zz);
    if (rc == TICO_OK) {
        for (size_t i = 0; i < n * 64; i++) {
            if (zz[i] < -32768 || zz[i] > 32767) {
                rc = TICO_E_RANGE;
                break;
            }
            out[i] = (int16_t)zz[i];
        }
    }
    free(zz);
    return rc;
}

/* huffman.py:12-33.  Trailing zeros dropped; runs >= 16 emit ZRL=(15,0) per 16; EOB=(0,0) always appended. */
int tico_rle_block(const int32_t ac[63], int32_t *runs, int32_t *vals) {
    int last = -1;
    for (int i = 62; i >= 0; i--) {
        if (ac[i] != 0) {
            last = i;
            break;
        }
    }
    int n = 0, run = 0;
    for (int i = 0; i <= last; i++) {
        if (ac[i] == 0) {
            run++;
            continue;
        }
        while (run >= 16) {
            runs[n] = 15;
            vals[n] = 0;
            n++;
            run -= 16;
        }
        runs[n] = run;
        vals[n] = ac[i];
        n++;
        run = 0;
    }
    runs[n] = 0;
    vals[n] = 0;
    n++;
    return n;
}

/* bitbuffer.py: MSB-first append; to_bytes() pads the last byte with zero bits (bitbuffer.py:17-18). */
typedef struct {
    uint8_t *p;
    size_t cap, nbits;
    int overflow;
} bitw;

static void bw_put(bitw *b, uint32_t value, int nbits) {
    for (int i = nbits - 1; i >= 0; i--) {
        size_t byte = b->nbits >> 3;
        if (byte >= b->cap) {
            b->overflow = 1;
            return;
        }
        if ((b->nbits & 7) == 0) b->p[byte] = 0;
        if ((value >> i) & 1) b->p[byte] |= (uint8_t)(0x80u >> (b->nbits & 7));
        b->nbits++;
    }
}

/* utils.py:9-10: ceil(log2(|x|+1)) == bit length of |x| */
static int bits_required(int32_t v) {
    uint32_t a = (uint32_t)(v < 0 ? -(int64_t)v : v);
    int n = 0;
    while (a) {
        n++;
        a >>= 1;
    }
    return n;
}

/* huffman.py:41-63: codeword of the category, then the low `size` bits of |v| (inverted when v < 0). */
static int put_symbol(bitw *b, const hufftab *t, int run, int32_t v, int is_dc) {
    int size = bits_required(v);
    int sym = is_dc ? size : ((run << 4) | size);
    if (size > 15 || (!is_dc && size > 10) || (is_dc && size > 11) || t->len[sym] == 0) return TICO_E_RANGE;
    bw_put(b, t->code[sym], t->len[sym]);
    if (size) {
        uint32_t a = (uint32_t)(v < 0 ? -v : v) & 0xffffu; /* astype(">u2") */
        if (v < 0) a = ~a;
        bw_put(b, a & ((1u << size) - 1u), size);
    }
    return TICO_OK;
}

int tico_entropy_encode(const int32_t *dc, const int32_t *ac, int h, int w, int quality, uint8_t *out, size_t cap,
                        size_t *out_len) {
    ensure_tables();
    if (quality < 0) return TICO_E_QUALITY;
    if (cap < 16) return TICO_E_SPACE;
    /* codec.py:102-114: struct.pack("III", h, w, q) native little-endian, then 32-bit flag 0 (default table) */
    uint32_t hdr[3] = {(uint32_t)h, (uint32_t)w, (uint32_t)quality};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 4; k++) out[i * 4 + k] = (uint8_t)(hdr[i] >> (8 * k));
    memset(out + 12, 0, 4);
    bitw b = {out, cap, 128, 0};
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    if (h == 0 || w == 0) n = 0;
    int32_t runs[64], vals[64];
    for (size_t i = 0; i < n; i++) { /* codec.py:153-162 */
        int rc = put_symbol(&b, &HT_DC, 0, dc[i], 1);
        if (rc) return rc;
        int m = tico_rle_block(ac + i * 63, runs, vals);
        for (int s = 0; s < m; s++) {
            rc = put_symbol(&b, &HT_AC, runs[s], vals[s], 0);
            if (rc) return rc;
        }
        if (b.overflow) return TICO_E_SPACE;
    }
    if (b.overflow) return TICO_E_SPACE;
    *out_len = (b.nbits + 7) >> 3;
    return TICO_OK;
}

size_t tico_compress_bound(int h, int w) {
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    return 16 + n * 208 + 8; /* 20 DC bits + 63*26 AC bits + 4 EOB bits = 1662 bits < 208 bytes per block */
}

int tico_compress(const uint8_t *image, int h, int w, ptrdiff_t stride, int quality, uint8_t *out, size_t cap,
                  size_t *out_len) {
    size_t n = (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
    if (h == 0 || w == 0) n = 0;
    int32_t *dc = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    int32_t *ac = (int32_t *)malloc((n ? n : 1) * 63 * sizeof(int32_t));
    int rc = (dc && ac) ? tico_encode(image, h, w, stride, quality, dc, ac) : TICO_E_SPACE;
    if (rc == TICO_OK) rc = tico_entropy_encode(dc, ac, h, w, quality, out, cap, out_len);
    free(dc);
    free(ac);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------
 * Decoder (codec.py:167-189, 46-70; huffman.py:36-38, 66-98; bitbuffer.py:20-23, 55-65).
 * ---------------------------------------------------------------------------------------------------- */
int tico_parse_header(const uint8_t *data, size_t len, int *h, int *w, int *quality, uint32_t *flag) {
    if (len < 16) return TICO_E_STREAM;
    uint32_t v[4];
    for (int i = 0; i < 4; i++) /* unpack("IIII") native little-endian, codec.py:119 */
        v[i] = (uint32_t)data[i * 4] | ((uint32_t)data[i * 4 + 1] << 8) | ((uint32_t)data[i * 4 + 2] << 16) |
               ((uint32_t)data[i * 4 + 3] << 24);
    *h = (int)v[0];
    *w = (int)v[1];
    *quality = (int)v[2];
    *flag = v[3];
    return TICO_OK;
}

typedef struct {
    const uint8_t *p;
    size_t nbits, pos;
} bitr;

/* bitbuffer.py:20-23: slicing past the end yields fewer (possibly zero) bits while pos still advances. */
static int br_bit(bitr *r, int *bit) {
    int ok = r->pos < r->nbits;
    if (ok) *bit = (r->p[r->pos >> 3] >> (7 - (r->pos & 7))) & 1;
    r->pos++;
    return ok;
}

/* huffman.py:66-74: extend the prefix one bit at a time until it is a codeword (<= 16 bits) */
static int read_code(bitr *r, const hufftab *t, int *sym) {
    unsigned code = 0;
    int len = 0;
    for (int i = 0; i <= 16; i++) {
        if (len > 0)
            for (int s = 0; s < 256; s++)
                if (t->len[s] == len && t->code[s] == code) {
                    *sym = s;
                    return 1;
                }
        if (i == 16) break;
        int bit;
        if (br_bit(r, &bit)) { /* a read past the end appends nothing to the prefix */
            code = (code << 1) | (unsigned)bit;
            len++;
        }
    }
    /* the reference loop reads up to 17 bits before raising; mirror the position advance */
    r->pos++;
    return 0;
}

/* bitbuffer.py:55-65 */
static int read_int(bitr *r, int size, int32_t *out) {
    if (size == 0) {
        *out = 0;
        return 1;
    }
    uint32_t v = 0;
    int first = -1, got = 0;
    for (int i = 0; i < size; i++) {
        int bit;
        if (br_bit(r, &bit)) {
            if (first < 0) first = bit;
            v = (v << 1) | (uint32_t)bit;
            got++;
        }
    }
    if (got == 0) return 0; /* ret[0] on an empty bitarray raises IndexError */
    if (first == 0) {
        v = (~v) & ((got >= 32) ? 0xffffffffu : ((1u << got) - 1u));
        *out = -(int32_t)v;
    } else {
        *out = (int32_t)v;
    }
    return 1;
}

int tico_decompress(const uint8_t *data, size_t len, uint8_t *out, size_t cap) {
    ensure_tables();
    int h, w, quality;
    uint32_t flag;
    int rc = tico_parse_header(data, len, &h, &w, &quality, &flag);
    if (rc) return rc;
    if (flag & (1u << 31)) return TICO_E_STREAM; /* custom-table streams: not restated (broken in the reference) */
    /* codec.py:127-128: flag 1<<30 marks a stream of the reference's C encoder; its quality field is an exponent (codec.py:59-62) */
    const int scaled = (flag & (1u << 30)) != 0;
    if (scaled ? (quality < 0 || quality > 62) : (quality < 1 || quality > 99)) return TICO_E_QUALITY;
    if ((size_t)h * (size_t)w > cap) return TICO_E_SPACE;
    int bh = (h + 7) / 8, bw = (w + 7) / 8;
    size_t n = (size_t)bh * (size_t)bw;
    int32_t *dc = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    int32_t *ac = (int32_t *)calloc((n ? n : 1) * 63, sizeof(int32_t));
    if (!dc || !ac) {
        free(dc);
        free(ac);
        return TICO_E_SPACE;
    }
    bitr r = {data, len * 8, 128};
    for (size_t i = 0; i < n; i++) { /* codec.py:178-186: any exception inside a block is swallowed */
        int sym;
        int32_t v;
        if (!read_code(&r, &HT_DC, &sym)) continue;
        if (!read_int(&r, sym, &v)) continue;
        dc[i] = v;
        int32_t blk[1200];
        int m = 0, ok = 1, too_long = 0;
        for (;;) { /* huffman.py:87-96 then decode_run_length (huffman.py:36-38) */
            if (!read_code(&r, &HT_AC, &sym)) {
                ok = 0;
                break;
            }
            int run = sym >> 4, size = sym & 15;
            if (!read_int(&r, size, &v)) {
                ok = 0;
                break;
            }
            /* the reference keeps reading symbols until EOB or a decode error however long the list grows, and only then
             * rejects a list longer than 63: keep consuming, stop storing */
            if (m + run + 1 > 1100) too_long = 1;
            if (!too_long) {
                for (int z = 0; z < run; z++) blk[m++] = 0;
                blk[m++] = v;
            }
            if (sym == 0) break; /* EOB */
        }
        if (!ok || too_long) continue;
        m -= 1;            /* [:-1] */
        if (m > 63) continue; /* ac[i, :len] = ... raises on a too-long block -> swallowed, ac stays zero */
        memcpy(ac + i * 63, blk, (size_t)m * sizeof(int32_t));
    }
    /* codec.py:46-70 */
    double div[64], ann[64], pow2 = 1.0;
    tico_divisors(scaled ? 50 : quality, div); /* codec.py:62: quality = 50 on the scaled branch */
    if (scaled) {
        for (int k = 0; k < 64; k++) ann[k] = (double)ANNSCALES_INT[k] / 2048.0; /* exact */
        pow2 = ldexp(1.0, quality);                                              /* 2 ** quality */
    }
    /* np.cumsum(dc) over all blocks in raster order (codec.py:53) */
    {
        int32_t run_dc = 0;
        for (size_t b = 0; b < n; b++) {
            run_dc += dc[b];
            dc[b] = run_dc;
        }
    }
    for (int by = 0; by < bh; by++) {
        for (int bx = 0; bx < bw; bx++) {
            size_t b = (size_t)by * bw + bx;
            double X[64], Y[64];
            if (scaled) { /* codec.py:60-61: coeffs / ANNSCALES, then *= 2 ** quality, then the inverse quantiser at quality 50 */
                X[0] = (((double)dc[b] / ann[0]) * pow2) * div[0];
                for (int k = 1; k < 64; k++) {
                    int nat = ZIGZAG[k];
                    X[nat] = (((double)ac[b * 63 + (k - 1)] / ann[nat]) * pow2) * div[nat];
                }
            } else {
                X[0] = (double)dc[b] * div[0];
                for (int k = 1; k < 64; k++) {
                    int nat = ZIGZAG[k];
                    X[nat] = (double)ac[b * 63 + (k - 1)] * div[nat]; /* utils.py:52 */
                }
            }
            tico_block_idct(X, Y);
            for (int i = 0; i < 8; i++) {
                int y = by * 8 + i;
                if (y >= h) break;
                for (int j = 0; j < 8; j++) {
                    int x = bx * 8 + j;
                    if (x >= w) break;
                    double v = Y[i * 8 + j] + 128.0; /* codec.py:68-70: clip then truncating astype(uint8) */
                    if (v < 0.0) v = 0.0;
                    if (v > 255.0) v = 255.0;
                    out[(size_t)y * w + x] = (uint8_t)v;
                }
            }
        }
    }
    free(dc);
    free(ac);
    return TICO_OK;
}
