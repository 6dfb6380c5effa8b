// tic_math.h - per-lane arithmetic of the transform stage, usable from device code (hipcc) and from host
// code (g++; used by the host-side constant builder and by tests that emulate a block on the CPU).
//
// Compile every translation unit that includes this with -ffp-contract=off: the exact path below requires each
// + - * to be a single IEEE-754 operation in scipy/pocketfft's order, and the fast path
// names its fused multiply-adds explicitly.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "tic_tables.h"

#if defined(__HIPCC__)
#define TIC_HD __host__ __device__ __forceinline__
#else
#define TIC_HD inline
#endif

namespace tic {

// ---------------------------------------------------------------------------------------------------------
// Exact path: scipy.fftpack.dct(x, norm="ortho") for N=8 in pocketfft's operation order
// (T_dcst23 type 2 -> radb2(ido=4) -> radb4(ido=1) -> scale -> twiddle post-pass).  Replaces the arithmetic of
// block_dct, utils.py:32-37.  In place on eight named values.
// ---------------------------------------------------------------------------------------------------------
TIC_HD void dct8_exact(double &c0, double &c1, double &c2, double &c3, double &c4, double &c5, double &c6,
                       double &c7) {
#pragma clang fp contract(off)
    // pre-processing
    c0 = c0 * 2.0;
    c7 = c7 * 2.0;
    double t;
    t = c2; c2 = t - c1; c1 = t + c1;
    t = c4; c4 = t - c3; c3 = t + c3;
    t = c6; c6 = t - c5; c5 = t + c5;
    // radb2
    double h0 = c0 + c7, h4 = c0 - c7;
    double h3 = 2.0 * c3, h7 = -2.0 * c4;
    double h1 = c1 + c5, tr2 = c1 - c5;
    double ti2 = c2 + c6, h2 = c2 - c6;
    double pa = kWR * ti2, pb = kWI * tr2;
    double h6 = pa + pb;
    double pc = kWR * tr2, pd = kWI * ti2;
    double h5 = pc - pd;
    // radb4, k = 0
    double a2 = h0 + h3, a1 = h0 - h3, a3 = 2.0 * h1, a4 = 2.0 * h2;
    double o0 = a2 + a3, o4 = a2 - a3, o6 = a1 + a4, o2 = a1 - a4;
    // radb4, k = 1
    double b2 = h4 + h7, b1 = h4 - h7, b3 = 2.0 * h5, b4 = 2.0 * h6;
    double o1 = b2 + b3, o5 = b2 - b3, o7 = b1 + b4, o3 = b1 - b4;
    // normalisation 1/sqrt(2N) = 0.25
    c0 = o0 * 0.25; c1 = o1 * 0.25; c2 = o2 * 0.25; c3 = o3 * 0.25;
    c4 = o4 * 0.25; c5 = o5 * 0.25; c6 = o6 * 0.25; c7 = o7 * 0.25;
    // post-processing (k, kc) = (1,7), (2,6), (3,5)
    double p1, p2, p3, p4, t1, t2;
    p1 = kTW0 * c7; p2 = kTW6 * c1; t1 = p1 + p2; p3 = kTW0 * c1; p4 = kTW6 * c7; t2 = p3 - p4;
    c1 = 0.5 * (t1 + t2); c7 = 0.5 * (t1 - t2);
    p1 = kTW1 * c6; p2 = kTW5 * c2; t1 = p1 + p2; p3 = kTW1 * c2; p4 = kTW5 * c6; t2 = p3 - p4;
    c2 = 0.5 * (t1 + t2); c6 = 0.5 * (t1 - t2);
    p1 = kTW2 * c5; p2 = kTW4 * c3; t1 = p1 + p2; p3 = kTW2 * c3; p4 = kTW4 * c5; t2 = p3 - p4;
    c3 = 0.5 * (t1 + t2); c5 = 0.5 * (t1 - t2);
    c4 = c4 * kTW3;
    c0 = c0 * kSq2h;
}

// scipy.fftpack.idct(x, norm="ortho") (DCT-III), pocketfft order: T_dcst23 type 3 -> radf4(ido=1) -> radf2(ido=4).
// Replaces the arithmetic of block_idct, utils.py:40-45.
// kQuarter = false leaves out the normalisation by 1/4 (eight multiplications per call): the results are then EXACTLY four times the
// normalised ones, bit for bit - every operation here is an addition, a subtraction or a multiplication by a constant, and scaling all
// inputs of such an operation by a power of two scales its rounded result by the same power (no overflow or underflow in reach: the
// values are sums of |coefficient x quantiser| <= 2^23 times constants near one, and a non-zero difference of two such doubles is no
// smaller than 2^-40).  The device decoder's fused kernel runs both passes that way and folds the 1/16 into the pixel conversion.
template <bool kQuarter>
TIC_HD void idct8_exact_impl(double &c0, double &c1, double &c2, double &c3, double &c4, double &c5, double &c6, double &c7) {
#pragma clang fp contract(off)
    c0 = c0 * kSqrt2;
    double t1, t2, p1, p2, p3, p4;
    t1 = c1 + c7; t2 = c1 - c7; p1 = kTW0 * t2; p2 = kTW6 * t1; p3 = kTW0 * t1; p4 = kTW6 * t2;
    c1 = p1 + p2; c7 = p3 - p4;
    t1 = c2 + c6; t2 = c2 - c6; p1 = kTW1 * t2; p2 = kTW5 * t1; p3 = kTW1 * t1; p4 = kTW5 * t2;
    c2 = p1 + p2; c6 = p3 - p4;
    t1 = c3 + c5; t2 = c3 - c5; p1 = kTW2 * t2; p2 = kTW4 * t1; p3 = kTW2 * t1; p4 = kTW4 * t2;
    c3 = p1 + p2; c5 = p3 - p4;
    c4 = c4 * (2.0 * kTW3);
    // radf4 (k = 0 -> a0..a3 from c0,c2,c4,c6 ; k = 1 -> a4..a7 from c1,c3,c5,c7)
    double tr1 = c6 + c2, a2 = c6 - c2, tr2 = c0 + c4, a1 = c0 - c4;
    double a0 = tr2 + tr1, a3 = tr2 - tr1;
    double ur1 = c7 + c3, a6 = c7 - c3, ur2 = c1 + c5, a5 = c1 - c5;
    double a4 = ur2 + ur1, a7 = ur2 - ur1;
    // radf2
    double r0 = a0 + a4, r7 = a0 - a4, r4 = -a7, r3 = a3;
    double m1 = kWR * a5, m2 = kWI * a6, q2 = m1 + m2;
    double m3 = kWR * a6, m4 = kWI * a5, qi = m3 - m4;
    double r1 = a1 + q2, r5 = a1 - q2, r2 = qi + a2, r6 = qi - a2;
    if (kQuarter) {
        c0 = r0 * 0.25; c1 = r1 * 0.25; c2 = r2 * 0.25; c3 = r3 * 0.25;
        c4 = r4 * 0.25; c5 = r5 * 0.25; c6 = r6 * 0.25; c7 = r7 * 0.25;
    } else {
        c0 = r0; c1 = r1; c2 = r2; c3 = r3;
        c4 = r4; c5 = r5; c6 = r6; c7 = r7;
    }
    double t;
    t = c1; c1 = t - c2; c2 = t + c2;
    t = c3; c3 = t - c4; c4 = t + c4;
    t = c5; c5 = t - c6; c6 = t + c6;
}
TIC_HD void idct8_exact(double &c0, double &c1, double &c2, double &c3, double &c4, double &c5, double &c6, double &c7) {
    idct8_exact_impl<true>(c0, c1, c2, c3, c4, c5, c6, c7);
}

// ---------------------------------------------------------------------------------------------------------
// a / b, correctly rounded (round to nearest even: the result of the IEEE division the reference's np.round(X / div) starts from,
// utils.py:53), from y = RN(1 / b) in five multiply-adds.  The compiler's own float64 division is ~13 instructions on gfx950, four of
// them at a quarter of the float64 rate (v_div_scale x 2, v_rcp_f64, v_div_fmas, v_div_fixup: they also cover subnormals, infinities
// and exponent extremes, none of which occur here); rational_quad runs it once per tie strip, and tie strips are a sixth of the strips of a
// noise frame and most strips of flat, banded or posterised content.
// Why it is exact (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, section 4.7): q0 = RN(a y) is within 2 ulps of a / b;
// r0 = a - q0 b is exactly representable (q0 is that close) and the fused multiply-add delivers it without rounding; q1 = RN(q0 + r0 y)
// is a FAITHFUL rounding of a / b (error below one ulp); r1 = a - q1 b is again exact; and for a faithful q1 and y = RN(1 / b) - relative
// error below 2^-53, which correct rounding of 1 / b guarantees - RN(q1 + r1 y) IS RN(a / b) (Markstein's theorem; a / b cannot lie on the
// midpoint of two floating-point numbers, and the perturbation r1 y - r1 / b is too small to carry it across one).  Preconditions: b and
// y normal, no overflow or underflow in a y and q b - here |a| <= 2^14 and 0.02 <= b <= 10^4.  y comes from the host's IEEE division
// (build_consts), never from an iteration on the device.  tests/native/div_selftest.cpp checks the function against the compiler's division on
// every divisor of every quality and 10^8 numerators, among them every tie point (k + 1/2) b and its neighbours within 8 ulps.
// ---------------------------------------------------------------------------------------------------------
TIC_HD double div_rn(double a, double b, double y /* RN(1 / b) */) {
    const double q0 = a * y;
    const double r0 = fma(-q0, b, a);
    const double q1 = fma(r0, y, q0);
    const double r1 = fma(-q1, b, a);
    return fma(r1, y, q1);
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: Arai-Agui-Nakajima scaled 8-point DCT in float32 (5 multiplies, FMAs named explicitly).
// Output k is the orthonormal DCT-II coefficient times aan[k]*sqrt(8), aan[0] = 1, aan[k] = sqrt(2)*cos(k*pi/16);
// the scale is folded into the quantiser multiplier.  o0 and o4 are plain sums/differences of the inputs, hence
// exact integers when the inputs are (needed by the exact sub-path for coefficients (0,0),(0,4),(4,0),(4,4)).
// ---------------------------------------------------------------------------------------------------------
TIC_HD float tic_fma(float a, float b, float c) { return fmaf(a, b, c); }
TIC_HD double tic_fma(double a, double b, double c) { return fma(a, b, c); }

template <typename T>
TIC_HD void dct8_aan(T &d0, T &d1, T &d2, T &d3, T &d4, T &d5, T &d6, T &d7) {
#pragma clang fp contract(off)
    const T c707 = (T)0.70710678118654752440, c382 = (T)0.38268343236508977173;
    const T c541 = (T)0.54119610014619698440, c1306 = (T)1.30656296487637652786;
    T t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    T t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    T t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    d0 = t10 + t11;
    d4 = t10 - t11;
    T s = t12 + t13;
    d2 = tic_fma(s, c707, t13);
    d6 = tic_fma(s, -c707, t13);
    T u10 = t4 + t5, u11 = t5 + t6, u12 = t6 + t7;
    T z5 = (u10 - u12) * c382;
    T z2 = tic_fma(u10, c541, z5);
    T z4 = tic_fma(u12, c1306, z5);
    T z11 = tic_fma(u11, c707, t7);
    T z13 = tic_fma(u11, -c707, t7);
    d5 = z13 + z2;
    d3 = z13 - z2;
    d1 = z11 + z4;
    d7 = z11 - z4;
}

// Guard band of the fast path per coefficient (u = vertical, v = horizontal frequency), in coefficient (X) units: a COMPLETE
// forward error bound of what the strip kernel computes against what the reference computes, for every block of uint8 pixels
// (tools/fastpath_error_bound.py, which prints this table; tests/test_host_cpu.py asserts table >= bound):
//   systematic  dct8_aan<float> multiplies by float32(0.7071...) etc., not by the real numbers: the largest value over the pixel
//               box of the difference between the float-constant algorithm in exact arithmetic and the true DCT, computed exactly;
//   rounding    half an ulp of the largest magnitude of every float32 result that is not provably exact (integer sums are),
//               carried through the two passes (rows, then columns);
//   multiplier  the float32 quantiser multiplier's representation error, 1024 * 2^-24 (the product is not rounded: quant_fma);
//   reference   the reference's own float64 error (~1e-12);
// times 1.02.  |X_fast - X_reference| < kGuard[u][v] for every input, so a rounding decided outside the band is the reference's
// rounding.  The float32 resolution of the accept test itself is taken out of the thresholds by build_consts() (thr_below).
// The largest error an adversarial search found (tools/fastpath_error_search.py) is 4.6e-4 at (7,7), a factor 6 below.
static constexpr double kGuard[64] = {
    6.2256e-05, 3.2315e-04, 2.6249e-04, 3.4557e-04, 6.2256e-05, 3.3290e-04, 2.0056e-04, 5.9516e-04,
    1.8850e-04, 3.9880e-04, 3.3279e-04, 4.4030e-04, 1.8850e-04, 4.2003e-04, 2.8730e-04, 7.1855e-04,
    1.1955e-04, 3.1685e-04, 2.7439e-04, 3.5239e-04, 1.1955e-04, 3.3094e-04, 2.2674e-04, 5.9192e-04,
    1.8674e-04, 3.7822e-04, 3.2467e-04, 4.1881e-04, 1.8674e-04, 3.9724e-04, 2.9065e-04, 6.9278e-04,
    6.2256e-05, 3.2315e-04, 2.6249e-04, 3.4557e-04, 6.2256e-05, 3.3290e-04, 2.0056e-04, 5.9516e-04,
    2.1405e-04, 4.9996e-04, 4.2070e-04, 5.6244e-04, 2.1405e-04, 5.6214e-04, 3.5821e-04, 9.0797e-04,
    1.1428e-04, 6.1485e-04, 5.1255e-04, 7.0510e-04, 1.1428e-04, 6.6162e-04, 3.4923e-04, 1.0793e-03,
    5.1054e-04, 1.5901e-03, 1.3262e-03, 1.8402e-03, 5.1054e-04, 1.7563e-03, 1.0117e-03, 2.7581e-03,
};
// The strip kernel takes the passes COLUMNS first (as the reference does), then rows: the bound above is symmetric under transposition
// of block and algorithm (the columns-first algorithm on x is the rows-first algorithm on the transposed block, read transposed), so
// its guard band at (u,v) is kGuard[v][u].
constexpr double guard_cf(int u, int v) { return kGuard[v * 8 + u]; }
// Adding 1.5*2^23 to a float |t| < 2^22 rounds it to an integer (half-even) whose two's complement sits in the
// low mantissa bits: the quantiser needs no v_rndne / v_cvt.
static constexpr float kMagic = 12582912.0f;

// Quality-dependent constants consumed by the kernels (one instance per quality, resident in HBM).
struct DctqConsts {
    double div[64];      // natural order u*8+v: (Q*factor)/100 as the reference computes it (utils.py:50-53)
    double rdiv[64];     // fl(1/div)
    float mulN[64];      // fast path, index u*8+v: 1 / (aan[u]*aan[v]*8*div[u][v]) (lane u of a block holds v = 0..7)
    double mul64[64];    // second level (the same butterflies in float64), index u*8+v: the same number in float64
    // the rows-first instantiation of the strip kernel (multi-round grids: DESIGN.md 5.1) holds a frequency COLUMN per lane:
    float mulT[64];      // index v*8+u: the multiplier of (u,v)
    float thrG[32];      // per column v: [4v] = accept threshold for u in {1,2,3}, [4v+1] for u in {5,6,7}, [4v+2] for u in {0,4} (band kGuard[u][v]/div)
    uint16_t zzofsT[64]; // index v*8+u: byte offset of (u,v) in the zig-zag image
    float thrR[32];      // strip kernel, per frequency row u: [4u] = accept threshold for v in {1,2,3}, [4u+1] for v in {5,6,7}, [4u+2] for
                         // v in {0,4}, [4u+3] unused (0.5 - largest guard band guard_cf(u,v)/div[u][v] of the group; accept when
                         // |t - rint(t)| <= thr).  Three groups: the largest guard/div of a group stands for all its members, and with the
                         // triples the sum over the 60 irrational coefficients of (group maximum) is 1.4x the sum of their own bands
                         // (2 x v_max3 + v_max, 3 compares per strip)
    double cosm[64];     // orthonormal DCT-II matrix, index k*8+n: c(k) cos((2n+1) k pi / 16) (direct float64 recompute)
    uint16_t zzofs[64];  // index u*8+v: byte offset of natural coefficient (u,v) in the block's zig-zag int16[64]
    uint8_t zznat[64];   // natural index u*8+v of scan position k (= kZigzag)
    // Everything the strip kernel needs, packed as the image its workgroups copy into LDS with one 16-byte load per lane
    // (164 lanes): [0,256) mulN, [256,384) thrR, [384,512) zzofs, [512,576) div, then 1/div, of the rational coefficients (0,0) (0,4) (4,0) (4,4),
    // [576,1088) cosm, [1088,1600) rdiv, [1600,2112) mul64, [2112,2368) mulT, [2368,2496) thrG, [2496,2624) zzofsT.
    alignas(16) unsigned char strip_blk[2624];
};
constexpr int kStripBlkBytes = 2624;
constexpr int kStripBlkPieces = kStripBlkBytes / 16;
constexpr int kBlkMul = 0, kBlkThr = 256, kBlkZz = 384, kBlkRat = 512, kBlkCos = 576, kBlkRdiv = 1088, kBlkMul64 = 1600; // offsets inside strip_blk
constexpr int kBlkMulT = 2112, kBlkThrT = 2368, kBlkZzT = 2496;

// Accept threshold of the fast path as a float: the kernel accepts a rounding when fl32(|t - rint(t)|) <= thr.  The distance
// is a float32 result in [0, 0.5]: above 0.25 it is rounded by at most 2^-26, i.e. by less than the gap between thr and the next
// float below 0.5.  Rounding tau DOWN to a float and stepping one float further down therefore guarantees
// accepted  =>  |t - rint(t)| < tau, whatever tau's and the distance's own roundings were.
inline float thr_below(double tau) {
    float f = (float)tau;
    if ((double)f > tau) f = nextafterf(f, 0.0f);
    return nextafterf(f, 0.0f);
}

// utils.py:50-53 divisor recipe: factor = 5000 / q if q < 50 else 200 - 2 q; divisor = (Q * factor) / 100, in
// that order, in float64.  `quality` is any number in [1, 99] - the reference computes with whatever it is given (an int quality gives
// an int factor for q >= 50, whose product with the table entry is the same number as the float64 product here: both are exact).
// Returns false when quality is outside 1..99 (or not a number).
inline bool build_consts(double quality, DctqConsts *c) {
#pragma clang fp contract(off)
    if (!(quality >= 1.0 && quality <= 99.0)) return false;
    const double factor = quality < 50.0 ? 5000.0 / quality : 200.0 - 2.0 * quality;
    for (int i = 0; i < 64; i++) {
        const double p = (double)kQTable[i] * factor;
        c->div[i] = p / 100.0;
        c->rdiv[i] = 1.0 / c->div[i];
    }
    double aan[8];
    aan[0] = 1.0;
    for (int k = 1; k < 8; k++) aan[k] = sqrt(2.0) * cos(k * 3.14159265358979323846 / 16.0);
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            c->mul64[u * 8 + v] = 1.0 / (aan[u] * aan[v] * 8.0 * c->div[u * 8 + v]);
            c->mulN[u * 8 + v] = (float)c->mul64[u * 8 + v];
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) c->mulT[v * 8 + u] = c->mulN[u * 8 + v];
    for (int v = 0; v < 8; v++) { // rows-first instantiation: the bound as tabulated, groups of u per column v
        double g3[3] = {0.0, 0.0, 0.0};
        for (int u = 0; u < 8; u++) {
            const double g = kGuard[u * 8 + v] / c->div[u * 8 + v];
            const int grp = (u == 0 || u == 4) ? 2 : (u < 4 ? 0 : 1);
            if (g > g3[grp]) g3[grp] = g;
        }
        for (int k = 0; k < 3; k++) c->thrG[4 * v + k] = thr_below(0.5 - g3[k]);
        c->thrG[4 * v + 3] = 0.0f;
    }
    for (int u = 0; u < 8; u++) {
        double g3[3] = {0.0, 0.0, 0.0};
        for (int v = 0; v < 8; v++) {
            const double g = guard_cf(u, v) / c->div[u * 8 + v];
            const int grp = (v == 0 || v == 4) ? 2 : (v < 4 ? 0 : 1);
            if (g > g3[grp]) g3[grp] = g;
        }
        for (int k = 0; k < 3; k++) c->thrR[4 * u + k] = thr_below(0.5 - g3[k]);
        c->thrR[4 * u + 3] = 0.0f;
    }
    for (int k = 0; k < 8; k++)
        for (int n = 0; n < 8; n++)
            c->cosm[k * 8 + n] = (k == 0 ? sqrt(0.125) : 0.5) * cos((2 * n + 1) * k * 3.14159265358979323846 / 16.0);
    for (int k = 0; k < 64; k++) {
        int nat = kZigzag[k];
        c->zzofs[nat] = (uint16_t)(2 * k);
        c->zzofsT[(nat & 7) * 8 + (nat >> 3)] = (uint16_t)(2 * k);
        c->zznat[k] = (uint8_t)nat;
    }
    {
        unsigned char *p = c->strip_blk;
        memset(p, 0, kStripBlkBytes);
        memcpy(p + kBlkMul, c->mulN, 256);
        memcpy(p + kBlkThr, c->thrR, 128);
        memcpy(p + kBlkZz, c->zzofs, 128);
        const int rat[4] = {0, 4, 32, 36};
        for (int k = 0; k < 4; k++) {
            memcpy(p + kBlkRat + 8 * k, &c->div[rat[k]], 8);
            memcpy(p + kBlkRat + 32 + 8 * k, &c->rdiv[rat[k]], 8);
        }
        memcpy(p + kBlkCos, c->cosm, 512);
        memcpy(p + kBlkRdiv, c->rdiv, 512);
        memcpy(p + kBlkMul64, c->mul64, 512);
        memcpy(p + kBlkMulT, c->mulT, 256);
        memcpy(p + kBlkThrT, c->thrG, 128);
        memcpy(p + kBlkZzT, c->zzofsT, 128);
    }
    return true;
}
inline bool build_consts(int quality, DctqConsts *c) { return build_consts((double)quality, c); }


} // namespace tic
