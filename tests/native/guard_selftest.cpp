// guard_selftest.cpp - the guard band of the float32 fast path (kGuard in tic_math.h), checked on the CPU.
//
// tic_math.h is host-compilable: this program runs the kernel's own dct8_aan<float> (columns, then rows: the pass order of
// the strip kernel since round 5, which is the reference's) and the kernel's quantiser arithmetic against dct8_exact (columns, then
// rows, float64) and asserts, for every coefficient of every block, with G[u][v] = guard_cf(u, v) = kGuard[v][u]
// (the bound table is stated for the rows-first algorithm and is symmetric under transposition of block and algorithm),
//     |Z_fast * scale - X_exact| < G[u][v]                           (coefficient units)
//     |t_fast - X_exact / div|   < G[u][v] / div                     (quantised units, q = 1, 10, 50, 90, 99; t = the exact product
//                                                                     z * mul that lives inside quant_fma's fused multiply-adds)
// and, end to end, the kernel's own decision: wherever the accept test of the strip kernel (d = fmaf(z, mul, magic - s) against
// the float thresholds thrR of build_consts) ACCEPTS a rounding, that rounding is rint(X_exact / div), the reference's.
// on: blocks read from a file (the adversarial blocks of tools/fastpath_error_search.py, tests/golden/adversarial_blocks.npz
// exported as raw bytes by the test), extreme patterns, and N random blocks (argv[2], default 2,000,000).
// It also prints kGuard so that the test can compare it with the rigorous bound of tools/fastpath_error_bound.py.
// Build: g++ -O2 -std=c++17 -ffp-contract=off (tests/test_host_cpu.py).
#include <initializer_list>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../tinyimgcodec_amd/csrc/tic_math.h"

using namespace tic;

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

static DctqConsts C1, C10, C50, C90, C99;
static long n_accept = 0, n_trip = 0, n_wrong = 0;
static long n_trip_q[5][2]; // per quality: trips of the 60 irrational / the 4 rational coefficients
static double aan[8];
static double worst_ratio[64];  // max over blocks of error / kGuard, coefficient units
static double worst_ratio_q[64]; // the same in quantised units (worst of the three qualities)
static long n_blocks = 0, n_viol = 0;

static void check_block_order(const uint8_t px[64], bool cols_first);
static void check_block(const uint8_t px[64]) { // both instantiations of the strip kernel
    check_block_order(px, true);
    check_block_order(px, false);
}
static void check_block_order(const uint8_t px[64], bool cols_first) {
    // fast path: pass 1 down the pixel columns (columns first) or along the pixel rows (rows first), level shift folded into output 0
    float z[8][8]; // z[u][v]
    if (cols_first) {
        float y[8][8]; // y[u][c]
        for (int c = 0; c < 8; c++) {
            float d[8];
            for (int r = 0; r < 8; r++) d[r] = (float)px[r * 8 + c];
            dct8_aan(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
            d[0] -= 1024.0f;
            for (int u = 0; u < 8; u++) y[u][c] = d[u];
        }
        for (int u = 0; u < 8; u++) {
            float e[8];
            for (int c = 0; c < 8; c++) e[c] = y[u][c];
            dct8_aan(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            for (int v = 0; v < 8; v++) z[u][v] = e[v];
        }
    } else {
        float y[8][8]; // y[r][v]
        for (int r = 0; r < 8; r++) {
            float d[8];
            for (int c = 0; c < 8; c++) d[c] = (float)px[r * 8 + c];
            dct8_aan(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
            d[0] -= 1024.0f;
            for (int v = 0; v < 8; v++) y[r][v] = d[v];
        }
        for (int v = 0; v < 8; v++) {
            float e[8];
            for (int r = 0; r < 8; r++) e[r] = y[r][v];
            dct8_aan(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            for (int u = 0; u < 8; u++) z[u][v] = e[u];
        }
    }
    // reference order in float64: axis -2 (down the columns) first, then axis -1
    double x[8][8];
    for (int c = 0; c < 8; c++) {
        double d[8];
        for (int r = 0; r < 8; r++) d[r] = (double)((int)px[r * 8 + c] - 128);
        dct8_exact(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        for (int u = 0; u < 8; u++) x[u][c] = d[u];
    }
    for (int u = 0; u < 8; u++) dct8_exact(x[u][0], x[u][1], x[u][2], x[u][3], x[u][4], x[u][5], x[u][6], x[u][7]);
    if (cols_first) n_blocks++;
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            const int i = u * 8 + v;
            const double fast = (double)z[u][v] / (aan[u] * aan[v] * 8.0);
            const double G = cols_first ? guard_cf(u, v) : kGuard[i];
            const double r1 = fabs(fast - x[u][v]) / G;
            if (r1 > worst_ratio[i]) worst_ratio[i] = r1;
            double rq = 0;
            for (const DctqConsts *C : {&C1, &C10, &C50, &C90, &C99}) {
                const float mul = cols_first ? C->mulN[i] : C->mulT[v * 8 + u];
                const double t_fused = (double)z[u][v] * (double)mul;       // the exact product inside the fma
                const double want = x[u][v] / C->div[i];
                const double r = fabs(t_fused - want) / (G / C->div[i]);
                if (r > rq) rq = r;
                // the kernel's accept test, operation for operation (quant_fma + the max/compare of the strip kernel)
                const float s = fmaf(z[u][v], mul, kMagic);
                const float nr = kMagic - s;
                const float d = fmaf(z[u][v], mul, nr);
                const float thr = cols_first ? C->thrR[4 * u + ((v == 0 || v == 4) ? 2 : (v < 4 ? 0 : 1))]   // three groups per frequency row
                                             : C->thrG[4 * v + ((u == 0 || u == 4) ? 2 : (u < 4 ? 0 : 1))];  // ... per frequency column
                if (fabsf(d) > thr) {
                    n_trip++;
                    const int qi = C == &C1 ? 0 : C == &C10 ? 1 : C == &C50 ? 2 : C == &C90 ? 3 : 4;
                    n_trip_q[qi][((u & 3) == 0 && (v & 3) == 0) ? 1 : 0]++;
                } else {
                    n_accept++;
                    uint32_t bits;
                    memcpy(&bits, &s, 4);
                    const int got = (int)(int16_t)(bits & 0xffffu);
                    if (got != (int)rint(want)) n_wrong++;
                }
            }
            if (rq > worst_ratio_q[i]) worst_ratio_q[i] = rq;
            if (r1 >= 1.0 || rq >= 1.0) n_viol++;
        }
}

int main(int argc, char **argv) {
    const char *file = argc > 1 ? argv[1] : "";
    const long n_random = argc > 2 ? atol(argv[2]) : 2000000;
    { // round 4: build_consts takes any number in [1, 99].  For every INTEGER quality its divisors must be bit-identical to the
      // reference's integer recipe (utils.py:50-53: an int factor 200 - 2 q for q >= 50), whether the quality arrives as int or double
        static DctqConsts a, b;
        for (int q = 1; q <= 99; q++) {
            if (!build_consts(q, &a) || !build_consts((double)q, &b)) { printf("FAIL build_consts(%d)\n", q); return 1; }
            for (int i = 0; i < 64; i++) {
                const double want = q < 50 ? ((double)kQTable[i] * (5000.0 / (double)q)) / 100.0 : (double)(kQTable[i] * (200 - 2 * q)) / 100.0;
                if (a.div[i] != want || b.div[i] != want || a.mulN[i] != b.mulN[i] || a.thrR[i & 31] != b.thrR[i & 31] || a.mulT[i] != b.mulT[i] || a.thrG[i & 31] != b.thrG[i & 31]) { printf("FAIL divisors of quality %d\n", q); return 1; }
            }
        }
        if (build_consts(0.999, &a) || build_consts(99.001, &a) || build_consts(nan(""), &a) || !build_consts(37.5, &a)) { printf("FAIL range of build_consts\n"); return 1; }
    }
    build_consts(1, &C1);
    build_consts(99, &C99);
    build_consts(10, &C10);
    build_consts(50, &C50);
    build_consts(90, &C90);
    aan[0] = 1.0;
    for (int k = 1; k < 8; k++) aan[k] = sqrt(2.0) * cos(k * 3.14159265358979323846 / 16.0);
    printf("kGuard");
    for (int i = 0; i < 64; i++) printf(" %.6e", kGuard[i]);
    printf("\n");
    uint8_t px[64];
    long n_file = 0;
    if (file[0]) {
        FILE *f = fopen(file, "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", file); return 2; }
        while (fread(px, 1, 64, f) == 64) { check_block(px); n_file++; }
        fclose(f);
    }
    // extreme patterns: constants, checkerboards, stripes, single pixels, half planes
    for (int pat = 0; pat < 64; pat++) {
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) {
                int v;
                switch (pat & 7) {
                case 0: v = (pat >> 3) * 36; break;
                case 1: v = ((r + c) & 1) ? 255 : 0; break;
                case 2: v = (c & 1) ? 255 : 0; break;
                case 3: v = (r & 1) ? 255 : 0; break;
                case 4: v = (r * 8 + c == (pat >> 3) * 9) ? 255 : 0; break;
                case 5: v = (c < (pat >> 3) + 1) ? 255 : 0; break;
                case 6: v = (r < (pat >> 3) + 1) ? 0 : 255; break;
                default: v = ((r / 2 + c / 2) & 1) ? 255 : 0; break;
                }
                px[r * 8 + c] = (uint8_t)(v > 255 ? 255 : v);
            }
        check_block(px);
    }
    for (long k = 0; k < n_random; k++) {
        const uint32_t mode = k & 3; // uniform bytes / binary 0,255 / near-saturated / low-contrast
        for (int j = 0; j < 64; j += 4) {
            uint32_t w = rnd();
            for (int t = 0; t < 4; t++) {
                uint32_t b = (w >> (8 * t)) & 0xff;
                if (mode == 1) b = (b & 1) ? 255 : 0;
                else if (mode == 2) b = (b & 1) ? 255 - (b >> 5) : (b >> 5);
                else if (mode == 3) b = 120 + (b & 15);
                px[j + t] = (uint8_t)b;
            }
        }
        check_block(px);
    }
    double w1 = 0, wq = 0;
    for (int i = 0; i < 64; i++) {
        if (worst_ratio[i] > w1) w1 = worst_ratio[i];
        if (worst_ratio_q[i] > wq) wq = worst_ratio_q[i];
    }
    printf("blocks %ld (file %ld) worst error/guard: coefficient units %.4f, quantised units %.4f, violations %ld\n", n_blocks, n_file, w1, wq, n_viol);
    printf("accept test: %ld accepted, %ld tripped, %ld accepted roundings differ from the reference\n", n_accept, n_trip, n_wrong);
    printf("trips per million coefficients (irrational / rational), q = 1, 10, 50, 90, 99:");
    for (int k = 0; k < 5; k++) printf("  %.1f / %.1f", 1e6 * n_trip_q[k][0] / (60.0 * n_blocks), 1e6 * n_trip_q[k][1] / (4.0 * n_blocks));
    printf("\n");
    if (n_viol != 0 || w1 >= 1.0 || wq >= 1.0 || n_wrong != 0) {
        printf("guard_selftest FAILED\n");
        return 1;
    }
    printf("guard_selftest ok\n");
    return 0;
}
