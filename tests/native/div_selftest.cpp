// div_selftest.cpp - tic_math.h's div_rn (a / b correctly rounded from y = RN(1 / b) in five multiply-adds: what rational_quad and
// rational_slim of the strip kernel divide with since round 6) against the compiler's IEEE division, on the CPU.
//
// Divisors: every entry of the quantiser table at every quality 1..99 (utils.py:50-53 as build_consts computes it), plus the
// divisors of 2,000 non-integral qualities.  Numerators, per divisor:
//   * every tie point T = (k + 1/2) b of the quantiser, k = -1100 .. 1100, rounded to a double, and its neighbours within 8 ulps -
//     the reference's rational coefficients on a tie are exactly such numbers (the exact value is T, pocketfft's rounding puts the
//     computed one a few ulps beside it), and which side decides the rounding;
//   * every multiple of 1/8 up to 1,100 (the exact values of the rational coefficients);
//   * random doubles of random magnitude up to 2^14.
// For each pair: div_rn(a, b, 1 / b) must be bit-identical to a / b, and rint of both the same integer.
// Build: g++ -O2 -std=c++17 -ffp-contract=off (tests/test_host_cpu.py); prints the counts, exits 1 on the first difference.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../tinyimgcodec_amd/csrc/tic_math.h"

using namespace tic;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static long n_checked = 0;
static int check(double a, double b, double y) {
    const double want = a / b, got = div_rn(a, b, y);
    n_checked++;
    if (memcmp(&want, &got, 8) != 0 || rint(want) != rint(got)) {
        printf("MISMATCH a = %a b = %a: a / b = %a, div_rn = %a\n", a, b, want, got);
        return 1;
    }
    return 0;
}
static int sweep(double b) {
    const double y = 1.0 / b;
    int bad = 0;
    for (int k = -1100; k <= 1100 && !bad; k++) {
        double t = ((double)k + 0.5) * b;
        bad |= check(t, b, y);
        double up = t, dn = t;
        for (int j = 0; j < 8; j++) {
            up = nextafter(up, INFINITY);
            dn = nextafter(dn, -INFINITY);
            bad |= check(up, b, y);
            bad |= check(dn, b, y);
        }
    }
    for (int e = -8800; e <= 8800 && !bad; e++) bad |= check((double)e / 8.0, b, y);
    for (int r = 0; r < 3000 && !bad; r++) {
        const uint64_t v = rnd64();
        const double m = (double)(v >> 11) * (1.0 / 9007199254740992.0); // [0, 1)
        const int ex = (int)(rnd64() % 30) - 15;
        bad |= check(ldexp(m + 0.5, ex) * ((v & 1) ? 1.0 : -1.0), b, y);
    }
    return bad;
}

int main() {
    int bad = 0;
    long ndiv = 0;
    DctqConsts *c = new DctqConsts();
    for (int q = 1; q <= 99 && !bad; q++) {
        if (!build_consts(q, c)) return 2;
        for (int i = 0; i < 64 && !bad; i++) {
            if (c->rdiv[i] != 1.0 / c->div[i]) { // the reciprocal the kernel is handed must be the correctly rounded one
                printf("rdiv[%d] of quality %d is not 1 / div\n", i, q);
                return 1;
            }
            bad |= sweep(c->div[i]);
            ndiv++;
        }
    }
    for (int r = 0; r < 2000 && !bad; r++) {
        const double q = 1.0 + 98.0 * ((double)(rnd64() >> 11) * (1.0 / 9007199254740992.0));
        if (!build_consts(q, c)) return 2;
        const int rat[4] = {0, 4, 32, 36};
        for (int k = 0; k < 4 && !bad; k++) {
            bad |= sweep(c->div[rat[k]]);
            ndiv++;
        }
    }
    delete c;
    printf("div_rn == IEEE division on %ld quotients over %ld divisors: %s\n", n_checked, ndiv, bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
