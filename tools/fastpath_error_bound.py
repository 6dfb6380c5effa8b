"""The guard band of the float32 fast path as a theorem: a complete forward error bound of what the strip kernel
computes (dct8_aan<float> of tic_math.h along the pixel rows, level shift, dct8_aan<float> down the columns, fused
quantiser) against what the reference computes (float64 pocketfft order, IEEE divide; utils.py:32-37, 48-53), for
every block of uint8 pixels.  kGuard[u][v] of tic_math.h must be >= guard_matrix()[u][v]; tests/test_host_cpu.py asserts it.

Terms, all in coefficient (orthonormal DCT) units unless stated, for coefficient (u,v):

  systematic  the kernel multiplies by float32(0.7071...), float32(0.3827...), float32(0.5412...), float32(1.3066...),
              not by the real numbers.  The float-constant algorithm in exact arithmetic is a linear map F of the 64
              pixels, the true scaled DCT a linear map T; max over the pixel box of |(F - T)(x)| is computed EXACTLY
              (F in rational arithmetic from the float32 bit patterns, T in 60-digit arithmetic): no triangle
              inequality over operations.  (Round 2 omitted this term: VERDICT r02, weak #1.)
  rounding    every float32 operation whose result is not provably exact contributes half an ulp of the binade of the
              largest magnitude its result can take (<= 2^-24 |result|max), carried to the outputs through the absolute
              values of the float32 constants downstream.  Integer-valued sums of integers below 2^24 are exact.
  multiplier  the quantiser multiplies by float32(1 / (aan_u aan_v 8 div)): relative error 2^-24, i.e. |X|max 2^-24.
              The product itself is not rounded (it lives inside the fused multiply-adds of quant_fma).
  reference   the reference's own float64 DCT (pocketfft order, same analysis with 2^-53) and its rounded quotient.

What the accept test itself loses to float32 (the threshold and the distance d are floats) is not part of the band: it is
taken out of the threshold in quantised units by build_consts() (tic_math.h: the threshold is rounded DOWN and lowered
by one more float, which covers the rounding of d: a distance in [0.25, 0.5) is rounded by at most 2^-26).

Run: python tools/fastpath_error_bound.py            prints the terms and the C table
"""
import math
from fractions import Fraction

import numpy as np

U32 = 2.0 ** -24
U64 = 2.0 ** -53
MARGIN = 1.02  # over the complete bound (covers the float64 evaluation of this script itself, ~1e-15 relative)

C_REAL = (0.70710678118654752440, 0.38268343236508977173, 0.54119610014619698440, 1.30656296487637652786)
C_F32 = tuple(float(np.float32(c)) for c in C_REAL)  # what `(float)0.7071...` is in the kernel


def half_ulp(m, unit):
    """Largest rounding error of a result whose magnitude is at most m: half an ulp of m's binade."""
    if m <= 0.0:
        return 0.0
    return 2.0 ** math.floor(math.log2(m)) * unit


class Node:
    """A value of the computation: exact linear functional `lin` of the pass inputs (with the constants the
    arithmetic really uses), accumulated absolute error bound `err`, input box [lo, hi]."""

    def __init__(self, lin, err, lo, hi, unit, int_inputs):
        self.lin, self.err, self.lo, self.hi = np.asarray(lin, float), float(err), lo, hi
        self.unit, self.int_inputs = unit, int_inputs

    def rng(self):
        a, b = self.lin * self.lo, self.lin * self.hi
        return np.minimum(a, b).sum(), np.maximum(a, b).sum()

    def maxabs(self):
        r = self.rng()
        return max(abs(r[0]), abs(r[1])) + self.err

    def _new(self, lin, err):
        return Node(lin, err, self.lo, self.hi, self.unit, self.int_inputs)

    def _round(self, may_be_exact):
        limit = 2.0 ** 24 if self.unit == U32 else 2.0 ** 53
        if (may_be_exact and self.int_inputs and self.err == 0.0 and np.array_equal(self.lin, np.round(self.lin))
                and self.maxabs() < limit):
            return self  # an integer below 2^24 (2^53): representable, the operation is exact
        self.err += half_ulp(self.maxabs(), self.unit)
        return self

    def __add__(self, o):
        return self._new(self.lin + o.lin, self.err + o.err)._round(True)

    def __sub__(self, o):
        return self._new(self.lin - o.lin, self.err + o.err)._round(True)

    def scale2(self, p):  # multiplication by a power of two: exact
        return self._new(self.lin * p, self.err * abs(p))

    def mulc(self, c):
        return self._new(self.lin * c, abs(c) * self.err)._round(False)

    def fma(self, c, o):  # self*c + o with a single rounding
        return self._new(self.lin * c + o.lin, abs(c) * self.err + o.err)._round(False)


def aan(d, consts):
    """dct8_aan of tic_math.h on Nodes (or on anything with + - mulc fma)."""
    c707, c382, c541, c1306 = consts
    t0, t7, t1, t6 = d[0] + d[7], d[0] - d[7], d[1] + d[6], d[1] - d[6]
    t2, t5, t3, t4 = d[2] + d[5], d[2] - d[5], d[3] + d[4], d[3] - d[4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o0, o4 = t10 + t11, t10 - t11
    s = t12 + t13
    o2, o6 = s.fma(c707, t13), s.fma(-c707, t13)
    u10, u11, u12 = t4 + t5, t5 + t6, t6 + t7
    z5 = (u10 - u12).mulc(c382)
    z2, z4 = u10.fma(c541, z5), u12.fma(c1306, z5)
    z11, z13 = u11.fma(c707, t7), u11.fma(-c707, t7)
    return [o0, z11 + z4, o2, z13 - z2, o4, z13 + z2, o6, z11 - z4]


K = np.arange(8)
AANSC = np.where(K == 0, 1.0, np.cos(K * np.pi / 16) * np.sqrt(2))  # output k of aan() = orthonormal coefficient * AANSC[k] * sqrt(8)
SCALE2 = np.outer(AANSC, AANSC) * 8.0


def rounding_bound():
    """[u][v], scaled (Z) units: accumulated float32 rounding of the two passes, rows first."""
    eye = np.eye(8)
    row = aan([Node(eye[k], 0.0, np.zeros(8), np.full(8, 255.0), U32, True) for k in range(8)], C_F32)
    out = np.zeros((8, 8))
    for v in range(8):
        lo, hi = row[v].rng()
        if v == 0:  # level shift: exact integer subtraction of 1024
            lo, hi = lo - 1024.0, hi - 1024.0
        e1 = row[v].err
        # pass 2 down the column: its 8 inputs are output v of 8 different pixel rows: box [lo,hi]^8, each carrying error e1;
        # they are integers exactly when e1 == 0 (v = 0, 4: plain sums and differences of the pixels)
        col = aan([Node(eye[r], e1, np.full(8, lo), np.full(8, hi), U32, e1 == 0.0) for r in range(8)], C_F32)
        for u in range(8):
            out[u, v] = col[u].err
    return out


class _Lin:
    """Exact linear functional over the rationals (the float-constant algorithm in exact arithmetic)."""

    def __init__(self, v):
        self.v = v

    def __add__(self, o):
        return _Lin([a + b for a, b in zip(self.v, o.v)])

    def __sub__(self, o):
        return _Lin([a - b for a, b in zip(self.v, o.v)])

    def mulc(self, c):
        return _Lin([a * c for a in self.v])

    def fma(self, c, o):
        return _Lin([a * c + b for a, b in zip(self.v, o.v)])


def systematic_bound(want_blocks=False):
    """[u][v], scaled (Z) units: max over pixels in [-128,127]^64 of |F(x) - T(x)|, F = the algorithm with its float32
    constants in exact arithmetic, T = the true scaled DCT.  (For v != 0 the rows of F sum to zero exactly - asserted -
    so the kernel's `pass 1 on 0..255, then subtract 1024 from output 0` equals F on pixel - 128.)"""
    import mpmath as mp

    mp.mp.dps = 60
    cf = tuple(Fraction(c) for c in C_F32)  # float -> exact rational
    F1 = aan([_Lin([Fraction(int(k == n)) for n in range(8)]) for k in range(8)], cf)
    F1 = [f.v for f in F1]  # F1[k][n]
    assert all(f == 1 for f in F1[0]) and all(sum(F1[k]) == 0 for k in range(1, 8))
    T1 = [[(mp.mpf(1) if k == 0 else mp.sqrt(2) * mp.cos(k * mp.pi / 16) * mp.sqrt(8) * mp.mpf(1) / 2 * mp.cos((2 * n + 1) * k * mp.pi / 16))
           for n in range(8)] for k in range(8)]
    F1m = [[mp.mpf(f.numerator) / mp.mpf(f.denominator) for f in r] for r in F1]
    out = np.zeros((8, 8))
    blocks = np.zeros((8, 8, 8, 8), np.uint8)  # [u][v]: a pixel block that attains the maximum
    for u in range(8):
        for v in range(8):
            pos = neg = mp.mpf(0)
            bp, bn = np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8)
            for r in range(8):
                for c in range(8):
                    d = F1m[u][r] * F1m[v][c] - T1[u][r] * T1[v][c]
                    pos += max(d * 127, d * -128)
                    neg += min(d * 127, d * -128)
                    bp[r, c] = 255 if d > 0 else 0
                    bn[r, c] = 0 if d > 0 else 255
            out[u, v] = float(max(pos, -neg))
            blocks[u, v] = bp if pos >= -neg else bn
    return (out, blocks) if want_blocks else out


def reference_bound():
    """[u][v], coefficient units: error of the reference's own float64 DCT (dct8_exact's operation graph, columns first;
    SURVEY Appendix A) plus the rounding of its quotient X/div, expressed in coefficient units."""
    WR, WI = float.fromhex("0x1.6a09e667f3bccp-1"), float.fromhex("0x1.6a09e667f3bcdp-1")
    TW = [float.fromhex(h) for h in ("0x1.f6297cff75cb0p-1", "0x1.d906bcf328d46p-1", "0x1.a9b66290ea1a3p-1", "0x1.6a09e667f3bccp-1",
                                     "0x1.1c73b39ae68c8p-1", "0x1.87de2a6aea963p-2", "0x1.8f8b83c69a60ap-3")]
    SQ2H = WI

    def exact8(c):
        c = list(c)
        c[0], c[7] = c[0].scale2(2.0), c[7].scale2(2.0)
        for k in (1, 3, 5):
            t = c[k + 1]
            c[k + 1] = t - c[k]
            c[k] = c[k] + t
        h = [None] * 8
        h[0], h[4] = c[0] + c[7], c[0] - c[7]
        h[3], h[7] = c[3].scale2(2.0), c[4].scale2(-2.0)
        h[1], tr2 = c[1] + c[5], c[1] - c[5]
        ti2, h[2] = c[2] + c[6], c[2] - c[6]
        h[6] = ti2.mulc(WR) + tr2.mulc(WI)
        h[5] = tr2.mulc(WR) - ti2.mulc(WI)
        o = [None] * 8
        for k in (0, 1):
            tr2_, tr1 = h[4 * k] + h[4 * k + 3], h[4 * k] - h[4 * k + 3]
            tr3, tr4 = h[4 * k + 1].scale2(2.0), h[4 * k + 2].scale2(2.0)
            o[k], o[k + 4] = tr2_ + tr3, tr2_ - tr3
            o[k + 6], o[k + 2] = tr1 + tr4, tr1 - tr4
        c = [x.scale2(0.25) for x in o]
        for k, kc in ((1, 7), (2, 6), (3, 5)):
            t1 = c[kc].mulc(TW[k - 1]) + c[k].mulc(TW[kc - 1])
            t2 = c[k].mulc(TW[k - 1]) - c[kc].mulc(TW[kc - 1])
            c[k], c[kc] = (t1 + t2).scale2(0.5), (t1 - t2).scale2(0.5)
        c[4] = c[4].mulc(TW[3])
        c[0] = c[0].mulc(SQ2H)
        return c

    eye = np.eye(8)
    col = exact8([Node(eye[k], 0.0, np.full(8, -128.0), np.full(8, 127.0), U64, True) for k in range(8)])
    out = np.zeros((8, 8))
    for u in range(8):
        lo, hi = col[u].rng()
        e1 = col[u].err
        # (the float64 constants differ from the real cosines by < 2^-53 relative: charged as one more rounding of the result)
        e1 += half_ulp(max(abs(lo), abs(hi)), U64) * 4
        rowp = exact8([Node(eye[k], e1, np.full(8, lo), np.full(8, hi), U64, False) for k in range(8)])
        for v in range(8):
            out[u, v] = rowp[v].err + half_ulp(rowp[v].maxabs(), U64) * 4 + 1024.0 * U64  # + constants, + the quotient's rounding
    return out


XMAX = 1024.0  # |X[u][v]| <= 8 * 128 for every coefficient


def terms():
    sysb = systematic_bound() / SCALE2
    rnd = rounding_bound() / SCALE2
    mul = np.full((8, 8), XMAX * U32)
    ref = reference_bound()
    return {"systematic": sysb, "rounding": rnd, "multiplier": mul, "reference": ref}


def bound_matrix():
    """The complete bound per coefficient [u][v], coefficient units (without margin)."""
    t = terms()
    # (the float32 multiplier also scales the fast path's own error by 1 + 2^-24)
    return (t["systematic"] + t["rounding"]) * (1.0 + U32) + t["multiplier"] + t["reference"]


def guard_matrix():
    """What kGuard must be at least: the complete bound times MARGIN."""
    return bound_matrix() * MARGIN


def c_table(g):
    """kGuard as C source: five significant digits, rounded UP."""
    lines = []
    for u in range(8):
        vals = []
        for v in range(8):
            e = math.floor(math.log10(g[u, v]))
            m = math.ceil(g[u, v] / 10.0 ** (e - 4)) * 10.0 ** (e - 4)
            vals.append("%.4e" % m)
        lines.append("    " + ", ".join(vals) + ",")
    return "\n".join(lines)


def main():
    np.set_printoptions(linewidth=150, precision=3, suppress=False)
    t = terms()
    for k, v in t.items():
        print("%s (x1e-4, rows u, columns v):" % k)
        print(np.round(v * 1e4, 4))
    b = bound_matrix()
    print("complete bound (x1e-4):")
    print(np.round(b * 1e4, 3))
    print("share of the systematic term: %.1f %% ... %.1f %%" % (100 * (t["systematic"] / b).min(), 100 * (t["systematic"] / b).max()))
    print("static constexpr double kGuard[64] = {  // = complete bound x %.2f (tools/fastpath_error_bound.py)" % MARGIN)
    print(c_table(guard_matrix()))
    print("};")


if __name__ == "__main__":
    main()
