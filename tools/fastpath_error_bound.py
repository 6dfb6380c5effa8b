"""Rigorous forward error bound of the float32 AAN fast path (dct8_aan in tic_math.h, rows then columns) for uint8
pixels: every float32 operation contributes u*|result|max (u = 2^-24; exact integer additions contribute nothing),
errors propagate linearly.  Prints the bound per coefficient in orthonormal-DCT units; kGuardX must exceed
max(bound) + 1024*2^-23 (quantiser multiply)."""
import numpy as np

U = 2.0 ** -24


class Node:
    def __init__(self, lin, err, lo, hi):
        self.lin, self.err, self.lo, self.hi = np.asarray(lin, float), float(err), lo, hi

    def rng(self):  # range of the exact linear functional over the input box
        a, b = self.lin * self.lo, self.lin * self.hi
        return np.minimum(a, b).sum(), np.maximum(a, b).sum()

    def maxabs(self):
        r = self.rng()
        return max(abs(r[0]), abs(r[1])) + self.err

    def _round(self, exact_int):
        if exact_int and np.allclose(self.lin, np.round(self.lin)) and self.maxabs() < 2 ** 24 and self.err == 0.0:
            return self
        self.err += U * self.maxabs()
        return self

    def __add__(self, o):
        return Node(self.lin + o.lin, self.err + o.err, self.lo, self.hi)._round(True)

    def __sub__(self, o):
        return Node(self.lin - o.lin, self.err + o.err, self.lo, self.hi)._round(True)

    def mulc(self, c):
        return Node(self.lin * c, abs(c) * self.err, self.lo, self.hi)._round(False)

    def fma(self, c, o):  # self*c + o with a single rounding
        return Node(self.lin * c + o.lin, abs(c) * self.err + o.err, self.lo, self.hi)._round(False)


def aan(d):
    c707, c382, c541, c1306 = 0.70710678118654752440, 0.38268343236508977173, 0.54119610014619698440, 1.30656296487637652786
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


def bound_matrix():
    """Rigorous bound per coefficient [u][v] (orthonormal-DCT units) of the float32 AAN fast path, rows then columns."""
    eye = np.eye(8)
    row = aan([Node(eye[k], 0.0, np.zeros(8), np.full(8, 255.0)) for k in range(8)])  # pass 1: one pixel row
    k = np.arange(8)
    aansc = np.where(k == 0, 1.0, np.cos(k * np.pi / 16) * np.sqrt(2))
    bound = np.zeros((8, 8))
    for v in range(8):
        lo, hi = row[v].rng()
        if v == 0:  # level shift: exact integer subtraction of 1024
            lo, hi = lo - 1024.0, hi - 1024.0
        e1 = row[v].err
        # pass 2 down the column: 8 inputs = output v of 8 different rows: box [lo,hi]^8, each carrying error e1
        col = aan([Node(eye[r], e1, np.full(8, lo), np.full(8, hi)) for r in range(8)])
        for u in range(8):
            bound[u, v] = col[u].err / (aansc[u] * aansc[v] * 8.0)
    return bound


QUANT_MUL = 1024 * 2.0 ** -23  # rounding of the float32 quantiser multiply, in coefficient units (|X| <= 1024)


def main():
    bound = bound_matrix()
    np.set_printoptions(linewidth=140)
    print("rigorous bound per coefficient (x1e-4, orthonormal units; rows u, columns v):")
    print(np.round(bound * 1e4, 2))
    q = QUANT_MUL
    print("max bound %.3e + quantiser multiply %.3e = %.3e  (kGuardX = 1.0e-3)" % (bound.max(), q, bound.max() + q))


if __name__ == "__main__":
    main()
