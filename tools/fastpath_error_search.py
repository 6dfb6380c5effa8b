"""Adversarial search for the largest error of the float32 AAN fast path against the float64 orthonormal DCT.

Emulates dct8_aan() of tic_math.h in float32 (FMAs through float64, which holds the product exactly) with the pass
order of the hybrid kernel (rows first), for pixel values in 0..255, and hill-climbs per output coefficient on the
absolute error in coefficient units.  The guard band kGuardX must exceed the result plus the quantiser's
1024 * 2^-23.  Run: python tools/fastpath_error_search.py [seconds]"""
import sys
import time

import numpy as np
from scipy.fftpack import dct

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f32)


def aan(d):
    d0, d1, d2, d3, d4, d5, d6, d7 = [d[..., k] for k in range(8)]
    c707, c382, c541, c1306 = f32(0.70710678118654752440), f32(0.38268343236508977173), f32(0.54119610014619698440), f32(1.30656296487637652786)
    t0, t7, t1, t6 = d0 + d7, d0 - d7, d1 + d6, d1 - d6
    t2, t5, t3, t4 = d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o0, o4 = t10 + t11, t10 - t11
    s = t12 + t13
    o2, o6 = fma(s, c707, t13), fma(s, -c707, t13)
    u10, u11, u12 = t4 + t5, t5 + t6, t6 + t7
    z5 = (u10 - u12) * c382
    z2, z4 = fma(u10, c541, z5), fma(u12, c1306, z5)
    z11, z13 = fma(u11, c707, t7), fma(u11, -c707, t7)
    return np.stack([o0, z11 + z4, o2, z13 - z2, o4, z13 + z2, o6, z11 - z4], -1).astype(f32)


k = np.arange(8)
aansc = np.where(k == 0, 1.0, np.cos(k * np.pi / 16) * np.sqrt(2))
scale2 = 1.0 / (np.outer(aansc, aansc) * 8)


def errors(blocks):  # blocks uint8-valued float arrays [n,8,8]
    x = blocks.astype(f32)
    y = aan(x)  # along rows (last axis)
    y[..., 0] -= f32(1024.0)
    z = aan(y.swapaxes(-1, -2)).swapaxes(-1, -2)  # down the columns
    fast = z.astype(np.float64) * scale2
    ref = dct(dct(blocks.astype(np.float64) - 128.0, norm="ortho", axis=-2), norm="ortho", axis=-1)
    return np.abs(fast - ref)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    save = sys.argv[2] if len(sys.argv) > 2 else None  # .npz: the worst block found for every coefficient (+ runners-up)
    rng = np.random.default_rng(1)
    best = np.zeros((8, 8))
    worst_blocks = np.zeros((64, 4, 8, 8), np.uint8)  # per coefficient: the 4 blocks with the largest error so far
    worst_err = np.zeros((64, 4))
    t_end = time.time() + budget
    pop = rng.choice([0.0, 255.0], (4096, 8, 8))
    it = 0
    while time.time() < t_end:
        e = errors(pop)
        best = np.maximum(best, e.max(0))
        ef = e.reshape(len(pop), 64)
        for c in range(64):
            k = int(np.argmax(ef[:, c]))
            j = int(np.argmin(worst_err[c]))
            if ef[k, c] > worst_err[c, j] and not any(np.array_equal(pop[k].astype(np.uint8), wb) for wb in worst_blocks[c]):
                worst_err[c, j] = ef[k, c]
                worst_blocks[c, j] = pop[k].astype(np.uint8)
        # keep the blocks that are best for any coefficient, mutate them (flip to extremes / random values)
        keep_idx = np.unique(np.argsort(-e.reshape(len(pop), 64), axis=0)[:24].ravel())
        keep = pop[keep_idx]
        kids = np.repeat(keep, max(1, 4096 // len(keep) - 1), axis=0)
        mask = rng.random(kids.shape) < 0.04
        vals = np.where(rng.random(kids.shape) < 0.7, rng.choice([0.0, 255.0], kids.shape), rng.integers(0, 256, kids.shape).astype(float))
        kids = np.where(mask, vals, kids)
        pop = np.concatenate([keep, kids])[:8192]
        it += 1
    print("iterations", it, "max error over all coefficients: %.3e" % best.max())
    print("per-coefficient max (x1e-4):")
    print(np.round(best * 1e4, 2))
    print("with the quantiser multiply (1024*2^-23 = 1.22e-4): %.3e  -> guard band 1.0e-3 margin x%.2f" % (
        best.max() + 1.22e-4, 1e-3 / (best.max() + 1.22e-4)))
    if save:
        np.savez_compressed(save, blocks=worst_blocks.reshape(-1, 8, 8), err=worst_err.reshape(-1))
        print("saved", worst_blocks.reshape(-1, 8, 8).shape, "blocks to", save)


if __name__ == "__main__":
    main()
