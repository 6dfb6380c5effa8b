"""Stress run: many back-to-back launches of the hybrid kernel on changing frames, each compared with the exact kernel
(catches timing-dependent errors such as a wrong hand-counted vmcnt, which a single launch may not show)."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(2026)
shapes = [(4096, 4096), (1080, 1920), (2560, 4160), (8192, 8192), (512, 512)]
bad = 0
t0 = time.time()
for it in range(iters):
    h, w = shapes[it % len(shapes)]
    q = int(rng.integers(1, 100))
    kind = it % 5
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:  # smooth content: many exact ties
        img = (np.add.outer(np.arange(h), np.arange(w)) // int(rng.integers(1, 9)) % 256).astype(np.uint8)
    elif kind == 2:  # two-level noise
        img = (rng.integers(0, 2, (h, w), dtype=np.uint8) * int(rng.integers(1, 256))).astype(np.uint8)
    elif kind == 3:  # mosaic of flat blocks (every grey level) with noise patches: the flat-block table of the overflow path
        cell = int(rng.choice([8, 8, 16, 24]))
        lv = rng.integers(0, 256, ((h + cell - 1) // cell, (w + cell - 1) // cell), dtype=np.uint8)
        img = np.kron(lv, np.ones((cell, cell), np.uint8))[:h, :w]
        noisy = np.kron(rng.random(((h + 63) // 64, (w + 63) // 64)) < 0.2, np.ones((64, 64), bool))[:h, :w]
        img = np.where(noisy, rng.integers(0, 256, (h, w), dtype=np.uint8), img).astype(np.uint8)
    else:            # posterised ramp + two-level checkerboard cells: dense rational ties, flat and not
        step = int(rng.integers(2, 33))
        img = ((np.add.outer(np.arange(h) // 3, np.arange(w) // 5) % 256) // step * step + int(rng.integers(0, 2))).astype(np.uint8)
        chk = (np.add.outer(np.arange(h), np.arange(w)) % 2 == 0) & (np.add.outer(np.arange(h) // 32, np.arange(w) // 32) % 3 == 0)
        img = np.where(chk, np.uint8(int(rng.integers(0, 256))), img).astype(np.uint8)
    n = L.tic_num_blocks(h, w)
    d_img, d_a, d_b = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * 128, C.byref(d_a)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * 128, C.byref(d_b)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    ms = C.c_float()
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, q, d_a, N.KERNEL_HYBRID, 5, C.byref(ms)))  # back to back
    ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, w, q, d_b, N.KERNEL_EXACT))
    a = np.empty((n, 64), np.int16); b = np.empty((n, 64), np.int16)
    ctx.check(L.tic_memcpy_d2h(ctx.handle, a.ctypes.data, d_a, n * 128))
    ctx.check(L.tic_memcpy_d2h(ctx.handle, b.ctypes.data, d_b, n * 128))
    if not np.array_equal(a, b):
        bad += 1
        print("MISMATCH iter %d shape %s q %d kind %d: %d coefficients differ" % (it, (h, w), q, kind, int((a != b).sum())), flush=True)
    for p in (d_img, d_a, d_b):
        L.tic_dev_free(ctx.handle, p)
    if it % 20 == 19:
        print("iter %d ok so far, %.0f s" % (it + 1, time.time() - t0), flush=True)
print("stress_parity: %d iterations, %d mismatches" % (iters, bad))
sys.exit(1 if bad else 0)
