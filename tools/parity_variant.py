"""Parity of internal kernel variants (through tic_dctq_dev_timed) against the exact kernel and the oracle on inputs that
exercise every rare path: random frames (ties + second level), tie-stress goldens, frames made of exact ties of rational and
of irrational coefficients (exact-order redo), ragged shapes, saturated content, large grids (chunked schedule).
Usage: python tools/parity_variant.py 50 320 ...   (exit code 1 on any mismatch)"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _ablate  # noqa: F401  (experiment build of the library)
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N

L = N.load(); ctx = T.Context(0)
variants = [int(v) for v in sys.argv[1:]] or [50]


def true_tie_block():
    """X(2,2) = (2(P+Q) + sqrt2 (P-Q+R))/16 with P-Q+R = 0 and (P+Q)/128 = 0.5: an exact tie of an irrational coefficient at q=50."""
    blk = np.full((8, 8), 128, np.int32)
    blk[0, 0] += 64
    blk[0, 1] -= 64
    return blk.astype(np.uint8)


def frames():
    rng = np.random.default_rng(2024)
    yield "rand 4096^2", np.random.default_rng(1234).integers(0, 256, (4096, 4096), dtype=np.uint8), (10, 50, 90)
    yield "rand 1080x1920", np.random.default_rng(99).integers(0, 256, (1080, 1920), dtype=np.uint8), (10, 50, 90)
    yield "rand ragged 1999x4171", np.random.default_rng(9).integers(0, 256, (1999, 4171), dtype=np.uint8), (50,)
    yield "rand 8x4096", np.random.default_rng(3).integers(0, 256, (8, 4096), dtype=np.uint8), (50,)
    yield "rand 4096x8192 (chunked)", np.random.default_rng(6).integers(0, 256, (4096, 8192), dtype=np.uint8), (50,)
    g = np.load(os.path.join("tests", "golden", "tie_blocks.npz"))["img"].astype(np.uint8)
    yield "tie goldens", g, (50, 90)
    yield "tie goldens tiled 2048^2", np.tile(g, (16, 16)), (50, 37)
    yield "constant 129 (DC tie in every block)", np.full((1024, 2048), 129, np.uint8), (50,)
    tt = np.tile(true_tie_block(), (128, 256))
    yield "irrational true tie in every block", tt, (50,)
    mix = np.random.default_rng(5).integers(0, 256, (1024, 2048), dtype=np.uint8)
    for k in range(0, 128 * 256, 37):
        by, bx = divmod(k, 256)
        mix[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = true_tie_block()
    yield "random + scattered irrational true ties", mix, (50,)
    yield "saturated 0/255", rng.choice(np.array([0, 255], dtype=np.uint8), (512, 1024)), (5, 50, 99)
    yield "checker", np.tile(np.array([[0, 255], [255, 0]], dtype=np.uint8), (256, 512)), (50, 99)


bad = 0
try:
    from oracle import pyoracle
    pyoracle.build()
except Exception as e:  # noqa: BLE001
    pyoracle = None
    print("oracle unavailable:", e)
for name, img, quals in frames():
    h, w = img.shape
    pitch = (w + 255) // 256 * 256
    host = np.zeros((h, pitch), np.uint8); host[:, :w] = img
    n = L.tic_num_blocks(h, w)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * 128, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, host.ctypes.data, host.size))
    ms = C.c_float()

    def run(q, v):
        ctx.check(L.tic_memset_dev(ctx.handle, d_out, 0x5A, n * 128))
        if v == 1:
            ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, pitch, q, d_out, N.KERNEL_EXACT))
        else:
            ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, v, 1, C.byref(ms)))
        zz = np.empty((n, 64), np.int16)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, zz.ctypes.data, d_out, n * 128))
        return zz

    for q in quals:
        ref = run(q, 1)
        if pyoracle is not None and h * w <= 2048 * 2048:
            if not np.array_equal(ref, pyoracle.encode_zz16(img, q)):
                bad += 1; print("EXACT KERNEL != ORACLE", name, q)
        for v in variants:
            got = run(q, v)
            ok = np.array_equal(got, ref)
            nb = int((got != ref).any(axis=1).sum())
            print("%-45s q=%2d variant %3d: %s" % (name, q, v, "ok" if ok else "MISMATCH in %d blocks" % nb), flush=True)
            bad += 0 if ok else 1
    L.tic_dev_free(ctx.handle, d_img); L.tic_dev_free(ctx.handle, d_out)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
