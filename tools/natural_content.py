"""Kernel time on natural content: the 512x512 Lenna pixels tiled to 4096^2 (and a smoothed, a flat and a gradient frame) against
the random frame of the benchmark.  Natural images have flat and smooth blocks, i.e. many exact ties of the rational coefficients."""
import ctypes as C, os, sys
SETTINGS = None
if "--settings" in sys.argv:  # python tools/natural_content.py --settings "TIC_ROT=0" "TIC_ROT=1;TIC_SPLIT=..." : the hooks build, one column per setting
    k = sys.argv.index("--settings"); SETTINGS = [dict(kv.split("=") for kv in a.split(";") if kv) for a in sys.argv[k + 1:]]; del sys.argv[k:]
    os.environ["TIC_TEST_HOOKS"] = "1"; os.environ["TIC_TUNE"] = "1"
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
# python tools/natural_content.py [other.so ...]: further builds of the library are timed in the same process (boxes differ by +-5 %)
libs = [("product", N.load())]
for pth in sys.argv[1:]:
    Lo = C.CDLL(pth)
    for name in ("tic_create", "tic_dev_alloc", "tic_dev_free", "tic_memcpy_h2d", "tic_dctq_dev_timed"):
        res, a = N.SIGNATURES[name]
        fn = getattr(Lo, name); fn.restype = res; fn.argtypes = a
    libs.append((pth.split("/")[-1], Lo))
L = N.load(); ctx = T.Context(0)
others = [(nm, Lo, Lo.tic_create(0)) for nm, Lo in libs[1:]]
lenna = np.load('tests/golden/lenna.npz')['img']
dim = 4096
frames = {
    "random (benchmark)": np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8),
    "lenna tiled 8x8": np.tile(lenna, (8, 8)),
    "lenna upscaled 8x (smooth)": np.kron(lenna, np.ones((8, 8), np.uint8)),
    "flat 201": np.full((dim, dim), 201, np.uint8),
    "horizontal gradient": np.tile((np.arange(dim) // 16).astype(np.uint8), (dim, 1)),
    "lenna tiled, quantised to 32 levels": (np.tile(lenna, (8, 8)) // 8 * 8 + 1).astype(np.uint8),
    "lenna upscaled 8x, 32 levels (flat blocks)": np.kron((lenna // 8 * 8 + 1).astype(np.uint8), np.ones((8, 8), np.uint8)),
    "checkerboard 100/102 (dense ties, not flat)": np.where(np.add.outer(np.arange(dim), np.arange(dim)) % 2 == 0, 100, 102).astype(np.uint8),
}
for name, img in frames.items():
    img = np.ascontiguousarray(img)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    ms = C.c_float()
    for q in (50, 90):
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, dim, dim, dim, q, d_out, 2, 3000, C.byref(ms)))
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, dim, dim, dim, q, d_out, 2, 2000, C.byref(ms)))
        t = ms.value * 1e3 / 2000
        L.tic_set_stats(ctx.handle, 1)  # (the statistics launch carries a diagnostic atomic: never timed)
        ctx.check(L.tic_dctq_dev(ctx.handle, d_img, dim, dim, dim, q, d_out, 2))
        st = (C.c_ulonglong * 4)()
        L.tic_last_rare_path_stats(ctx.handle, st)
        fb = C.c_ulonglong(st[0])
        L.tic_set_stats(ctx.handle, 0)
        n = C.c_size_t()
        extra = ""
        for st_ in SETTINGS or ():
            for k_, v_ in st_.items(): os.environ[k_] = v_
            ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, dim, dim, dim, q, d_out, 2, 3000, C.byref(ms)))
            ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, dim, dim, dim, q, d_out, 2, 2000, C.byref(ms)))
            for k_ in st_: os.environ.pop(k_)
            to = ms.value * 1e3 / 2000
            extra += "  | %s %6.2f us (%.1f %%)" % (";".join("%s=%s" % kv for kv in st_.items()), to, 3.0 * dim * dim / to / 1e3 / 80)
        for nm, Lo, co in others:
            assert Lo.tic_dctq_dev_timed(co, d_img, dim, dim, dim, q, d_out, 2, 3000, C.byref(ms)) == 0
            assert Lo.tic_dctq_dev_timed(co, d_img, dim, dim, dim, q, d_out, 2, 2000, C.byref(ms)) == 0
            to = ms.value * 1e3 / 2000
            extra += "  | %s %6.2f us (%.1f %%)" % (nm, to, 3.0 * dim * dim / to / 1e3 / 80)
        print("%-38s q=%d  kernel %6.2f us  (%.1f %% of 8 TB/s)  second-level blocks %d, tie strips %d, exact strips %d, largest batch %d%s" % (name, q, t, 3.0 * dim * dim / t / 1e3 / 80, fb.value, st[1], st[2], st[3], extra), flush=True)
    L.tic_dev_free(ctx.handle, d_img); L.tic_dev_free(ctx.handle, d_out)
