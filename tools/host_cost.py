"""Host-side cost of the C-ABI calls (asynchronous launches on a tiny frame, where the device is never the bottleneck)."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
for dim in (64, 4096):
    h = w = dim
    img = np.random.default_rng(1).integers(0, 256, (h, w), dtype=np.uint8)
    cap = L.tic_compress_bound(h, w)
    d_img, d_out, d_zz = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_zz)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    n = C.c_size_t()
    reps = 2000
    for name, fn in (("tic_dctq_dev (async launch)", lambda: L.tic_dctq_dev(ctx.handle, d_img, h, w, w, 50, d_zz, 2)),
                     ("tic_compress_dev (3 launches + wait)", lambda: L.tic_compress_dev(ctx.handle, d_img, h, w, w, 50, d_out, cap, C.byref(n))),
                     ("tic_entropy_encode_dev (2 launches + wait)", lambda: L.tic_entropy_encode_dev(ctx.handle, d_zz, h, w, 50, d_out, cap, C.byref(n))),
                     ("tic_dctq_dev + tic_sync", lambda: (L.tic_dctq_dev(ctx.handle, d_img, h, w, w, 50, d_zz, 2), L.tic_sync(ctx.handle))),
                     ("tic_num_blocks (ctypes call only)", lambda: L.tic_num_blocks(h, w))):
        for _ in range(50): fn()
        L.tic_sync(ctx.handle)
        t = time.perf_counter()
        for _ in range(reps): fn()
        L.tic_sync(ctx.handle)
        dt = (time.perf_counter() - t) / reps
        print("%5d^2 %-40s %7.2f us per call" % (dim, name, dt * 1e6), flush=True)
