"""Strip-schedule experiment: times the hybrid kernel (full / no post-pass / streaming skeleton) on a large frame and on a
batch of 1080p frames under the schedule selected by TIC_SCHED / TIC_CHUNK (read once per process by the library)."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
tag = "sched=%s chunk=%s" % (os.environ.get("TIC_SCHED", "0"), os.environ.get("TIC_CHUNK", "-"))
dims = [int(x) for x in (sys.argv[1:] or ["16384"])]
for dim in dims:
    h = w = dim
    img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    ms = C.c_float()
    for name, v in (("full", 2), ("no post-pass", 12), ("skeleton", 15)):
        iters = 50 if dim <= 4096 else 10
        best = 1e9
        for rep in range(3):
            ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, v, iters, C.byref(ms)))
            best = min(best, ms.value * 1e3 / iters)
        print("%-18s %5d^2 %-14s %9.2f us  %7.1f GB/s" % (tag, dim, name, best, 3.0 * h * w / best / 1e3), flush=True)
    # parity of this schedule: hybrid output == exact kernel output
    got = np.empty(img.size, np.int16); ref = np.empty(img.size, np.int16)
    ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, w, 50, d_out, 2)); ctx.check(L.tic_sync(ctx.handle))
    ctx.check(L.tic_memcpy_d2h(ctx.handle, got.ctypes.data, d_out, got.nbytes))
    ctx.check(L.tic_dctq_dev(ctx.handle, d_img, h, w, w, 50, d_out, 1)); ctx.check(L.tic_sync(ctx.handle))
    ctx.check(L.tic_memcpy_d2h(ctx.handle, ref.ctypes.data, d_out, ref.nbytes))
    print("%-18s %5d^2 hybrid == exact kernel: %s" % (tag, dim, bool(np.array_equal(got, ref))), flush=True)
    del got, ref
    L.tic_dev_free(ctx.handle, d_img); L.tic_dev_free(ctx.handle, d_out)
# batch of 1080p frames, one launch
nf, h, w = 256, 1080, 1920
frames = np.random.default_rng(7).integers(0, 256, (nf, h, w), dtype=np.uint8)
nblk = L.tic_num_blocks(h, w)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, frames.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, nf * nblk * 128, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, frames.ctypes.data, frames.size))
import time
for rep in range(3):
    ctx.check(L.tic_sync(ctx.handle))
    t = time.perf_counter()
    for k in range(5):
        ctx.check(L.tic_dctq_dev_frames(ctx.handle, d_img, nf, h, w, w, h * w, 50, d_out, nblk * 128, 2))
    ctx.check(L.tic_sync(ctx.handle))
    dt = (time.perf_counter() - t) / 5
    print("%-18s 256x1080p batched launch %9.1f us  %7.1f GB/s" % (tag, dt * 1e6, 3.0 * nf * h * w / dt / 1e9), flush=True)
