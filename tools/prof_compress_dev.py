"""Drives tic_compress_dev (transform + device entropy stage, frame resident in HBM) a few times: run under
rocprofv3 --kernel-trace --stats to see the per-kernel split.  Usage: python tools/prof_compress_dev.py [dim] [reps]"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import os
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
if os.environ.get('TIC_ENT_LANE'):  # the lane-per-block packing kernel (not the default)
    ctx.check(L.tic_set_entropy_lane_kernel(ctx.handle, 99))
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = w = dim
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
if os.environ.get('TIC_CONTENT') == 'lenna':  # natural content: the 512x512 Lenna pixels tiled
    img = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (dim // 512, dim // 512)))
cap = L.tic_compress_bound(h, w)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
n = C.c_size_t()
for q in (50,):
    for k in range(3):
        ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, q, d_out, cap, C.byref(n)))
    t = time.perf_counter()
    for k in range(reps):
        ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, q, d_out, cap, C.byref(n)))
    dt = (time.perf_counter() - t) / reps
    print("tic_compress_dev %dx%d q=%d: %.1f us per frame, %.1f Gpix/s, stream %d bytes" % (h, w, q, dt * 1e6, h * w / dt / 1e9, n.value))
