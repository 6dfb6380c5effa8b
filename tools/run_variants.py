"""Launches each timing variant of the hybrid kernel a few times (for rocprofv3 --pmc passes: the template instances have
distinct kernel names).  Usage: python tools/run_variants.py [dim] [iters]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
h = w = dim
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
ms = C.c_float()
for v in (2, 12, 15, 10):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, v, iters, C.byref(ms)))
    print(v, ms.value * 1e3 / iters, "us")
