"""Kernel efficiency against frame width at a constant pixel count (dense rows: pitch = width): tic_dctq_dev_timed on random frames.
Usage: python tools/width_sweep.py [pixels_M=134] [w_lo=1024] [w_hi=4096] [step=64]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
px = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 134_000_000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
st = int(sys.argv[4]) if len(sys.argv) > 4 else 64
pitch_extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0
buf = np.random.default_rng(1).integers(0, 256, px + (1 << 22), dtype=np.uint8)
d_in, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, buf.size + (1 << 26), C.byref(d_in)))
ctx.check(L.tic_dev_alloc(ctx.handle, 2 * buf.size, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_in, buf.ctypes.data, buf.size))
ms = C.c_float()
for w in range(lo, hi + 1, st):
    pitch = w + pitch_extra
    h = px // pitch // 8 * 8
    for it in (10, 30):
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_in, h, w, pitch, 50, d_out, 2, it, C.byref(ms)))
    us = ms.value * 1e3 / 30
    print("w %5d (%3d strips/row, pitch %5d) h %7d: %8.2f us  %.3f of 8 TB/s" % (w, w // 64, pitch, h, us, 3.0 * h * w / (us * 1e-6) / 8e12), flush=True)
