"""Drives tic_decompress_dev_async with four tickets open (a 4096^2 stream, resident): run under rocprofv3 --kernel-trace to see the two
kernels of neighbouring frames overlap on the device.  Usage: python tools/prof_decompress_async.py [frames=64] [quality=50]"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dim = 4096
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
s = np.frombuffer(T.compress(img, q, ctx=ctx), dtype=np.uint8)
d_s = C.c_void_p(); ctx.check(L.tic_dev_alloc(ctx.handle, s.size + 64, C.byref(d_s)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_s, s.ctypes.data, s.size))
d_p = []
for k in range(4):
    p = C.c_void_p(); ctx.check(L.tic_dev_alloc(ctx.handle, dim * dim, C.byref(p))); d_p.append(p)
for k in range(3):  # two equal headers in a row: from the third call on the launches go out on the guess
    ctx.check(L.tic_decompress_dev(ctx.handle, d_s, s.size, d_p[0], dim, dim * dim, None, None))
def burst():
    tk = []
    for k in range(frames):
        if len(tk) == 4:
            ctx.check(L.tic_decompress_async_result(ctx.handle, tk.pop(0), 1, None, None))
        t = C.c_longlong()
        ctx.check(L.tic_decompress_dev_async(ctx.handle, d_s, s.size, d_p[k % 4], dim, dim * dim, C.byref(t)))
        tk.append(t.value)
    while tk:
        ctx.check(L.tic_decompress_async_result(ctx.handle, tk.pop(0), 1, None, None))
burst()
t0 = time.perf_counter(); burst(); dt = time.perf_counter() - t0
print("tic_decompress_dev_async, 4 tickets open: %.1f us per frame over %d frames" % (dt / frames * 1e6, frames))
