"""Where tic_decompress_dev's time outside its kernels goes: the synchronous 16-byte header read, the launches, the final wait.
Usage: python tools/dec_overheads.py [dim=4096] [quality=50]"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 50
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
s = np.frombuffer(T.compress(img, q, ctx=ctx), dtype=np.uint8)
d_s, d_p = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, s.size + 64, C.byref(d_s)))
ctx.check(L.tic_dev_alloc(ctx.handle, dim * dim, C.byref(d_p)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_s, s.ctypes.data, s.size))
head = np.zeros(16, np.uint8)
def timed(fn, reps=200):
    for _ in range(10): fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e6
print("16-byte device -> host copy (tic_memcpy_d2h): %.1f us" % timed(lambda: L.tic_memcpy_d2h(ctx.handle, head.ctypes.data, d_s, 16)))
print("tic_sync on an idle stream: %.1f us" % timed(lambda: L.tic_sync(ctx.handle)))
for k in range(3):
    print("tic_decompress_dev %dx%d q=%d: %.1f us" % (dim, dim, q, timed(lambda: ctx.check(L.tic_decompress_dev(ctx.handle, d_s, s.size, d_p, dim, dim * dim, None, None)), 100)))
