"""Would two HIP streams help the pipelined encoder?  Two contexts on one device, a thread each, compress resident 4096^2 frames through
tic_compress_dev_async in bursts of 64: frames per second of both together against one context alone (the kernels of the two streams may
overlap: the transform is bound by HBM, the pack kernel by vector issue).  Usage: python tools/two_context_compress.py [dim=4096] [quality=50]"""
import ctypes as C, sys, threading, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load()
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 50
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
cap = L.tic_compress_bound(dim, dim)
def setup():
    ctx = T.Context(0)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap + 64, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    return ctx, d_img, d_out
def burst(st, bursts, n=64):
    ctx, d_img, d_out = st
    for _ in range(bursts):
        tk = []
        for k in range(n):
            t = C.c_longlong()
            ctx.check(L.tic_compress_dev_async(ctx.handle, d_img, dim, dim, dim, q, d_out, cap, C.byref(t)))
            tk.append(t.value)
        nn = C.c_size_t()
        for t in tk:
            ctx.check(L.tic_async_result(ctx.handle, t, 1, C.byref(nn)))
a, b = setup(), setup()
burst(a, 2); burst(b, 2)
t0 = time.perf_counter(); burst(a, 8); t1 = time.perf_counter()
print("one context : %.1f us per frame" % ((t1 - t0) / (8 * 64) * 1e6))
th = [threading.Thread(target=burst, args=(s, 8)) for s in (a, b)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
t1 = time.perf_counter()
print("two contexts: %.1f us per frame (both together)" % ((t1 - t0) / (2 * 8 * 64) * 1e6))
