"""Where a per-call tic_compress of a 512 x 512 frame spends its time, in the orders bench.py's bench_set leg and tools/bench_set_timing.py call it."""
import ctypes as C, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
px = np.load('tests/golden/benchmark_set.npz')['pixels']
n, h, w = px.shape
frames = [np.ascontiguousarray(px[i]) for i in range(n)]
cap = L.tic_compress_bound(h, w)
o1, n1 = np.empty(cap, np.uint8), C.c_size_t()
def loop(tag):
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            ctx.check(L.tic_compress(ctx.handle, frames[i].ctypes.data, h, w, w, 50, o1.ctypes.data, cap, C.byref(n1)))
        print("%s: tic_compress %.1f us per call" % (tag, (time.perf_counter() - t0) / n * 1e6), flush=True)
loop("fresh context")
pool = np.empty((n, cap), dtype=np.uint8)
inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames]); outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, 50, outp, caps, lens, 0))
loop("after tic_compress_batch")
streams = [pool[i, : lens[i]].copy() for i in range(n)]
block = np.empty((n, h, w), np.uint8)
sp = (C.c_void_p * n)(*[s.ctypes.data for s in streams]); sl = (C.c_size_t * n)(*[s.size for s in streams])
pp = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)]); pc = (C.c_size_t * n)(*([h * w] * n))
ctx.check(L.tic_decompress_batch(ctx.handle, sp, sl, n, pp, pc, None, None))
loop("after tic_decompress_batch")
s = T.compress(frames[0], 50, ctx=ctx)
loop("after T.compress")
