"""Same-box, same-clock comparison of frame shapes of EQUAL input footprint (h x pitch ~ 265 MB, at the edge of the 256 MiB Infinity Cache;
the config-4 shard at 531 MB): rounds interleaved, long runs.  (Round 4 listed the padded shape with the dense shape's height: 283 MB against
265 MB - the 0.65 it read for pitch 2048 was the cache footprint, profiles/r05_geometry.txt.)"""
import ctypes as C, statistics, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
shapes = [(138240, 1920, 1920), (129600, 1920, 2048), (129600, 2048, 2048), (16384, 16384, 16384), (276480, 1920, 1920), (64800, 4096, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
big = max(h * p for h, w, p in shapes)
buf = np.random.default_rng(1).integers(0, 256, big + 4096, dtype=np.uint8)
d_in, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, buf.size, C.byref(d_in)))
ctx.check(L.tic_dev_alloc(ctx.handle, 2 * buf.size, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_in, buf.ctypes.data, buf.size))
ms = C.c_float()
def run(h, w, p, it):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_in, h, w, p, 50, d_out, 2, it, C.byref(ms)))
    return ms.value * 1e3 / it
for s in shapes: run(*s, 100)
res = {s: [] for s in shapes}
for r in range(5):
    for s in shapes:
        res[s].append(run(*s, 150))
for (h, w, p), v in res.items():
    us = statistics.median(v)
    print("%7d x %5d (pitch %5d): median %8.2f us  min %8.2f  max %8.2f   %.3f of 8 TB/s" % (h, w, p, us, min(v), max(v), 3.0 * h * w / (us * 1e-6) / 8e12), flush=True)
