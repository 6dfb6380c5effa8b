"""Does the cold number depend on how the rotating buffers were allocated?  12 frame / coefficient pairs of 4096^2 (604 MB) as 24 separate
device allocations (what bench.py did up to round 3) against ONE allocation carved at 2 MB-aligned offsets, rounds interleaved, one process."""
import ctypes as C, statistics, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h = w = 4096; q = 50; pairs = 12
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
def alloc(n):
    p = C.c_void_p(); ctx.check(L.tic_dev_alloc(ctx.handle, n, C.byref(p))); return p.value
sep_i = [alloc(img.size) for _ in range(pairs)]; sep_o = [alloc(img.size * 2) for _ in range(pairs)]
M2 = 2 << 20
stride_i, stride_o = (img.size + M2 - 1) // M2 * M2, (img.size * 2 + M2 - 1) // M2 * M2
base = alloc(pairs * (stride_i + stride_o) + M2)
base_al = (base + M2 - 1) // M2 * M2
pool_i = [base_al + k * stride_i for k in range(pairs)]; pool_o = [base_al + pairs * stride_i + k * stride_o for k in range(pairs)]
for p in sep_i + pool_i: ctx.check(L.tic_memcpy_h2d(ctx.handle, C.c_void_p(p), img.ctypes.data, img.size))
ms = C.c_float()
def cold(ii, oo, iters):
    pi = (C.c_void_p * pairs)(*ii); po = (C.c_void_p * pairs)(*oo)
    ctx.check(L.tic_dctq_dev_timed_rotating(ctx.handle, pi, po, pairs, h, w, w, q, 2, iters, C.byref(ms)))
    return ms.value * 1e3 / iters
def warm(i, o, iters):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, C.c_void_p(i), h, w, w, q, C.c_void_p(o), 2, iters, C.byref(ms)))
    return ms.value * 1e3 / iters
warm(sep_i[0], sep_o[0], 6000)
res = {"separate cold": [], "pooled cold": [], "separate warm": [], "pooled warm": []}
for r in range(7):
    res["separate cold"].append(cold(sep_i, sep_o, 1200)); res["pooled cold"].append(cold(pool_i, pool_o, 1200))
    res["separate warm"].append(warm(sep_i[0], sep_o[0], 3000)); res["pooled warm"].append(warm(pool_i[0], pool_o[0], 3000))
for k, v in res.items():
    print("%-14s median %.3f us  min %.3f  max %.3f" % (k, statistics.median(v), min(v), max(v)))
