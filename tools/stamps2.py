"""In-kernel s_memtime stamps of the strip kernel with wave-local rare paths (diagnostic build, variant 52).
Per wave: 0 entry, 1 set-up done (first pixel load about to issue), 2 first strip's pixels arrived, 3 loop left and stores
landed, 6 = strips | second-level blocks << 32, 7 = exact-redo mask."""
import ctypes as C, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _ablate  # noqa: F401  (experiment build of the library)
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h = w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
n = 2048 * 4 * 8
buf = (C.c_ulonglong * n)()
ms = C.c_float()
for rep in range(3):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, 50, 300, C.byref(ms)))  # warm clocks
    ctx.check(L.tic_debug_stamps(ctx.handle, d_img, h, w, w, 50, d_out, buf, n, 52))
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.int64)
ran = s[:, 0, 0] > 0
wg_ids = np.nonzero(ran)[0]
s = s[ran]
print("workgroups:", len(s))
# s_memtime counters are not synchronised across the chip: cluster workgroups by counter value (gaps > 100k ticks)
starts = s[:, 0, 0]
order = np.argsort(starts)
groups, cur = [], [order[0]]
for k in order[1:]:
    if starts[k] - starts[cur[-1]] > 100000:
        groups.append(cur); cur = [k]
    else:
        cur.append(k)
groups.append(cur)
rows = []
for g in groups:
    sx = s[g]
    b0 = sx[:, :, 0].min()
    e, p, f, l = sx[:, :, 0] - b0, sx[:, :, 1] - b0, sx[:, :, 2] - b0, sx[:, :, 3] - b0
    rows.append((len(g), np.median(e), e.max(), np.median(p - e), np.median(f - p), np.median(l - f), np.median(l), l.max()))
    if len(g) <= 10 and len(rows) <= 6:
        print("  one domain: " + "  ".join("wg %4d: entry %5d first-data %5d end %5d" % (wg_ids[k], s[k, :, 0].min() - b0, s[k, :, 2].min() - b0, s[k, :, 3].max() - b0) for k in sorted(g, key=lambda k: wg_ids[k])))
small = [r for r in rows if r[0] <= 10]
a = np.array(small)
print("per-CU domains (%d): entry p50 %.0f max %.0f | set-up p50 %.0f | first data after set-up p50 %.0f | loop p50 %.0f | end p50 %.0f max p50 %.0f max %.0f cycles" % (
    len(small), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), np.median(a[:, 4]), np.median(a[:, 5]), np.median(a[:, 6]), np.median(a[:, 7]), a[:, 7].max()))
strips = s[:, :, 6] & 0xffffffff
sec = s[:, :, 6] >> 32
print("strips per wave: mean %.2f max %d | second-level blocks per wave mean %.3f | waves with an exact-redo mask: %d" % (strips.mean(), strips.max(), sec.mean(), (s[:, :, 7] != 0).sum()))
