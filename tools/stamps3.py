"""In-kernel s_memtime stamps of the queue kernel (diagnostic build, variant 71): per wave 0 entry, 1 first loads issued,
2 first strip's pixels arrived, 3 strip loop left (last run), 4 end, 6 = strips | second-level blocks << 32 | flushes << 48."""
import ctypes as C, os, sys
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
WPW = 16
n = 2048 * 4 * 8
buf = (C.c_ulonglong * n)()
ms = C.c_float()
for rep in range(3):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, 70, 300, C.byref(ms)))  # warm clocks
    ctx.check(L.tic_debug_stamps(ctx.handle, d_img, h, w, w, 50, d_out, buf, n, 71))
s = np.frombuffer(buf, dtype=np.uint64).reshape(-1, WPW, 8).astype(np.int64)
s = s[s[:, 0, 0] > 0]
print("teams:", len(s))
rows = []
for k in range(len(s)):
    x = s[k]
    x = x[x[:, 0] > 0]
    b0 = x[:, 0].min()
    rows.append((np.median(x[:, 1] - b0), np.median(x[:, 2] - b0), np.median(x[:, 3] - b0), (x[:, 3] - b0).max(), np.median(x[:, 4] - b0), (x[:, 4] - b0).max(),
                 (x[:, 6] & 0xffff).mean(), (x[:, 6] & 0xffff).min(), (x[:, 6] & 0xffff).max()))
a = np.array(rows)
print("per team (median over teams): first loads issued %.0f | first data %.0f | loop end p50 %.0f last %.0f | wave end p50 %.0f last %.0f cycles | strips per wave mean %.1f min %.0f max %.0f" % tuple(np.median(a, axis=0)))
print("last wave end: p50 %.0f p95 %.0f max %.0f" % (np.median(a[:, 5]), np.percentile(a[:, 5], 95), a[:, 5].max()))
for k in range(2):
    x = s[k]
    b0 = x[:, 0].min()
    print("team", k)
    for wv in range(WPW):
        y = x[wv]
        print("   wave %2d: entry %5d loads issued %5d first data %5d loop end %6d end %6d | %2d strips, %5d cycles per strip | flushes %d" % (
            wv, y[0] - b0, y[1] - b0, y[2] - b0, y[3] - b0, y[4] - b0, y[6] & 0xffff, (y[3] - y[2]) // max(1, y[6] & 0xffff), (y[6] >> 48) & 0xff))
