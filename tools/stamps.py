"""In-kernel s_memtime stamps of the hybrid kernel's phases (diagnostic build; shares, not absolute time)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h = w = 4096
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
n = 2048 * 4 * 8
buf = (C.c_ulonglong * n)()
for rep in range(3):
    ctx.check(L.tic_debug_stamps(ctx.handle, d_img, h, w, w, 50, d_out, buf, n))
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.int64)
t0 = s[:, :, 0].min()
names = ["loop start", "loop end", "stores drained", "barrier passed", "fetched+transposed", "pass computed", "patches landed"]
for wave, label in ((0, "tie waves (0,2)"), (1, "redo waves (1,3)")):
    sel = s[:, wave::2, :].reshape(-1, 8)
    work = sel[sel[:, 7] > 0]
    idle = sel[sel[:, 7] == 0]
    print(label, "with entries:", len(work), "without:", len(idle))
    for k, nm in enumerate(names):
        v = work[:, k] - t0
        v = v[work[:, k] > 0]
        if len(v):
            print("   %-20s median %7d  p95 %7d  max %7d cycles" % (nm, np.median(v), np.percentile(v, 95), v.max()))
    if len(work):
        d = work[work[:, 6] > 0]
        for q in (50, 95, 99, 100):
            print("   deltas p%-3d: drain %d, barrier %d, fetch %d, compute %d, land %d | loop %d | post total %d" % ((q,) + tuple(
                np.percentile(d[:, k + 1] - d[:, k], q) for k in range(1, 6)) + (np.percentile(d[:, 1] - d[:, 0], q), np.percentile(d[:, 6] - d[:, 1], q))))
        print("   entries per wave: mean %.2f max %d" % (d[:, 7].mean(), d[:, 7].max()))
print("kernel span (last stamp - first): %d cycles" % (s[:, :, :7].max() - t0))
