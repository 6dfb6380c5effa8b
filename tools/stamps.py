"""In-kernel s_memtime stamps of the hybrid kernel's phases (diagnostic build; shares, not absolute time)."""
import ctypes as C, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import _ablate  # noqa: F401  (experiment build of the library)
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
    ctx.check(L.tic_debug_stamps(ctx.handle, d_img, h, w, w, 50, d_out, buf, n, 17))
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.int64)
wg_ids = np.nonzero(s[:, 0, 0] > 0)[0]
s = s[s[:, 0, 0] > 0]                     # workgroups that ran
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
shown = 0
for g in groups:
    if len(g) <= 10 and shown < 4:
        shown += 1
        b0 = s[g][:, :, 0].min()
        print("  one domain: " + "  ".join("wg %4d: start %5d end %5d last %5d" % (wg_ids[k], s[k, :, 0].min() - b0, s[k, :, 1].max() - b0, s[k, :, :7].max() - b0) for k in sorted(g, key=lambda k: wg_ids[k])))
for g in groups:
    sx = s[g]
    b0 = sx[:, :, 0].min()
    ls, le = sx[:, :, 0] - b0, sx[:, :, 1] - b0
    last = sx[:, :, :7].max() - b0
    print("clock domain with %4d workgroups: loop start p5 %6d p50 %6d p95 %6d max %6d | loop end p5 %6d p50 %6d p95 %6d max %6d | last stamp %6d | loop len p50 %6d" % (
        len(g), np.percentile(ls, 5), np.median(ls), np.percentile(ls, 95), ls.max(), np.percentile(le, 5), np.median(le), np.percentile(le, 95), le.max(), last, np.median(le - ls)))
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

