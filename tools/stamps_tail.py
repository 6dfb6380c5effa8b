"""Who ends a launch of the strip kernel: in-kernel s_memtime stamps (diagnostic build, variant 52), analysed for the tail.
Per wave: 0 entry, 1 set-up done, 2 first pixels arrived, 3 loop left, 4 batch pass done, 6 = strips | irrational entries << 32 |
batch entries << 48.  Prints, per clock domain (a CU's workgroups share one counter), when the last waves leave the loop and the
kernel, and the batch pass's duration by number and kind of entries."""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import _ablate  # noqa: F401
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h = w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(sys.argv[2]) if len(sys.argv) > 2 else 50
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
n = 2048 * 4 * 8
buf = (C.c_ulonglong * n)()
ms = C.c_float()
for rep in range(3):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, q, d_out, 50, 300, C.byref(ms)))
    ctx.check(L.tic_debug_stamps(ctx.handle, d_img, h, w, w, q, d_out, buf, n, 52))
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.int64)
s = s[s[:, 0, 0] > 0]
starts = s[:, 0, 0]
order = np.argsort(starts)
groups, cur = [], [order[0]]
for k in order[1:]:
    if starts[k] - starts[cur[-1]] > 100000:
        groups.append(cur); cur = [k]
    else:
        cur.append(k)
groups.append(cur)
nE = (s[:, :, 6] >> 48) & 0xff
nI = (s[:, :, 6] >> 32) & 0xffff
dur = s[:, :, 4] - s[:, :, 3]
print("%dx%d q=%d: %d workgroups in %d clock domains; batch entries per wave mean %.2f (irrational %.2f); waves with a batch %.0f %%" % (h, w, q, len(s), len(groups), nE.mean(), nI.mean(), 100 * (nE > 0).mean()))
print("batch pass duration (cycles, loop left -> stores issued) by entries / irrational entries:")
for e in range(0, 9):
    for i in range(0, 4):
        m = (nE == e) & (nI == i)
        if m.sum() >= 5:
            print("   entries %d irrational %d: waves %5d  p50 %5.0f  p95 %5.0f  max %5.0f" % (e, i, m.sum(), np.median(dur[m]), np.percentile(dur[m], 95), dur[m].max()))
rows = []
for g in groups:
    sx = s[g]
    b0 = sx[:, :, 0].min()
    le, we = (sx[:, :, 3] - b0).ravel(), (sx[:, :, 4] - b0).ravel()
    ee, ii = ((sx[:, :, 6] >> 48) & 0xff).ravel(), ((sx[:, :, 6] >> 32) & 0xffff).ravel()
    k = np.argmax(we)
    rows.append((np.median(le), le.max(), np.median(we), we.max(), ee[k], ii[k], we.max() - le[k], np.sort(we)[-2] if len(we) > 1 else we.max()))
a = np.array(rows, dtype=float)
print("per domain (median over %d domains): loop end p50 %.0f, last %.0f | wave end p50 %.0f, last %.0f | the last wave has %.1f entries (%.2f irrational), its batch pass took %.0f; second-to-last wave ends %.0f before it" % (
    len(a), np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), a[:, 4].mean(), a[:, 5].mean(), np.median(a[:, 6]), np.median(a[:, 3] - a[:, 7])))
print("slowest domain: last wave ends at %.0f (loop end of that domain: p50 %.0f last %.0f)" % (a[:, 3].max(), a[np.argmax(a[:, 3]), 0], a[np.argmax(a[:, 3]), 1]))
