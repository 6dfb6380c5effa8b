"""In-kernel s_memtime stamps of the strip kernel with wave-local rare paths (diagnostic build, variant 52).
Per wave: 0 entry, 1 set-up done (first pixel load about to issue), 2 first strip's pixels arrived, 3 loop left, 4 batch pass
done (its stores issued), 5 = batch entries, 6 = strips | second-level blocks << 32, 7 = exact-redo mask."""
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
    e, p, f, l, z = sx[:, :, 0] - b0, sx[:, :, 1] - b0, sx[:, :, 2] - b0, sx[:, :, 3] - b0, sx[:, :, 4] - b0
    withb = ((sx[:, :, 6] >> 48) & 0xff) > 0
    rows.append((len(g), np.median(e), e.max(), np.median(p - e), np.median(f - p), np.median(l - f), np.median(l), l.max(),
                 np.median((z - l)[withb]) if withb.any() else 0, np.median(z), z.max()))
    if len(g) <= 10 and len(rows) <= 6:
        print("  one domain: " + "  ".join("wg %4d: entry %5d first-data %5d loop-end %5d end %5d" % (wg_ids[k], s[k, :, 0].min() - b0, s[k, :, 2].min() - b0, s[k, :, 3].max() - b0, s[k, :, 4].max() - b0) for k in sorted(g, key=lambda k: wg_ids[k])))
small = [r for r in rows if r[0] <= 10]
a = np.array(small)
print("per-CU domains (%d): entry p50 %.0f max %.0f | set-up p50 %.0f | first data after set-up p50 %.0f | loop p50 %.0f | loop end p50 %.0f, last wave p50 %.0f | batch pass p50 %.0f | wave end p50 %.0f, last wave p50 %.0f max %.0f cycles" % (
    len(small), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), np.median(a[:, 4]), np.median(a[:, 5]), np.median(a[:, 6]), np.median(a[:, 7]), np.median(a[:, 8]), np.median(a[:, 9]), np.median(a[:, 10]), a[:, 10].max()))
nE = (s[:, :, 6] >> 48) & 0xff
print("batch entries per wave: mean %.2f max %d, waves with a batch: %.0f %%" % (nE.mean(), nE.max(), 100.0 * (nE > 0).mean()))
rnd = wg_ids // 256
# per-round end times relative to the start of the workgroup's clock domain (CU)
dom0 = np.zeros(len(s), np.int64)
for g in groups:
    dom0[g] = s[g][:, :, 0].min()
for r in range(int(rnd.max()) + 1):
    x = s[rnd == r]
    rel = x[:, :, 4] - dom0[rnd == r][:, None]
    relL = x[:, :, 3] - dom0[rnd == r][:, None]
    ok = rel < 200000
    print("round %d: loop end p50 %6.0f p95 %6.0f | wave end p50 %6.0f p95 %6.0f max %6.0f (since the CU's first entry)" % (
        r, np.median(relL[ok]), np.percentile(relL[ok], 95), np.median(rel[ok]), np.percentile(rel[ok], 95), rel[ok].max()))
for r in range(int(rnd.max()) + 1):
    x = s[rnd == r]
    print("round %d: kernel args arrived after p50 %5.0f | descriptor after p50 %5.0f | set-up done p50 %5.0f | first data p50 %5.0f (all since the wave's own entry); loop p50 %5.0f cycles, strips %.1f" % (
        r, np.median(x[:, :, 5] - x[:, :, 0]), np.median(x[:, :, 7] - x[:, :, 0]), np.median(x[:, :, 1] - x[:, :, 0]), np.median(x[:, :, 2] - x[:, :, 0]),
        np.median(x[:, :, 3] - x[:, :, 2]), (x[:, :, 6] & 0xffff).mean()))
strips = s[:, :, 6] & 0xffff
sec = (s[:, :, 6] >> 32) & 0xffff
print("strips per wave: mean %.2f max %d | second-level blocks per wave mean %.3f | waves with an exact-redo mask: %d" % (strips.mean(), strips.max(), sec.mean(), ((s[:, :, 6] >> 63) != 0).sum()))
# per-wave table of a few CUs: round, wave, first data, loop end, wave end (since the CU's first entry), strips, batch entries
shown = 0
for g in groups:
    if len(g) != 5 or shown >= 3:
        continue
    shown += 1
    b0 = s[g][:, :, 0].min()
    print("CU with workgroups", sorted(int(wg_ids[k]) for k in g))
    for k in sorted(g, key=lambda k: wg_ids[k]):
        for wv in range(4):
            x = s[k, wv]
            print("   round %d wave %d: entry %5d first data %5d loop end %6d end %6d | loop %5d cycles for %2d strips = %5d per strip | batch %d entries, %4d cycles" % (
                wg_ids[k] // 256, wv, x[0] - b0, x[2] - b0, x[3] - b0, x[4] - b0, x[3] - x[2], x[6] & 0xffff, (x[3] - x[2]) // max(1, x[6] & 0xffff), (x[6] >> 48) & 0xff, x[4] - x[3]))
