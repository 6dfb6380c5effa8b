"""Kernel efficiency against frame width at a constant pixel count (dense rows: pitch = width + pitch_extra): tic_dctq_dev_timed on
random frames.  Usage: python tools/width_sweep.py [pixels_M=134] [w_lo=1024] [w_hi=4096] [step=64] [pitch_extra=0]

Round 5: the chip's clocks are settled first (150 ms of launches) and the widths are measured in three interleaved rounds (median).
Round 4's version measured every width once, in ascending order, right after the process had started: the first widths of the list
(1024 ... 1280) were timed on a chip that was still ramping up, which read as a "dip at 1088" (0.58 of 8 TB/s against 0.75 for the
same width measured fourth in another order, same box, same minute: profiles/r05_geometry.txt)."""
import ctypes as C, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
px = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 134_000_000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
st = int(sys.argv[4]) if len(sys.argv) > 4 else 64
pitch_extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0
buf = np.random.default_rng(1).integers(0, 256, px + (1 << 22), dtype=np.uint8)
d_in, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, buf.size + (1 << 26), C.byref(d_in)))
ctx.check(L.tic_dev_alloc(ctx.handle, 2 * buf.size, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_in, buf.ctypes.data, buf.size))
ms = C.c_float()
widths = list(range(lo, hi + 1, st))
def run(w, it):
    pitch = w + pitch_extra
    h = px // pitch // 8 * 8
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_in, h, w, pitch, 50, d_out, 2, it, C.byref(ms)))
    return ms.value * 1e3 / it, h, pitch
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:  # clock settling
    run(widths[len(widths) // 2], 20)
res = {w: [] for w in widths}
for r in range(3):
    for w in (widths if r % 2 == 0 else widths[::-1]):
        res[w].append(run(w, 20)[0])
for w in widths:
    us = statistics.median(res[w])
    _, h, pitch = run(w, 1)
    print("w %5d (%3d strips/row, pitch %5d) h %7d: %8.2f us  %.3f of 8 TB/s   (rounds: %s)" % (w, w // 64, pitch, h, us, 3.0 * h * w / (us * 1e-6) / 8e12, " ".join("%.1f" % v for v in res[w])), flush=True)
