"""Which walk of the strip kernel suits which frame geometry: widths / pitches that the default (chunked) schedule handles badly, under the
launcher's knobs (test hooks: TIC_TUNE re-reads them at every launch).  One process, settings interleaved.
Usage: python tools/geometry_probe.py [pixels_M=134]"""
import ctypes as C, os, statistics, sys
os.environ["TIC_TEST_HOOKS"] = "1"
os.environ["TIC_TUNE"] = "1"
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
other = None
if "--lib" in sys.argv:  # a second build of the library, timed on the same buffers (same process, same box)
    k = sys.argv.index("--lib"); path = sys.argv[k + 1]; del sys.argv[k:k + 2]
    Lo = C.CDLL(path)
    for name in ("tic_create", "tic_dctq_dev_timed"):
        res, a = N.SIGNATURES[name]; fn = getattr(Lo, name); fn.restype = res; fn.argtypes = a
    other = (Lo, Lo.tic_create(0))
ITERS = 30
if "--iters" in sys.argv:
    k = sys.argv.index("--iters"); ITERS = int(sys.argv[k + 1]); del sys.argv[k:k + 2]
if "--shapes" in sys.argv:
    k = sys.argv.index("--shapes"); SH = [tuple(int(v) for v in a.split("x")) for a in sys.argv[k + 1].split(",")]; del sys.argv[k:k + 2]
else:
    SH = None
px = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 134_000_000
shapes = [(1024, 1024), (1088, 1088), (1152, 1152), (1280, 1280), (1920, 1920), (1920, 2048), (2048, 2048), (4096, 4096), (1920, 4096), (3840, 4096)]
settings = [{}, {"TIC_CHUNK": "4"}, {"TIC_CHUNK": "16"}, {"TIC_CHUNK": "2"}, {"TIC_SCHED": "0"}, {"TIC_SCHED": "2"}, {"TIC_SCHED": "2", "TIC_CHUNK": "4"}]
if SH: shapes = SH
if len(sys.argv) > 2:
    settings = [dict(kv.split("=") for kv in a.split(";") if kv) for a in sys.argv[2:]]
buf = np.random.default_rng(1).integers(0, 256, px + (1 << 24), dtype=np.uint8)
d_in, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, buf.size, C.byref(d_in)))
ctx.check(L.tic_dev_alloc(ctx.handle, 2 * buf.size, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_in, buf.ctypes.data, buf.size))
ms = C.c_float()
KNOBS = ("TIC_SPLIT", "TIC_SCHED", "TIC_CHUNK", "TIC_MAX_WGS", "TIC_ORDER")
def run(h, w, p, it, st, lib=None):
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(st)
    if lib is None:
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_in, h, w, p, 50, d_out, 2, it, C.byref(ms)))
    else:
        assert lib[0].tic_dctq_dev_timed(lib[1], d_in, h, w, p, 50, d_out, 2, it, C.byref(ms)) == 0
    return ms.value * 1e3 / it
print("pixels %.0f M; columns: %s" % (px / 1e6, " | ".join(",".join("%s=%s" % kv for kv in s.items()) or "default" for s in settings)))
for w, p in shapes:
    h = px // p // 8 * 8
    for s in settings: run(h, w, p, 10, s)
    res = [[] for _ in settings]
    for r in range(5 if ITERS > 100 else 3):
        for k, s in enumerate(settings): res[k].append(run(h, w, p, ITERS, s))
    print("w %5d pitch %5d h %7d: " % (w, p, h) + "  ".join("%7.2f us %.3f" % (statistics.median(v), 3.0 * h * w / (statistics.median(v) * 1e-6) / 8e12) for v in res), flush=True)
    if other:
        for s in settings: run(h, w, p, 10, s, other)
        res = [[] for _ in settings]
        for r in range(3):
            for k, s in enumerate(settings): res[k].append(run(h, w, p, ITERS, s, other))
        print("   (other library)            : " + "  ".join("%7.2f us %.3f" % (statistics.median(v), 3.0 * h * w / (statistics.median(v) * 1e-6) / 8e12) for v in res), flush=True)
