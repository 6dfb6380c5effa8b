"""A/B of the batch pipeline between builds of the library in one process: 256 x 1080p host -> host through tic_compress_batch, pageable
frames (pinned in place by the call) and one caller-registered block, rounds interleaved.  Usage: python tools/ab_batch_libs.py other.so [...]"""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
from tinyimgcodec_amd import _native as N
names = ("tic_create", "tic_compress_batch", "tic_compress_bound", "tic_host_register", "tic_host_unregister", "tic_last_error")
def bind(path):
    L = C.CDLL(path)
    for name in names:
        res, a = N.SIGNATURES[name]
        fn = getattr(L, name); fn.restype = res; fn.argtypes = a
    return L
libs = {"product": bind(N.LIB_PATH)}
for pth in sys.argv[1:]:
    libs[os.path.basename(pth).replace("lib", "").replace(".so", "")] = bind(pth)
h, w, n, q = 1080, 1920, 256, 50
frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
block = np.stack(frames)
L0 = libs["product"]
cap = L0.tic_compress_bound(h, w)
pool = np.zeros((n, cap), dtype=np.uint8)
outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
inp_p = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
inp_r = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)])
ctxs = {name: L.tic_create(0) for name, L in libs.items()}
def run(name, inp):
    L, ctx = libs[name], ctxs[name]
    t = time.perf_counter()
    rc = L.tic_compress_batch(ctx, inp, n, h, w, w, q, outp, caps, lens, 0)
    assert rc == 0, L.tic_last_error(ctx)
    return (time.perf_counter() - t) * 1e3
ref = None
for name in libs:
    run(name, inp_p); run(name, inp_p)
    sizes = [int(lens[i]) for i in range(n)]
    ref = ref or sizes
    assert sizes == ref
res = {(nm, k): [] for nm in libs for k in ("pageable", "registered")}
for r in range(9):
    for name in libs:
        res[(name, "pageable")].append(run(name, inp_p))
    assert L0.tic_host_register(ctxs["product"], block.ctypes.data, block.nbytes) == 0
    for name in libs:
        res[(name, "registered")].append(run(name, inp_r))
    assert L0.tic_host_unregister(ctxs["product"], block.ctypes.data) == 0
for (name, k), v in res.items():
    print("%-10s %-10s median %6.2f ms  min %6.2f  max %6.2f  (%d frames/s)" % (name, k, statistics.median(v), min(v), max(v), n / statistics.median(v) * 1e3))
