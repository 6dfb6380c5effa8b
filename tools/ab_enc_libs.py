"""A/B of encoder builds in one process: tic_compress_dev (synchronous) and tic_compress_dev_async in bursts of 64 (image and stream resident in HBM) of a 4096^2
frame through the product library and other builds of it (tools/Makefile bin/libvar_%.so VARSRC=tic_entropy_gpu.hip), rounds interleaved.
Usage: python tools/ab_enc_libs.py tools/bin/libvar_86.so [...] [--quality 50]"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("other", nargs="*")
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--quality", type=int, default=50)
args = ap.parse_args()
dim, q = args.dim, args.quality
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
names = ("tic_create", "tic_dev_alloc", "tic_memcpy_h2d", "tic_memcpy_d2h", "tic_compress_dev", "tic_compress_dev_async", "tic_async_result", "tic_compress_bound", "tic_last_error")
def bind(path):
    L = C.CDLL(path)
    for name in names:
        res, a = N.SIGNATURES[name]
        fn = getattr(L, name); fn.restype = res; fn.argtypes = a
    return L
libs = {"product": bind(N.LIB_PATH)}
for pth in args.other:
    libs[os.path.basename(pth).replace("lib", "").replace(".so", "")] = bind(pth)
state, ref = {}, None
for name, L in libs.items():
    ctx = L.tic_create(0); assert ctx
    cap = L.tic_compress_bound(dim, dim)
    d_i, d_o = C.c_void_p(), C.c_void_p()
    assert L.tic_dev_alloc(ctx, img.size, C.byref(d_i)) == 0 and L.tic_dev_alloc(ctx, cap + 64, C.byref(d_o)) == 0
    assert L.tic_memcpy_h2d(ctx, d_i, img.ctypes.data, img.size) == 0
    state[name] = (L, ctx, d_i, d_o, cap)
n = C.c_size_t()
def sync(name, reps=30):
    L, ctx, d_i, d_o, cap = state[name]
    t = time.perf_counter()
    for _ in range(reps):
        assert L.tic_compress_dev(ctx, d_i, dim, dim, dim, q, d_o, cap, C.byref(n)) == 0
    return (time.perf_counter() - t) / reps * 1e6
def burst(name, reps=3):
    L, ctx, d_i, d_o, cap = state[name]
    tk = [C.c_longlong() for _ in range(64)]
    t = time.perf_counter()
    for _ in range(reps):
        for k in range(64):
            assert L.tic_compress_dev_async(ctx, d_i, dim, dim, dim, q, d_o, cap, C.byref(tk[k])) == 0
        for k in range(64):
            assert L.tic_async_result(ctx, tk[k].value, 1, C.byref(n)) == 0
    return (time.perf_counter() - t) / (64 * reps) * 1e6
for name in libs:
    sync(name, 3); burst(name, 1)
    L, ctx, d_i, d_o, cap = state[name]
    back = np.empty(n.value, np.uint8)
    assert L.tic_memcpy_d2h(ctx, back.ctypes.data, d_o, n.value) == 0
    if ref is None: ref = back
    print("%-10s stream equal to the product's: %s (%d bytes)" % (name, bool(np.array_equal(back, ref)), n.value))
rs, rb = {k: [] for k in libs}, {k: [] for k in libs}
for r in range(args.rounds):
    for name in libs:
        rs[name].append(sync(name)); rb[name].append(burst(name))
for name in libs:
    print("%-10s tic_compress_dev median %.1f us (min %.1f max %.1f)   async bursts of 64: median %.1f us per frame (min %.1f max %.1f)"
          % (name, statistics.median(rs[name]), min(rs[name]), max(rs[name]), statistics.median(rb[name]), min(rb[name]), max(rb[name])))
