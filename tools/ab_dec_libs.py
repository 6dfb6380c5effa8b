"""A/B of decoder builds in one process: tic_decompress_dev (stream and pixels resident in HBM) of a 4096^2 stream through the product
library and other builds of it (tools/Makefile bin/libvar_%.so VARSRC=tic_entropy_dec_gpu.hip), rounds interleaved.
Usage: python tools/ab_dec_libs.py tools/bin/libvar_1.so [...] [--quality 50] [--content noise|lenna]"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("other", nargs="*")
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--quality", type=int, default=50)
ap.add_argument("--content", default="noise")
args = ap.parse_args()
dim, q = args.dim, args.quality
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
if args.content == "lenna":
    img = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (dim // 512, dim // 512)))
s = np.frombuffer(T.compress(img, q), dtype=np.uint8)
want = T.decompress(s.tobytes())
names = ("tic_create", "tic_dev_alloc", "tic_memcpy_h2d", "tic_memcpy_d2h", "tic_decompress_dev", "tic_last_decode_path", "tic_last_error")
def bind(path):
    L = C.CDLL(path)
    for name in names:
        res, a = N.SIGNATURES[name]
        fn = getattr(L, name); fn.restype = res; fn.argtypes = a
    return L
libs = {"product": bind(N.LIB_PATH)}
for pth in args.other:
    libs[os.path.basename(pth).replace("lib", "").replace(".so", "")] = bind(pth)
state = {}
for name, L in libs.items():
    ctx = L.tic_create(0); assert ctx
    d_s, d_p = C.c_void_p(), C.c_void_p()
    assert L.tic_dev_alloc(ctx, s.size + 64, C.byref(d_s)) == 0 and L.tic_dev_alloc(ctx, dim * dim, C.byref(d_p)) == 0
    assert L.tic_memcpy_h2d(ctx, d_s, s.ctypes.data, s.size) == 0
    state[name] = (L, ctx, d_s, d_p)
def run(name, reps):
    L, ctx, d_s, d_p = state[name]
    t = time.perf_counter()
    for _ in range(reps):
        rc = L.tic_decompress_dev(ctx, d_s, s.size, d_p, dim, dim * dim, None, None)
        assert rc == 0, L.tic_last_error(ctx)
    return (time.perf_counter() - t) / reps * 1e6
for name in libs:
    run(name, 3)
    L, ctx, d_s, d_p = state[name]
    back = np.empty((dim, dim), np.uint8)
    assert L.tic_memcpy_d2h(ctx, back.ctypes.data, d_p, back.size) == 0
    print("%-10s pixels equal the decoder's: %s (path %d)" % (name, bool(np.array_equal(back, want)), L.tic_last_decode_path(ctx)))
res = {n: [] for n in libs}
for r in range(args.rounds):
    for name in libs:
        res[name].append(run(name, args.reps))
for name, v in res.items():
    print("%-10s tic_decompress_dev median %.1f us  min %.1f  max %.1f" % (name, statistics.median(v), min(v), max(v)))
