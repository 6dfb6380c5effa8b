"""Drives tic_decompress on a long stream a few times: run under rocprofv3 --kernel-trace --stats to see the per-kernel split of the
device Huffman decoder + inverse stage.  Usage: python tools/prof_decompress.py [dim] [reps] [quality]"""
import ctypes as C, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
if os.environ.get('TIC_LIB'):  # another build of the library (tools/Makefile bin/libvar_%.so)
    N.LIB_PATH = N.HOOKS_LIB_PATH = os.environ['TIC_LIB']
L = N.load(); ctx = T.Context(0)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
q = int(sys.argv[3]) if len(sys.argv) > 3 else 50
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
if os.environ.get('TIC_CONTENT') == 'lenna':
    img = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (dim // 512, dim // 512)))
s = np.frombuffer(T.compress(img, q, ctx=ctx), dtype=np.uint8)
out = np.zeros((dim, dim), np.uint8)
for k in range(3):
    ctx.check(L.tic_decompress(ctx.handle, s.ctypes.data, s.size, out.ctypes.data, out.size))
t = time.perf_counter()
for k in range(reps):
    ctx.check(L.tic_decompress(ctx.handle, s.ctypes.data, s.size, out.ctypes.data, out.size))
dt = (time.perf_counter() - t) / reps
print("tic_decompress %dx%d q=%d: %.2f ms per frame (%.1f Gpix/s), stream %d bytes, decoder path %d (giveup %d)" % (dim, dim, q, dt * 1e3, dim * dim / dt / 1e9, s.size, L.tic_last_decode_path(ctx.handle), L.tic_last_decode_giveup(ctx.handle)))

# the same with stream and pixels resident in HBM (tic_decompress_dev): what the decoder costs without PCIe
d_s, d_p = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, s.size + 64, C.byref(d_s)))
ctx.check(L.tic_dev_alloc(ctx.handle, dim * dim, C.byref(d_p)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_s, s.ctypes.data, s.size))
for k in range(3):
    ctx.check(L.tic_decompress_dev(ctx.handle, d_s, s.size, d_p, dim, dim * dim, None, None))
t = time.perf_counter()
for k in range(reps):
    ctx.check(L.tic_decompress_dev(ctx.handle, d_s, s.size, d_p, dim, dim * dim, None, None))
dt = (time.perf_counter() - t) / reps
back = np.empty((dim, dim), np.uint8)
ctx.check(L.tic_memcpy_d2h(ctx.handle, back.ctypes.data, d_p, back.size))
print("tic_decompress_dev %dx%d q=%d: %.2f ms per frame (%.1f Gpix/s), same pixels: %s, decoder path %d" % (dim, dim, q, dt * 1e3, dim * dim / dt / 1e9, bool(np.array_equal(back, out)), L.tic_last_decode_path(ctx.handle)))
