"""Host-level round-trip times of the codec on random frames (Python compress / decompress, C-ABI tic_idctq and
tic_compress with preallocated buffers)."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, tinyimgcodec_amd as T
ctx = T.Context(0)
for dim in (1024, 4096):
    img = np.random.default_rng(1).integers(0,256,(dim,dim),dtype=np.uint8)
    bs = T.compress(img, 50, ctx=ctx)
    T.decompress(bs, ctx=ctx)
    t=time.perf_counter(); out = T.decompress(bs, ctx=ctx); dt=time.perf_counter()-t
    t=time.perf_counter(); bs2 = T.compress(img, 50, ctx=ctx); dc=time.perf_counter()-t
    print("%d^2: compress %.1f ms (%.0f Mpix/s), decompress %.1f ms (%.0f Mpix/s), stream %d B" % (dim, dc*1e3, dim*dim/dc/1e6, dt*1e3, dim*dim/dt/1e6, len(bs)))
# split of decompress: GPU part alone (host coefficients -> pixels through tic_idctq)
import ctypes as C
from tinyimgcodec_amd import _native as N
L = N.load()
dim = 4096
img = np.random.default_rng(1).integers(0, 256, (dim, dim), dtype=np.uint8)
d = T.encode(img, 50, ctx=ctx)
nblk = L.tic_num_blocks(dim, dim)
zz = np.zeros((nblk, 64), np.int16)
zz[:, 0] = np.cumsum(d["dc"]).astype(np.int16)
zz[:, 1:] = d["ac"].astype(np.int16)
out = np.zeros((dim, dim), np.uint8)
for k in range(3):
    t = time.perf_counter()
    ctx.check(L.tic_idctq(ctx.handle, zz.ctypes.data, dim, dim, 50, out.ctypes.data, out.size))
    dt = time.perf_counter() - t
print("tic_idctq 4096^2 host->host: %.1f ms" % (dt * 1e3), "| decoded == decompress(compress(img)):", bool(np.array_equal(out, T.decompress(T.compress(img, 50, ctx=ctx), ctx=ctx))))
# C-ABI tic_compress with a preallocated, pre-touched output buffer (no Python object handling)
cap = L.tic_compress_bound(dim, dim)
outb = np.zeros(cap, np.uint8)
nn = C.c_size_t()
for k in range(4):
    t = time.perf_counter()
    ctx.check(L.tic_compress(ctx.handle, img.ctypes.data, dim, dim, dim, 50, outb.ctypes.data, cap, C.byref(nn)))
    dt = time.perf_counter() - t
print("tic_compress (C-ABI) 4096^2 host->host: %.2f ms, %d bytes" % (dt * 1e3, nn.value))
