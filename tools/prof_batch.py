"""Host-thread phase times of tic_compress_batch (256 x 1080p, device entropy stage) through tic_last_batch_phases:
stage = pageable -> pinned copies, submit = enqueue of H2D + kernels + length copies, wait = hipEventSynchronize on the chunk,
readback = D2H of the streams into pinned memory (enqueue + wait), handout = memcpy into the caller's buffers."""
import ctypes as C, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
import os
if os.environ.get('TIC_LIB'):  # another build of the library (tools/Makefile bin/libvar_%.so)
    N.LIB_PATH = N.HOOKS_LIB_PATH = os.environ['TIC_LIB']
L = N.load(); ctx = T.Context(0)
n, h, w = 256, 1080, 1920
frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
cap = L.tic_compress_bound(h, w)
outs = [np.empty(cap, np.uint8) for _ in range(n)]
imgs = (C.c_void_p * n)(*[f.ctypes.data for f in frames]); outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
tr = (C.c_double * 8)()
for rep in range(4):
    t = time.perf_counter()
    ctx.check(L.tic_compress_batch(ctx.handle, imgs, n, h, w, w, 50, outp, caps, lens, 0))
    dt = time.perf_counter() - t
    ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
    tr = (C.c_double * 8)(*[v / 1e3 for v in tr])
    print("rep %d: %.2f ms total | submitting thread: stage %.2f enqueue %.2f slot wait %.2f | finishing thread: chunk wait %.2f readback %.2f handout %.2f ms" % (rep, dt * 1e3, tr[0] * 1e3, tr[1] * 1e3, tr[5] * 1e3, tr[2] * 1e3, tr[3] * 1e3, tr[4] * 1e3), flush=True)
