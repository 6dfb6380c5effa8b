import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h, w, n, q = 1080, 1920, 256, 50
frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
block = np.stack(frames)
cap = L.tic_compress_bound(h, w)
pool = np.zeros((n, cap), dtype=np.uint8)
outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
inp_r = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)])
ctx.check(L.tic_host_register(ctx.handle, block.ctypes.data, block.nbytes))
tr = (C.c_double * 8)()
for r in range(5):
    t = time.perf_counter()
    ctx.check(L.tic_compress_batch(ctx.handle, inp_r, n, h, w, w, q, outp, caps, lens, 0))
    dt = time.perf_counter() - t
    ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
    print("registered: %.2f ms | submit: register %.2f enqueue %.2f slot wait %.2f | reader: chunk wait %.2f read-back %.2f | hand-out %.2f" % (dt * 1e3, tr[0], tr[1], tr[5], tr[2], tr[3], tr[4]))
# kernels only for a chunk: time the device work per chunk without copies?  (transform + entropy of 16 frames resident)
