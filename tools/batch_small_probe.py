"""Phases of tic_compress_batch / tic_decompress_batch on the reference's benchmark set (49 frames of 512 x 512): host clock per call and the pipeline threads' own phase times."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
px = np.load('tests/golden/benchmark_set.npz')['pixels']
n, h, w = px.shape
block_in = np.ascontiguousarray(px)
frames = [block_in[i] for i in range(n)]
cap = L.tic_compress_bound(h, w)
pool = np.empty((n, cap), dtype=np.uint8)
inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames]); outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
tr = (C.c_double * 8)()
for q in (90, 50, 5):
    for rep in range(6):
        t0 = time.perf_counter()
        ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        dt = time.perf_counter() - t0
        ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
        zc = C.c_int(); ctx.check(L.tic_last_batch_zero_copy(ctx.handle, C.byref(zc)))
        if rep >= 3:
            print("q=%d compress_batch %.3f ms: stage/register %.3f enqueue %.3f chunk_wait %.3f read_back %.3f hand_out %.3f slot_wait %.3f join+sync %.3f unregister %.3f zero-copy streams %d" % (q, dt * 1e3, tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6], tr[7], zc.value), flush=True)
    streams = [pool[i, : lens[i]].copy() for i in range(n)]
    block = np.empty((n, h, w), np.uint8)
    sp = (C.c_void_p * n)(*[s.ctypes.data for s in streams]); sl = (C.c_size_t * n)(*[s.size for s in streams])
    pp = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)]); pc = (C.c_size_t * n)(*([h * w] * n))
    for rep in range(6):
        t0 = time.perf_counter()
        ctx.check(L.tic_decompress_batch(ctx.handle, sp, sl, n, pp, pc, None, None))
        dt = time.perf_counter() - t0
        ctx.check(L.tic_last_batch_phases(ctx.handle, tr))
        if rep >= 3: print("q=%d decompress_batch %.3f ms: pack %.3f enqueue %.3f wait+download %.3f hand_out %.3f" % (q, dt * 1e3, tr[0], tr[1], tr[2], tr[4]), flush=True)
