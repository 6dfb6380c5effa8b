import os, sys, time
os.environ["TIC_TEST_HOOKS"] = "1"
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
for q in (90, 10):
    img = np.random.default_rng(5).integers(0, 256, (16384, 16384), dtype=np.uint8)
    s = T.compress(img, q, ctx=ctx)
    t = time.time(); a = T.decompress(s, ctx=ctx); ta = time.time() - t
    rb, tr = C.c_int(), C.c_int(); L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr))
    path = L.tic_last_decode_path(ctx.handle)
    os.environ["TIC_DECODE_HOST"] = "1"
    t = time.time(); b = T.decompress(s, ctx=ctx); tb = time.time() - t
    os.environ.pop("TIC_DECODE_HOST")
    print("16384^2 q=%d: stream %d bytes, device decoder path %d range %d runs %d (%.3f s), host decoder %.3f s, equal: %s" % (q, len(s), path, rb.value, tr.value, ta, tb, bool(np.array_equal(a, b))), flush=True)
    del a, b, s, img
