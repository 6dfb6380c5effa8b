"""A/B of the batch pipeline's thread binding (tic_set_numa_binding) and of registered input, host -> host, 256 x 1080p through
tic_compress_batch, variants interleaved round-robin in one process.  Prints the device's NUMA node and where the frames live."""
import ctypes as C, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
h, w, n, q = 1080, 1920, 256, 50
frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
block = np.stack(frames)
cap = L.tic_compress_bound(h, w)
pool = np.empty((n, cap), dtype=np.uint8)
outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
inp_p = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
inp_r = (C.c_void_p * n)(*[block[i].ctypes.data for i in range(n)])
node, ncpus = C.c_int(), C.c_int()
ctx.check(L.tic_numa_info(ctx.handle, C.byref(node), C.byref(ncpus)))
print("device NUMA node %d, %d CPUs of it in this process" % (node.value, ncpus.value))
ctx.check(L.tic_host_register(ctx.handle, block.ctypes.data, block.nbytes))
variants = [("pageable, threads bound", inp_p, 1), ("pageable, threads unbound", inp_p, 0), ("registered, bound", inp_r, 1), ("registered, unbound", inp_r, 0)]
res = {v[0]: [] for v in variants}
for rnd in range(7):
    for name, inp, bind in variants:
        ctx.check(L.tic_set_numa_binding(ctx.handle, bind))
        t = time.perf_counter()
        ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        dt = time.perf_counter() - t
        if rnd: res[name].append(dt * 1e3)
for name, _, _ in variants:
    r = res[name]
    print("%-28s median %6.2f ms  min %6.2f  max %6.2f   %8.0f frames/s" % (name, statistics.median(r), min(r), max(r), n / statistics.median(r) * 1e3))
ctx.check(L.tic_host_unregister(ctx.handle, block.ctypes.data))
