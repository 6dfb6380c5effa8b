import sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
ctx = T.Context(0)
frames = [np.random.default_rng(1234 + i).integers(0, 256, (1080, 1920), dtype=np.uint8) for i in range(64)]
for rep in range(3):
    t0 = time.perf_counter(); out = T.compress_batch(frames, 50, threads=0, ctx=ctx); dt = time.perf_counter() - t0
    print("gpu entropy batch of 64: %.1f ms" % (dt * 1e3))
img = np.random.default_rng(1234).integers(0, 256, (4096, 4096), dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); bs = T.compress(img, 50, ctx=ctx); dt = time.perf_counter() - t0
    print("compress 4096^2: %.2f ms, %d bytes" % (dt * 1e3, len(bs)))
