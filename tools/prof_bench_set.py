"""The reference's benchmark loop on ONE 512 x 512 image at one quality, many times: run under rocprofv3 --kernel-trace --memory-copy-trace --stats to see
where a 60 us compress() and a 95 us decompress() spend their time.  Usage: python tools/prof_bench_set.py [quality=50] [reps=200]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
q = int(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
img = np.load('tests/golden/benchmark_set.npz')['pixels'][0]
ctx = T.Context(0)
for _ in range(10):
    s = T.compress(img, q, ctx=ctx); out = T.decompress(s, ctx=ctx)
t0 = time.perf_counter()
for _ in range(reps): s = T.compress(img, q, ctx=ctx)
t1 = time.perf_counter()
for _ in range(reps): out = T.decompress(s, ctx=ctx)
t2 = time.perf_counter()
print("compress %.1f us, decompress %.1f us per call (%d bytes)" % ((t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6, len(s)))
