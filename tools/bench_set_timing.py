"""The reference's own benchmark loop (tests/benchmark.py:12-23: 49 images of 512 x 512, six qualities; compress() then decompress() per image) through
the Python mirror, per call, host clock: what a user of the reference's API sees.  Usage: python tools/bench_set_timing.py"""
import ctypes as C, hashlib, json, os, sys, time, statistics
if len(sys.argv) > 1:  # python tools/bench_set_timing.py <min blocks> <min bits>: the device decoder's thresholds (hooks build)
    os.environ["TIC_TEST_HOOKS"] = "1"; os.environ["TIC_DECODE_MIN_BLOCKS"] = sys.argv[1]; os.environ["TIC_DECODE_MIN_BITS"] = sys.argv[2]
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load()
px = np.load('tests/golden/benchmark_set.npz')['pixels']
ctx = T.Context(0)
hooks = len(sys.argv) > 1
by = {(e['image'], e['quality']): e for e in json.load(open('tests/golden/benchmark_set.json'))['entries']}
rb, tr = C.c_int(), C.c_int()
for q in (90, 80, 50, 20, 10, 5):
    tc, td, on_dev, second = [], [], 0, 0
    for rep in range(3):
        for i in range(len(px)):
            img = px[i]
            t0 = time.perf_counter(); s = T.compress(img, q, ctx=ctx); t1 = time.perf_counter(); out = T.decompress(s, ctx=ctx); t2 = time.perf_counter()
            if rep:
                tc.append(t1 - t0); td.append(t2 - t1); on_dev += L.tic_last_decode_path(ctx.handle) == 1
                L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr)); second += tr.value > 1
            else:  # the reference's own stream and pixels (tests/golden/benchmark_set.json)
                e = by[(i + 1, q)]
                assert hashlib.sha256(s).hexdigest() == e['sha256'] and hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == e['decoded_sha256'], (q, i)
            if not rep and hooks:  # the same stream through the host decoder: same pixels
                os.environ["TIC_DECODE_HOST"] = "1"; ref = T.decompress(s, ctx=ctx); del os.environ["TIC_DECODE_HOST"]
                assert np.array_equal(out, ref), (q, i)
    print("q=%2d: compress median %6.1f us  decompress median %6.1f us  max %6.1f  (stream %5.1f KB, %d images x 2 passes, %d of %d on the device decoder, %d second runs)"
          % (q, statistics.median(tc) * 1e6, statistics.median(td) * 1e6, max(td) * 1e6, len(s) / 1024, len(px), on_dev, len(td), second))
