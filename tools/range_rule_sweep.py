"""The device decoder's range rule (TIC_DECODE_RULE = "<average blocks per range>,<least 32-bit words per range>", a test hook) against what it
costs and what it risks: the reference's benchmark loop (decompress() of the 49 x 6 streams, host clock, second runs), device-resident decodes of
4096^2 frames, and tools/stress_decoder.py's second runs.  Usage: python tools/range_rule_sweep.py "3,17" "3,9" ...  [--stress 300]"""
import ctypes as C, os, re, statistics, subprocess, sys, time
os.environ["TIC_TEST_HOOKS"] = "1"
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N

rules = [a for a in sys.argv[1:] if "," in a]
stress = int(sys.argv[sys.argv.index("--stress") + 1]) if "--stress" in sys.argv else 0
L = N.load(); ctx = T.Context(0)
px = np.load('tests/golden/benchmark_set.npz')['pixels']
streams = {q: [T.compress(px[i], q, ctx=ctx) for i in range(len(px))] for q in (90, 50, 10)}
rng = np.random.default_rng(1234)
noise = rng.integers(0, 256, (4096, 4096), dtype=np.uint8)
lenna = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (8, 8)))
big = [("noise q50", noise, 50), ("noise q90", noise, 90), ("noise q10", noise, 10), ("lenna q50", lenna, 50)]
dev = []
for name, img, q in big:
    s = np.frombuffer(T.compress(img, q, ctx=ctx), dtype=np.uint8)
    d_s, d_p = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, s.size + 64, C.byref(d_s)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_p)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_s, s.ctypes.data, s.size))
    os.environ["TIC_DECODE_HOST"] = "1"; want = T.decompress(s.tobytes(), ctx=ctx); del os.environ["TIC_DECODE_HOST"]
    dev.append((name, s, d_s, d_p, want))
rb, tr = C.c_int(), C.c_int()
for rule in rules:
    os.environ["TIC_DECODE_RULE"] = rule
    line = []
    for q in (90, 50, 10):
        td, second, bits = [], 0, set()
        for rep in range(3):
            for i, s in enumerate(streams[q]):
                t0 = time.perf_counter(); out = T.decompress(s, ctx=ctx); t1 = time.perf_counter()
                assert np.array_equal(out, T.decompress(s, ctx=ctx)) if rep == 0 else True
                if rep:
                    td.append(t1 - t0); L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr)); second += tr.value > 1; bits.add(rb.value)
        line.append("q%d %.1f us (%d second runs, ranges %s)" % (q, statistics.median(td) * 1e6, second, "/".join(str(b) for b in sorted(bits))))
    print("rule %-5s 512^2 decompress(): %s" % (rule, "; ".join(line)), flush=True)
    line = []
    for name, s, d_s, d_p, want in dev:
        f = lambda: ctx.check(L.tic_decompress_dev(ctx.handle, d_s, s.size, d_p, 4096, 4096 * 4096, None, None))
        for _ in range(5): f()
        L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr))
        t = time.perf_counter()
        for _ in range(50): f()
        us = (time.perf_counter() - t) / 50 * 1e6
        got = np.empty_like(want); ctx.check(L.tic_memcpy_d2h(ctx.handle, got.ctypes.data, d_p, got.size))
        line.append("%s %.1f us (%d bits, runs %d%s)" % (name, us, rb.value, tr.value, "" if np.array_equal(got, want) else ", PIXELS DIFFER"))
    print("rule %-5s 4096^2 tic_decompress_dev: %s" % (rule, "; ".join(line)), flush=True)
    if stress:
        r = subprocess.run([sys.executable, "tools/stress_decoder.py", str(stress)], capture_output=True, text=True, env=dict(os.environ))
        for ln in r.stdout.splitlines():
            if re.match(r"^\d+ streams|^valid streams of +(4|8|16|32|64) ", ln): print("    " + ln[:230], flush=True)
