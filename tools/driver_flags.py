#!/usr/bin/env python3
"""The driver's bench flags (--steps 20 --warmup 5) against the long interval (--steps 5000 --warmup 2000), in ONE process, with the
three ways of bracketing the timed launches:
  idle   : warm-up call, synchronise, then event / K launches / event (tic_dctq_dev_timed: rounds 1-5's bench line)
  warm   : W launches, event, K launches, event in one submission (tic_dctq_dev_timed_warm)
  preroll: the same with PRE (argv[2], default 32) more untimed launches in front of the W (the host is milliseconds ahead of the device at the first event)
  steps  : `warm` with a start and a stop event on every timed launch's own dispatch packet (hipExtLaunchKernelGGL): per-launch
           durations and gaps without marker packets in the queue
Each measurement is preceded by 60 ms of settling bursts, as in bench.py.  python tools/driver_flags.py [rounds]"""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyimgcodec_amd as T  # noqa: E402
from tinyimgcodec_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 32  # untimed launches in front of the W of the `preroll` form
L = N.load()
ctx = T.Context(0)
h = w = 4096
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, L.tic_num_blocks(h, w) * 128, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
ms = C.c_float()


def settle(ms_):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms_:
        ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, 2, 256, C.byref(ms)))


def idle(K, W):
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, 2, W, C.byref(ms)))
    ctx.check(L.tic_sync(ctx.handle))
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, w, 50, d_out, 2, K, C.byref(ms)))
    return ms.value * 1e3 / K, None


def warm(K, W):
    ctx.check(L.tic_sync(ctx.handle))
    ctx.check(L.tic_dctq_dev_timed_warm(ctx.handle, d_img, h, w, w, 50, d_out, 2, W, K, C.byref(ms), None))
    return ms.value * 1e3 / K, None


def steps(K, W):
    per = (C.c_float * (2 * K))()
    ctx.check(L.tic_sync(ctx.handle))
    ctx.check(L.tic_dctq_dev_timed_warm(ctx.handle, d_img, h, w, w, 50, d_out, 2, W, K, C.byref(ms), per))
    dur = [per[2 * i] * 1e3 for i in range(K)]
    end = [per[2 * i + 1] * 1e3 for i in range(K)]
    gap = [end[i] - end[i - 1] - dur[i] for i in range(1, K)]
    d, g = sorted(dur), sorted(gap)
    return ms.value * 1e3 / K, (d[0], d[len(d) // 2], d[-1], g[0], g[len(g) // 2], g[-1], end[-1] / K)


def preroll(K, W):
    return warm(K, W + PRE)


res = {}
for r in range(rounds):
    for name, fn in (("idle", idle), ("warm", warm), ("preroll", preroll), ("steps", steps)):
        for K, W in ((20, 5), (5000, 2000)):
            settle(60.0)
            us, per = fn(K, W)
            res.setdefault((name, K), []).append(us)
            print("round %d %-7s K=%-4d W=%-4d %7.3f us/step%s" % (r, name, K, W, us, "" if per is None else "   kernel min %.2f median %.2f max %.2f, gap min %.2f median %.2f max %.2f, first start .. last end / K %.3f" % per), flush=True)
print()
for (name, K), v in sorted(res.items()):
    print("%-7s K=%-4d median %7.3f  min %7.3f  max %7.3f us/step  (frac of 8 TB/s at the median %.3f)" % (name, K, statistics.median(v), min(v), max(v), 3.0 * h * w / statistics.median(v) / 8e6))
ctx.close()
