#!/usr/bin/env python3
"""How the step time of the 4096^2 launch moves over the first 400 ms after an idle period: windows of 5 + 20 launches (the last 20 timed,
tic_dctq_dev_timed_warm), back to back, host clock at each window's end.  python tools/clock_profile.py [idle_ms] [total_ms]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyimgcodec_amd as T  # noqa: E402
from tinyimgcodec_amd import _native as N  # noqa: E402

idle_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 500.0
total_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
L = N.load()
ctx = T.Context(0)
h = w = 4096
img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
d_img, d_out = C.c_void_p(), C.c_void_p()
ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
ctx.check(L.tic_dev_alloc(ctx.handle, L.tic_num_blocks(h, w) * 128, C.byref(d_out)))
ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
ms = C.c_float()
for rep in range(3):
    time.sleep(idle_ms / 1e3)
    rows = []
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < total_ms:
        ctx.check(L.tic_dctq_dev_timed_warm(ctx.handle, d_img, h, w, w, 50, d_out, 2, 5, 20, C.byref(ms), None))
        rows.append(((time.perf_counter() - t0) * 1e3, ms.value * 1e3 / 20))
    print("## run %d: %d windows after %.0f ms idle" % (rep, len(rows), idle_ms))
    # averages over 10 ms bins, and the slowest window of each bin
    b, acc = 0, []
    for t, us in rows:
        while t >= (b + 1) * 10.0:
            if acc:
                print("  %4d-%4d ms: mean %6.2f  min %6.2f  max %6.2f us/step (%d windows)" % (b * 10, b * 10 + 10, sum(acc) / len(acc), min(acc), max(acc), len(acc)))
            acc, b = [], b + 1
        acc.append(us)
    sys.stdout.flush()
ctx.close()
