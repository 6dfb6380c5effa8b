import os, sys
os.environ["TIC_TEST_HOOKS"] = "1"; os.environ["TIC_DECODE_TRACE"] = "1"
sys.path.insert(0, '.')
import numpy as np, ctypes as C
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
rng = np.random.default_rng(7)
for k in range(12):
    h, w = int(rng.integers(1200, 2600)), int(rng.integers(1200, 2600)); q = int(rng.integers(20, 90))
    kind = k % 3
    if kind == 0: img = (np.add.outer(np.arange(h), np.arange(w)) // int(rng.integers(1, 9)) % 256).astype(np.uint8)
    elif kind == 1: img = (rng.integers(0, 256, (h // 8 + 1, w // 8 + 1), dtype=np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w] ^ rng.integers(0, 4, (h, w), dtype=np.uint8)).astype(np.uint8)
    else: img = np.clip(rng.normal(128, 3, (h, w)), 0, 255).astype(np.uint8)
    s = T.compress(img, q, ctx=ctx)
    n = ((h + 7) // 8) * ((w + 7) // 8)
    print("kind %d %dx%d q=%d: %d bytes, %.1f bits per block" % (kind, h, w, q, len(s), len(s) * 8 / n), flush=True)
    sys.stderr.flush()
    T.decompress(s, ctx=ctx)
