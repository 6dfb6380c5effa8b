"""Which forced range lengths (TIC_DECODE_RANGE, a test hook) hold on a stream, i.e. need no second run with the longest range.
Usage: python tools/range_probe.py <dim> <quality> <noise|lenna> r1 r2 ..."""
import os, sys
os.environ["TIC_TEST_HOOKS"] = "1"
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
dim, q, content = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
if content == "lenna":
    img = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (dim // 512, dim // 512)))
s = T.compress(img, q, ctx=ctx)
os.environ["TIC_DECODE_HOST"] = "1"
want = T.decompress(s, ctx=ctx)
os.environ.pop("TIC_DECODE_HOST")
rb, tr = C.c_int(), C.c_int()
print("%dx%d q=%d %s: %d bytes, %.1f bits per block" % (dim, dim, q, content, len(s), len(s) * 8 / (dim // 8) ** 2))
for r in sys.argv[4:]:
    os.environ["TIC_DECODE_RANGE"] = r
    got = T.decompress(s, ctx=ctx)
    L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr))
    print("  forced %5s: last run %4d bits, runs %d, path %d, giveup %d, pixels %s" % (r, rb.value, tr.value, L.tic_last_decode_path(ctx.handle), L.tic_last_decode_giveup(ctx.handle), "equal" if np.array_equal(got, want) else "DIFFER"))
