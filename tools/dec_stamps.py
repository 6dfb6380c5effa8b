"""Where the device decoder's two kernels spend their time on one stream: the 100 MHz clock at the phase boundaries, taken by the first and the last
wave / workgroup of a launch (a diagnostic build: make -C tools bin/libvar_700.so VARSRC=tic_entropy_dec_gpu.hip).
Usage: TIC_LIB=tools/bin/libvar_700.so python tools/dec_stamps.py [dim=512] [quality=50] [noise|lenna]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
N.LIB_PATH = N.HOOKS_LIB_PATH = os.environ['TIC_LIB']
L = N.load(); ctx = T.Context(0)
D = C.CDLL(os.environ['TIC_LIB'])
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
q = int(sys.argv[2]) if len(sys.argv) > 2 else 50
content = sys.argv[3] if len(sys.argv) > 3 else "lenna"
img = np.random.default_rng(1234).integers(0, 256, (dim, dim), dtype=np.uint8)
if content == "lenna":
    img = np.ascontiguousarray(np.tile(np.load('tests/golden/lenna.npz')['img'], (dim // 512, dim // 512)))
s = T.compress(img, q, ctx=ctx)
names_m = ["entry", "tables + window staged", "walk", "shuffle", "first stitch", "hand-over rounds", "sum + look-back", "positions written"]
names_f = ["entry", "tables + window staged", "DC symbol", "AC symbols", "look-back, DC", "columns", "rows + stores"]
acc = {}
reps = 20
for rep in range(reps + 3):
    out = T.decompress(s, ctx=ctx)
    st = (C.c_ulonglong * 64)()
    assert D.tic_debug_dec_stamps(st) == 0
    if rep < 3: continue
    for row, (tag, names) in enumerate((("measure, first wave", names_m), ("measure, last wave", names_m), ("fused, first workgroup", names_f), ("fused, last workgroup", names_f))):
        v = [st[row * 16 + k] for k in range(len(names))]
        acc.setdefault(tag, []).append([(v[k] - v[0]) / 100.0 for k in range(len(names))])
    acc.setdefault("measure first entry -> fused first entry", []).append([(st[32] - st[0]) / 100.0])
    acc.setdefault("measure first entry -> fused last end", []).append([(st[48 + 6] - st[0]) / 100.0])
assert np.array_equal(out, T.decompress(s, ctx=ctx))
print("%dx%d %s q=%d, %d bytes, %.1f bits per block; microseconds since the wave's entry, mean of %d calls" % (dim, dim, content, q, len(s), len(s) * 8 / (dim // 8) ** 2, reps))
for tag, rows in acc.items():
    m = np.mean(np.array(rows), axis=0)
    names = names_m if tag.startswith("measure,") else (names_f if tag.startswith("fused,") else [""])
    print("  %-42s %s" % (tag, "  ".join("%s %.2f" % (names[k], m[k]) for k in range(len(m)))))
