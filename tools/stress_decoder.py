"""Stress run of the device Huffman decoder: random shapes (>= 16,384 blocks), contents and qualities, valid and damaged streams;
tic_decompress with the device decoder against the host decoders on the same stream, back to back."""
import os, sys, time
os.environ["TIC_TEST_HOOKS"] = "1"
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load(); ctx = T.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4242)
bad = dev = host_fallback = short = 0
path_counts = {}  # valid streams by log2(bits per block): [streams, second runs, left to the host]
second_runs = {0: 0, 1: 0, 2: 0}  # streams whose first choice of range did not hold, by kind of damage (0 = valid stream)
rb, tr = C.c_int(), C.c_int()
t0 = time.time()
for it in range(iters):
    h = int(rng.integers(1024, 3000)); w = int(rng.integers(1100, 3000))
    q = int(rng.integers(5, 97))
    kind = it % 5
    if kind == 0: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1: img = (np.add.outer(np.arange(h), np.arange(w)) // int(rng.integers(1, 9)) % 256).astype(np.uint8)
    elif kind == 2: img = (rng.integers(0, 2, (h, w), dtype=np.uint8) * int(rng.integers(1, 256))).astype(np.uint8)
    elif kind == 3: img = (rng.integers(0, 256, (h // 8 + 1, w // 8 + 1), dtype=np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w] ^ rng.integers(0, 4, (h, w), dtype=np.uint8)).astype(np.uint8)
    else: img = np.clip(rng.normal(128, int(rng.integers(2, 60)), (h, w)), 0, 255).astype(np.uint8)
    try:
        s = bytearray(T.compress(img, q, ctx=ctx))
    except KeyError:
        continue
    dmg = it % 4
    if dmg == 1: s[int(rng.integers(16, len(s)))] ^= 1 << int(rng.integers(0, 8))
    elif dmg == 2: s = s[: int(len(s) * rng.uniform(0.3, 0.99))]
    s = bytes(s)
    os.environ.pop("TIC_DECODE_HOST", None)
    a = T.decompress(s, ctx=ctx)
    path = L.tic_last_decode_path(ctx.handle)
    L.tic_last_decode_range(ctx.handle, C.byref(rb), C.byref(tr))
    if dmg in (0, 3) and path_counts is not None:
        bpb = len(s) * 8 / (((h + 7) // 8) * ((w + 7) // 8))
        bk = min(int(np.log2(max(bpb, 1))), 9)
        path_counts.setdefault(bk, [0, 0, 0]); path_counts[bk][0] += 1; path_counts[bk][1] += tr.value > 1; path_counts[bk][2] += path == 2
    if tr.value > 1:
        second_runs[dmg if dmg < 3 else 0] += 1
        print("second run: it", it, h, w, "q", q, "kind", kind, "damage", dmg, "bits per block %.0f" % (len(s) * 8 / (((h + 7) // 8) * ((w + 7) // 8))), "path", path, flush=True)
        if dmg in (0, 3):  # a valid stream: why (the runs' give-up bits on stderr)
            os.environ["TIC_DECODE_TRACE"] = "1"; sys.stdout.flush(); T.decompress(s, ctx=ctx); sys.stderr.flush(); os.environ.pop("TIC_DECODE_TRACE")
    os.environ["TIC_DECODE_HOST"] = "1"
    b = T.decompress(s, ctx=ctx)
    os.environ["TIC_DECODE_SERIAL"] = "1"
    c = T.decompress(s, ctx=ctx)
    os.environ.pop("TIC_DECODE_SERIAL"); os.environ.pop("TIC_DECODE_HOST")
    dev += path == 1
    nblk = ((h + 7) // 8) * ((w + 7) // 8)
    takes = (nblk >= 16384 and len(s) * 8 >= 128 + (1 << 21)) or (nblk >= 1024 and len(s) * 8 >= 128 + (1 << 13))
    if path == 2 and takes: host_fallback += 1
    elif path == 2: short += 1
    if not (np.array_equal(a, b) and np.array_equal(b, c)):
        bad += 1
        print("MISMATCH it", it, h, w, q, kind, dmg, "path", path, flush=True)
print("%d streams (%d x %d .. ), device decoder on %d, given up to the host on %d long ones, %d too short; mismatches %d; second runs (longest range, or the 2,048-bit margin): %d on valid streams, %d on streams with a flipped bit, %d on cut streams; %.0f s" % (iters, 1024, 1100, dev, host_fallback, short, bad, second_runs[0], second_runs[1], second_runs[2], time.time() - t0))
for bk in sorted(path_counts):
    print("valid streams of %4d .. %4d bits per block: %3d, second runs %3d, left to the host decoder %3d" % (1 << bk, (2 << bk) - 1, *path_counts[bk]))
sys.exit(1 if bad else 0)
