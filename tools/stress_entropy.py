"""Stress run of the device entropy stage: random shapes, contents and qualities, tic_compress (device pack + place) against the host
coder on the same coefficients, back to back (catches timing-dependent errors a single launch may not show)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
ctx = T.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(77)
bad = 0
skipped = 0
t0 = time.time()
for it in range(iters):
    h = int(rng.integers(1, 2049)); w = int(rng.integers(1, 2049))
    if it % 7 == 0: h, w = (int(rng.integers(1, 9)) * 8, int(rng.integers(1, 300)) * 8)
    q = int(rng.integers(1, 100))
    kind = it % 5
    if kind == 0: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1: img = (np.add.outer(np.arange(h), np.arange(w)) // int(rng.integers(1, 9)) % 256).astype(np.uint8)
    elif kind == 2: img = (rng.integers(0, 2, (h, w), dtype=np.uint8) * int(rng.integers(1, 256))).astype(np.uint8)
    elif kind == 3: img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    else:
        img = np.full((h, w), 128, np.uint8)
        k = max(1, h * w // int(rng.integers(200, 20000)))
        img[rng.integers(0, h, k), rng.integers(0, w, k)] = rng.integers(0, 256, k)
    zz = T.dctq(img, q, ctx=ctx)
    try:
        want = T.entropy_encode(zz, h, w, q)
    except KeyError:
        skipped += 1
        continue
    got = T.compress(img, q, ctx=ctx)
    if got != want:
        bad += 1
        print("MISMATCH it=%d shape=%dx%d q=%d kind=%d len %d vs %d" % (it, h, w, q, kind, len(got), len(want)), flush=True)
    if it % 50 == 49: print("iter %d ok so far, %d s" % (it + 1, time.time() - t0), flush=True)
print("stress_entropy: %d iterations, %d without a Huffman code (skipped), %d mismatches" % (iters, skipped, bad))
sys.exit(1 if bad else 0)
