import sys, time
sys.path.insert(0, '.')
import numpy as np
import tinyimgcodec_amd as T
img = np.random.default_rng(1234).integers(0, 256, (4096, 4096), dtype=np.uint8)
s = T.compress(img, 50)
for f, name in ((lambda: T.decompress(s), "decompress(bytes) -> new array"), (lambda: T.compress(img, 50), "compress(array) -> bytes")):
    for _ in range(3): f()
    t = time.perf_counter()
    for _ in range(20): f()
    print("%-34s %.2f ms per call" % (name, (time.perf_counter() - t) / 20 * 1e3))
