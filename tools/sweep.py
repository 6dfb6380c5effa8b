"""BASELINE configs 3 and 5 on one MI355X (run on the GPU box): prints one JSON line per measurement.

  config 5: single 16384x16384 frame, q in {10,50,90}: kernel time, GB/s vs roofline, digest check vs the
            reference goldens (tests/golden/manifest.json).
  config 3: 256 x 1080p frames: (a) transform kernel only, frames resident in HBM; (b) the stream-overlapped
            pipeline incl. H2D/D2H over PCIe (tic_dctq_batch); (c) whole compress incl. host entropy stage.
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyimgcodec_amd as T  # noqa: E402
from tinyimgcodec_amd import _native as N  # noqa: E402

L = N.load()
ctx = T.Context(0)
manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["entries"]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def timed(d_img, h, w, pitch, q, d_out, iters):
    ms = C.c_float()
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, N.KERNEL_HYBRID, 2, C.byref(ms)))
    ctx.check(L.tic_dctq_dev_timed(ctx.handle, d_img, h, w, pitch, q, d_out, N.KERNEL_HYBRID, iters, C.byref(ms)))
    return ms.value / iters


def config5():
    h = w = 16384
    img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size * 2, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    for q in (10, 50, 90):
        ms = timed(d_img, h, w, w, q, d_out, 10)
        zz = np.empty((h * w // 64, 64), np.int16)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, zz.ctypes.data, d_out, zz.nbytes))
        dc = zz[:, 0].astype(np.int32)
        dc[1:] = np.diff(zz[:, 0].astype(np.int32))
        ok = sha(dc.astype("<i4").tobytes()) == manifest[f"rand1234_16384x16384_q{q}"]["dc_i4_sha256"] and sha(
            zz[:, 1:].astype("<i4").tobytes()
        ) == manifest[f"rand1234_16384x16384_q{q}"]["ac_i4_sha256"]
        gbs = 3.0 * h * w / (ms * 1e-3) / 1e9
        print(json.dumps({"config": 5, "frame": "16384x16384", "quality": q, "kernel_us": round(ms * 1e3, 2),
                          "Mpix_s": round(h * w / (ms * 1e-3) / 1e6, 1), "GB_s_3Bpx": round(gbs, 1),
                          "frac_of_8TBs": round(gbs / 8000, 4), "bit_exact_vs_reference_digest": bool(ok)}), flush=True)
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)


def config3(n=256):
    h, w = 1080, 1920
    frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
    px = float(n) * h * w
    # (a) kernel only, all frames resident in HBM, one launch per frame back to back
    pitch = 2048
    host = np.zeros((n, h, pitch), np.uint8)
    for i, f in enumerate(frames):
        host[i, :, :w] = f
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * 32400 * 128, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, host.ctypes.data, host.size))
    for rep in range(2):
        ctx.check(L.tic_sync(ctx.handle))
        t0 = time.perf_counter()
        for i in range(n):
            ctx.check(L.tic_dctq_dev(ctx.handle, d_img.value + i * h * pitch, h, w, pitch, 50,
                                     d_out.value + i * 32400 * 128, N.KERNEL_HYBRID))
        ctx.check(L.tic_sync(ctx.handle))
        dt = time.perf_counter() - t0
    print(json.dumps({"config": 3, "what": "transform kernels only, 256 x 1080p resident in HBM, one launch per frame",
                      "ms_total": round(dt * 1e3, 3), "Mpix_s": round(px / dt / 1e6, 1),
                      "GB_s_3Bpx": round(3 * px / dt / 1e9, 1)}), flush=True)
    # (a2) the same 256 resident frames in ONE launch (grid row per frame)
    ms = C.c_float()
    for rep in range(3):
        ctx.check(L.tic_sync(ctx.handle))
        t0 = time.perf_counter()
        ctx.check(L.tic_dctq_dev_frames(ctx.handle, d_img, n, h, w, pitch, h * pitch, 50, d_out, 32400 * 128, N.KERNEL_HYBRID))
        ctx.check(L.tic_sync(ctx.handle))
        dt = time.perf_counter() - t0
    zz0 = np.empty((32400, 64), np.int16)
    ctx.check(L.tic_memcpy_d2h(ctx.handle, zz0.ctypes.data, d_out.value + 255 * 32400 * 128, zz0.nbytes))
    print(json.dumps({"config": 3, "what": "transform stage, 256 x 1080p resident in HBM, ONE batched launch",
                      "ms_total": round(dt * 1e3, 3), "Mpix_s": round(px / dt / 1e6, 1),
                      "GB_s_3Bpx": round(3 * px / dt / 1e9, 1), "frac_of_8TBs": round(3 * px / dt / 8e12, 4)}), flush=True)
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)
    # (b) pipeline with H2D + kernel + D2H overlapped on two streams (PCIe-inclusive)
    inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    zz = [np.empty((32400, 64), np.int16) for _ in range(n)]
    outp = (C.c_void_p * n)(*[z.ctypes.data for z in zz])
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.check(L.tic_dctq_batch(ctx.handle, inp, n, h, w, w, 50, outp))
        dt = time.perf_counter() - t0
    print(json.dumps({"config": 3, "what": "stream-overlapped pipeline incl. PCIe H2D (1 B/px) + D2H (2 B/px), host buffers",
                      "ms_total": round(dt * 1e3, 2), "Mpix_s": round(px / dt / 1e6, 1),
                      "PCIe_GB_s": round(3 * px / dt / 1e9, 2)}), flush=True)
    # (c) whole compress incl. host entropy coding on worker threads
    for threads in (16, 0):
        t0 = time.perf_counter()
        out = T.compress_batch(frames, 50, threads=threads, ctx=ctx)
        dt = time.perf_counter() - t0
        ok = sha(out[0]) == manifest["rand1234_1080x1920_q50"]["sha256"]
        print(json.dumps({"config": 3, "what": ("compress_batch: pipeline + host Huffman on %d threads" % threads) if threads else "compress_batch: pipeline + device entropy stage",
                          "ms_total": round(dt * 1e3, 1), "Mpix_s": round(px / dt / 1e6, 1),
                          "frame0_stream_matches_reference": bool(ok)}), flush=True)


def config3_capi(n=256):
    """C-ABI level timing of tic_compress_batch (no Python object handling): outputs preallocated and pre-touched."""
    h, w = 1080, 1920
    frames = [np.random.default_rng(1234 + i).integers(0, 256, (h, w), dtype=np.uint8) for i in range(n)]
    px = float(n) * h * w
    cap = 1 << 20  # 1 MiB per frame is ample for these streams (890 KB); tic_compress_bound() is the safe size
    pool = np.zeros((n, cap), np.uint8)
    inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
    caps = (C.c_size_t * n)(*([cap] * n))
    lens = (C.c_size_t * n)()
    for threads in (16, 0):
        for rep in range(3):
            t0 = time.perf_counter()
            ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, 50, outp, caps, lens, threads))
            dt = time.perf_counter() - t0
        ok = sha(pool[0][: lens[0]].tobytes()) == manifest["rand1234_1080x1920_q50"]["sha256"]
        print(json.dumps({"config": 3, "what": "tic_compress_batch (C-ABI), 256 x 1080p host frames -> host streams, "
                          + ("host Huffman on %d threads" % threads if threads else "device entropy stage"),
                          "ms_total": round(dt * 1e3, 2), "Mpix_s": round(px / dt / 1e6, 1), "frames_per_s": round(n / dt, 1),
                          "frame0_stream_matches_reference": bool(ok)}), flush=True)


def config2_compress():
    """BASELINE config 2 frame through the whole codec, image and stream resident in HBM (tic_compress_dev)."""
    h = w = 4096
    img = np.random.default_rng(1234).integers(0, 256, (h, w), dtype=np.uint8)
    cap = L.tic_compress_bound(h, w)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    nlen = C.c_size_t()
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, 50, d_out, cap, C.byref(nlen)))
        dt = time.perf_counter() - t0
    print(json.dumps({"config": 2, "what": "tic_compress_dev: transform + device entropy stage, 4096x4096 resident in HBM",
                      "ms": round(dt * 1e3, 3), "Mpix_s": round(h * w / dt / 1e6, 1), "stream_bytes": nlen.value}), flush=True)
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["5", "3"]
    if "5" in which:
        config5()
    if "3" in which:
        config3()
        config3_capi()
    if "2" in which or len(sys.argv) == 1:
        config2_compress()
