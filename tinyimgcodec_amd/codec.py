"""Host-side mirror of the reference's Python API (tinyimgcodec/codec.py) on top of the C-ABI.

Same names, argument meaning and error behaviour as the reference so that it is a drop-in:

    compress(image, quality=50, auto_generate_huffman_table=False) -> bytes      codec.py:133-164
    decompress(data) -> np.ndarray[uint8]                                        codec.py:167-189
    encode(image, quality=50) -> {"height","width","quality","dc","ac"}          codec.py:26-43
    decode(data) -> np.ndarray[uint8]                                            codec.py:46-70

The transform stage (pad, level shift, DCT, quantise, zig-zag) and the entropy stage (DPCM, run lengths, Huffman codes, bit
packing; the Huffman decode and the inverse transform of decompress()) run in hand-written gfx950 kernels; the library's host
entropy coder (C++) serves encode()-style callers that hold coefficients on the host, and streams the device decoder hands back
(damaged or very short ones).  There is no Python or CPU fallback for the device path: without the HIP library and an MI355X these
functions raise tinyimgcodec_amd.NativeUnavailable.

Documented differences from the reference (all outside its working domain):
  * integer pixel values outside 0..255 are transformed as the reference does (exact float64 path on the device) by
    encode() and compress(); the batch and device-layout helpers (compress_batch, dctq) are 8-bit only;
  * encode()/decode() accept float qualities like the reference, integral or not (utils.py:50-53 computes with any number: the
    constants of a non-integral quality in [1, 99] are built per call, tic_set_custom_quality); qualities below 1 or negative (for
    which the reference computes with huge or negative divisors) raise ValueError;
  * auto_generate_huffman_table=True raises NotImplementedError (that path is broken in the reference:
    the table flag is written big-endian and read little-endian, codec.py:111/119);
  * quality > 100 raises ValueError (the reference produces streams with negative divisors);
  * streams whose little-endian flag word has bit 31 set (an embedded Huffman table) raise ValueError in decompress() - what the
    reference raises for every such stream that does not hold a well-formed table; scaled_dct streams (the reference's C encoder) are
    decoded for exponents 0..62 (the C encoder writes 0..3).
"""
import ctypes as C
import struct

import numpy as np

from . import _native as N


def _ctx(ctx):
    return ctx if ctx is not None else N.default_context()


def _check_quality(quality, packs_header):
    """Validates `quality` the way the reference would fail and returns it as an int.

    packs_header=True  (compress): the reference evaluates `5000 / quality` in encode() and then struct.pack("III", ...) in
                       make_header (codec.py:103-108): floats and negatives end in struct.error.
    packs_header=False (encode, dctq, decode): nothing is packed; the reference computes with whatever number it gets.
                       Integral floats give the same divisors as the int and are returned as the int; a non-integral quality
                       in [1, 99] is returned as the float it is (the caller installs its constants: _q_arg); negative values
                       and values below 1 raise ValueError - a documented difference."""
    if isinstance(quality, (bool, np.bool_)):
        quality = int(quality)
    if isinstance(quality, (float, np.floating)):
        if quality == 0:
            raise ZeroDivisionError("float division by zero")  # utils.py:50
        if packs_header:
            struct.pack("I", quality)  # raises struct.error: required argument is not an integer
        if quality != quality:
            raise ValueError("quality is not a number")
        if quality != int(quality):
            if not (1.0 <= quality <= 99.0):
                raise ValueError("non-integral quality %r outside 1..99 is not supported by the MI355X path" % (quality,))
            return float(quality)
        quality = int(quality)
    elif not isinstance(quality, (int, np.integer)):
        raise TypeError("quality must be a number")
    quality = int(quality)
    if quality == 0:
        raise ZeroDivisionError("division by zero")  # utils.py:50
    if quality < 0:
        if packs_header:
            struct.pack("I", quality)  # struct.error, codec.py:103-108
        raise ValueError("negative quality is not supported by the MI355X path")
    if quality == 100:
        if packs_header:
            raise KeyError((0, 0))  # factor 0 -> inf/NaN coefficients -> no Huffman code (huffman.py:62)
        raise ValueError("quality 100 makes every divisor zero (the reference returns inf/NaN garbage)")
    if quality > 100:
        raise ValueError("quality must be in 1..99")
    return quality


def _q_arg(ctx, q):
    """The quality argument of a transform / inverse entry point: the integer itself, or - for a non-integral quality, whose
    constants are installed in the context's spare slot first - TIC_QUALITY_CUSTOM.  Call with ctx.lock held."""
    if isinstance(q, float):
        ctx.check(N.load().tic_set_custom_quality(ctx.handle, q))
        return N.QUALITY_CUSTOM
    return q


def _as_image(image):
    """-> (array, height, width, wide).  uint8-representable images go to the 8-bit device path as uint8; integer images with
    values outside 0..255 (the reference transforms any integers after astype(int32), codec.py:29) as int32 (`wide`)."""
    image = np.asarray(image)
    height, width = image.shape  # ValueError for non 2-D input, as codec.py:27
    if image.dtype == np.uint8:  # the common case needs no conversion pass
        return np.ascontiguousarray(image), int(height), int(width), False
    a = image.astype(np.int32)  # codec.py:29 (truncation of floats, as the reference)
    if a.size and (a.min() < 0 or a.max() > 255):
        return np.ascontiguousarray(a), int(height), int(width), True
    return np.ascontiguousarray(a.astype(np.uint8)), int(height), int(width), False


def _as_u8_image(image):
    img, h, w, wide = _as_image(image)
    if wide:
        raise ValueError("pixel values outside 0..255: use encode()/compress() (the int16 device layout of dctq()/compress_batch() is 8-bit only)")
    return img, h, w


def _encode_wide(img, h, w, q, ctx):
    L = N.load()
    n = L.tic_num_blocks(h, w)
    dc = np.zeros(n, dtype=np.int32)
    ac = np.zeros((n, 63), dtype=np.int32)
    if n:
        with ctx.lock:
            ctx.check(L.tic_encode_wide(ctx.handle, img.ctypes.data, h, w, img.strides[0] // 4, _q_arg(ctx, q), dc.ctypes.data, ac.ctypes.data))
    return dc, ac


def dctq(image, quality=50, ctx=None):
    """Transform stage in the device layout: int16 [N, 64], zig-zag order, DC not differenced."""
    img, h, w = _as_u8_image(image)
    quality = _check_quality(quality, packs_header=False)
    ctx = _ctx(ctx)
    n = N.load().tic_num_blocks(h, w)
    zz = np.zeros((n, 64), dtype=np.int16)
    if n:
        with ctx.lock:
            ctx.check(N.load().tic_dctq(ctx.handle, img.ctypes.data, h, w, img.strides[0], _q_arg(ctx, quality), zz.ctypes.data))
    return zz


def encode(image, quality=50, ctx=None):
    img, h, w, wide = _as_image(image)
    q = _check_quality(quality, packs_header=False)
    ctx = _ctx(ctx)
    if wide:
        dc, ac = _encode_wide(img, h, w, q, ctx)
        return {"height": h, "width": w, "quality": quality, "dc": dc, "ac": ac}
    n = N.load().tic_num_blocks(h, w)
    dc = np.zeros(n, dtype=np.int32)
    ac = np.zeros((n, 63), dtype=np.int32)
    if n:
        with ctx.lock:
            ctx.check(N.load().tic_encode(ctx.handle, img.ctypes.data, h, w, img.strides[0], _q_arg(ctx, q), dc.ctypes.data, ac.ctypes.data))
    return {"height": h, "width": w, "quality": quality, "dc": dc, "ac": ac}


def compress(image, quality=50, auto_generate_huffman_table=False, ctx=None):
    img, h, w, wide = _as_image(image)
    q = _check_quality(quality, packs_header=True)
    if auto_generate_huffman_table:
        raise NotImplementedError("auto_generate_huffman_table=True is not supported (broken in the reference)")
    ctx = _ctx(ctx)
    L = N.load()
    if wide:  # integer pixels outside 0..255: exact float64 transform on the device, host entropy coder
        dc, ac = _encode_wide(img, h, w, q, ctx)
        zz = np.empty((dc.shape[0], 64), dtype=np.int64)
        zz[:, 0] = np.cumsum(dc.astype(np.int64))
        zz[:, 1:] = ac
        if zz.size and np.abs(zz).max() > 32767:
            raise KeyError("coefficient magnitude has no Huffman code")  # (>= 1024 already has none in the reference)
        return entropy_encode(zz.astype(np.int16), h, w, q)
    cap = L.tic_compress_bound(h, w)
    # worst-case sized landing buffer kept on the context: a fresh 50 MB mapping per call would be faulted in page by page
    # under the device-to-host copy (20+ ms for a 4096x4096 frame whose whole C-level round trip takes 0.6 ms)
    # (the buffer belongs to the context: the context's lock is held from the call to the copy into the returned bytes)
    with ctx.lock:
        out = getattr(ctx, "_out_buf", None)
        if out is None or out.size < cap:
            out = ctx._out_buf = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = L.tic_compress(ctx.handle, img.ctypes.data, h, w, img.strides[0] if img.size else max(w, 1), q, out.ctypes.data, cap, C.byref(n))
        if rc == N.TIC_E_RANGE:
            raise KeyError("coefficient magnitude has no Huffman code")  # as the reference's dict lookup
        ctx.check(rc)
        return out[: n.value].tobytes()


def compress_batch(images, quality=50, threads=0, ctx=None, devices=None):
    """Batch of equally sized frames through the stream-overlapped pipeline -> list of bytes (frame order).

    threads=0: entropy stage on the GPU; threads>0: host entropy coder on that many worker threads.
    devices=[0, 1, ...]: the batch is cut into contiguous shards, one per listed device, and every shard runs its own pipeline on its
    own context and host thread inside this process (tic_compress_batch_multi; the GIL is released for the whole call) - the
    multi-GPU form of a loop over images (/root/reference/tests/benchmark.py:12-23) without a launcher.  A device may be listed
    twice (two pipelines on one GPU)."""
    q = _check_quality(quality, packs_header=True)
    frames = [_as_u8_image(im) for im in images]
    if not frames:
        return []
    if devices is not None:
        return _compress_batch_multi(frames, q, int(threads), list(devices))
    ctx = _ctx(ctx)
    h, w = frames[0][1], frames[0][2]
    if any((f[1], f[2]) != (h, w) for f in frames):
        raise ValueError("all frames of a batch must have the same shape")
    L = N.load()
    n = len(frames)
    cap = L.tic_compress_bound(h, w)
    # one mapping, kept on the context: only the bytes actually written are ever touched, and a second batch of the same
    # geometry finds them already faulted in (first-touch page faults cost more than the whole GPU pipeline)
    with ctx.lock:
        pool = getattr(ctx, "_batch_pool", None)
        if pool is None or pool.shape[0] < n or pool.shape[1] != cap:
            pool = ctx._batch_pool = np.empty((n, cap), dtype=np.uint8)
        outs = [pool[i] for i in range(n)]
        inp = (C.c_void_p * n)(*[f[0].ctypes.data for f in frames])
        outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*([cap] * n))
        lens = (C.c_size_t * n)()
        rc = L.tic_compress_batch(ctx.handle, inp, n, h, w, max(w, 1), q, outp, caps, lens, int(threads))
        if rc == N.TIC_E_RANGE:
            raise KeyError("coefficient magnitude has no Huffman code")
        ctx.check(rc)
        return [outs[i][: lens[i]].tobytes() for i in range(n)]


def _compress_batch_multi(frames, q, threads, devices):
    if not devices:
        raise ValueError("devices must name at least one device")
    ctxs = N.device_contexts(devices)
    h, w = frames[0][1], frames[0][2]
    if any((f[1], f[2]) != (h, w) for f in frames):
        raise ValueError("all frames of a batch must have the same shape")
    L = N.load()
    n = len(frames)
    cap = L.tic_compress_bound(h, w)
    pool = np.empty((n, cap), dtype=np.uint8)
    handles = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    inp = (C.c_void_p * n)(*[f[0].ctypes.data for f in frames])
    outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
    caps = (C.c_size_t * n)(*([cap] * n))
    lens = (C.c_size_t * n)()
    failed = C.c_int(-1)
    locked = sorted(ctxs, key=id)  # (one global order: two threads that list the same devices in different orders cannot deadlock)
    for c in locked:
        c.lock.acquire()
    try:
        rc = L.tic_compress_batch_multi(handles, len(ctxs), inp, n, h, w, max(w, 1), q, outp, caps, lens, threads, C.byref(failed))
        if rc == N.TIC_E_RANGE:
            raise KeyError("coefficient magnitude has no Huffman code")
        if rc != N.TIC_OK:
            ctxs[max(failed.value, 0)].check(rc)
    finally:
        for c in reversed(locked):
            c.lock.release()
    return [pool[i, : lens[i]].tobytes() for i in range(n)]


def entropy_encode(coeffs_zz, height, width, quality):
    """Host entropy stage alone (no GPU needed): int16 [N,64] zig-zag coefficients -> stream bytes."""
    L = N.load()
    zz = np.ascontiguousarray(coeffs_zz, dtype=np.int16)
    cap = L.tic_compress_bound(height, width)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = L.tic_entropy_encode(zz.ctypes.data, int(height), int(width), int(quality), out.ctypes.data, cap, C.byref(n))
    if rc == N.TIC_E_RANGE:
        raise KeyError("coefficient magnitude has no Huffman code")
    if rc != N.TIC_OK:
        raise N.NativeError(rc, "tic_entropy_encode failed")
    return out[: n.value].tobytes()


def _as_bytes_view(data):
    """uint8 view of a bytes-like object without copying it (bytes, bytearray, memoryview, uint8 arrays)."""
    if isinstance(data, np.ndarray) and data.dtype == np.uint8 and data.flags.c_contiguous:
        return data.reshape(-1)
    try:
        return np.frombuffer(data, dtype=np.uint8)
    except (TypeError, ValueError):
        return np.frombuffer(bytes(data), dtype=np.uint8)


def parse_header(data):
    L = N.load()
    buf = _as_bytes_view(data)
    h, w, q, flag = C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
    rc = L.tic_parse_header(buf.ctypes.data, buf.size, C.byref(h), C.byref(w), C.byref(q), C.byref(flag))
    if rc != N.TIC_OK:
        raise struct.error("unpack requires a buffer of 16 bytes")  # codec.py:119
    return {"height": h.value, "width": w.value, "quality": q.value, "flag": flag.value}


def decompress(data, ctx=None):
    ctx = _ctx(ctx)
    buf = _as_bytes_view(data)  # one view for the header and the payload: the stream is not copied
    hdr = parse_header(buf)     # struct.error for fewer than 16 bytes, as codec.py:118-119
    if hdr["flag"] & (1 << 31):
        # codec.py:124-126 parses a Huffman table from the stream here.  The reference's own writer cannot produce such a stream
        # (it writes the flag MSB-first and reads it back little-endian: codec.py:111/119), and on anything that is not a
        # well-formed table its read_huffman_table ends in ValueError (tests/golden/decoder_edges.npz); embedded tables are out
        # of scope here, so every such stream gets that exception
        raise ValueError("stream carries an embedded Huffman table (little-endian flag bit 31): not supported")
    out = np.zeros((hdr["height"], hdr["width"]), dtype=np.uint8)
    with ctx.lock:
        ctx.check(N.load().tic_decompress(ctx.handle, buf.ctypes.data, buf.size, out.ctypes.data, out.size))
    return out


def decompress_batch(streams, ctx=None):
    """decompress() of many streams in one call -> list of uint8 arrays (stream order): the mirror of compress_batch, i.e. the loop of the
    reference's benchmark (/root/reference/tests/benchmark.py:12-23: `decompress(data)` per image) as ONE call.  Every element is what
    decompress() returns for that stream; the exceptions are decompress()'s (the first offending stream's, before anything is decoded).
    The streams go up in one copy, two kernel launches decode all of them, the pixels come down in one copy (tic_decompress_batch)."""
    ctx = _ctx(ctx)
    bufs = [_as_bytes_view(d) for d in streams]
    n = len(bufs)
    if n == 0:
        return []
    hdrs = [parse_header(b) for b in bufs]  # struct.error for fewer than 16 bytes, as codec.py:118-119
    for hd in hdrs:
        if hd["flag"] & (1 << 31):
            raise ValueError("stream carries an embedded Huffman table (little-endian flag bit 31): not supported")
    # one block for all frames: equal geometries follow each other in memory, and the library copies the pixels of a chunk straight into it
    sizes = [max(hd["height"], 0) * max(hd["width"], 0) for hd in hdrs]
    block = np.zeros(sum(sizes), dtype=np.uint8)
    outs, at = [], 0
    for hd, sz in zip(hdrs, sizes):
        outs.append(block[at:at + sz].reshape(max(hd["height"], 0), max(hd["width"], 0)) if sz else np.zeros((max(hd["height"], 0), max(hd["width"], 0)), np.uint8))
        at += sz
    L = N.load()
    sp = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sl = (C.c_size_t * n)(*[b.size for b in bufs])
    op = (C.c_void_p * n)(*[o.ctypes.data if o.size else None for o in outs])
    oc = (C.c_size_t * n)(*[o.size for o in outs])
    with ctx.lock:
        ctx.check(L.tic_decompress_batch(ctx.handle, sp, sl, n, op, oc, None, None))
    return outs


def decode(data, ctx=None):
    """decode() of the reference: dict with height, width, quality, scaled_dct, dc (DPCM'd), ac."""
    ctx = _ctx(ctx)
    height, width, quality = data["height"], data["width"], data["quality"]
    scaled = bool(data["scaled_dct"])
    dc = np.cumsum(np.asarray(data["dc"], dtype=np.int64))  # codec.py:53
    ac = np.asarray(data["ac"])
    n = N.load().tic_num_blocks(int(height), int(width))
    if dc.shape[0] != n or ac.shape != (n, 63):
        raise ValueError("dc/ac shapes do not match the image geometry")
    if n and (np.abs(dc).max() > 32767 or np.abs(ac).max(initial=0) > 32767):
        raise ValueError("coefficients exceed the int16 device layout")
    zz = np.empty((n, 64), dtype=np.int16)
    zz[:, 0] = dc
    zz[:, 1:] = ac
    out = np.zeros((int(height), int(width)), dtype=np.uint8)
    if scaled:  # codec.py:59-62: the quality field is an exponent, the inverse quantiser runs at quality 50
        if int(quality) != quality or not (0 <= int(quality) <= 62):
            raise ValueError("scaled_dct exponent outside 0..62")
        if n:
            with ctx.lock:
                ctx.check(N.load().tic_idctq_scaled(ctx.handle, zz.ctypes.data, int(height), int(width), int(quality), out.ctypes.data, out.size))
        return out
    q = _check_quality(quality, packs_header=False)
    if n:
        with ctx.lock:
            ctx.check(N.load().tic_idctq(ctx.handle, zz.ctypes.data, int(height), int(width), _q_arg(ctx, q), out.ctypes.data, out.size))
    return out
