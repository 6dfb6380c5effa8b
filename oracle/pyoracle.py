"""ctypes binding of oracle/libtic_oracle.so (the C restatement of the reference's Python codec).

TEST INFRASTRUCTURE ONLY: the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtic_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "tic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtic_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i32p, i16p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_int32, C.c_int16, C.c_double))
        L.tico_dct8.argtypes = [f64p]
        L.tico_idct8.argtypes = [f64p]
        L.tico_block_dct.argtypes = [i32p, f64p]
        L.tico_block_idct.argtypes = [f64p, f64p]
        L.tico_divisors.argtypes = [C.c_int, f64p]
        L.tico_encode.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, i32p, i32p]
        L.tico_encode_i32.argtypes = [i32p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, i32p, i32p]
        L.tico_encode_zz16.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, i16p]
        L.tico_rle_block.argtypes = [i32p, i32p, i32p]
        L.tico_entropy_encode.argtypes = [i32p, i32p, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.tico_compress.argtypes = [u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.tico_decompress.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.tico_parse_header.argtypes = [u8p, C.c_size_t] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_uint32)]
        L.tico_compress_bound.argtypes = [C.c_int, C.c_int]
        L.tico_compress_bound.restype = C.c_size_t
        L.tico_dump_tables.argtypes = [C.c_char_p, C.c_size_t]
        L.tico_dump_tables.restype = C.c_size_t
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _img(image):
    a = np.ascontiguousarray(np.asarray(image).astype(np.int32).astype(np.uint8))
    if a.ndim != 2:
        raise ValueError("2-D image expected")
    return a


def nblocks(h, w):
    return 0 if h == 0 or w == 0 else ((h + 7) // 8) * ((w + 7) // 8)


def dct8(v):
    a = np.array(v, dtype=np.float64)
    lib().tico_dct8(_p(a, C.c_double))
    return a


def idct8(v):
    a = np.array(v, dtype=np.float64)
    lib().tico_idct8(_p(a, C.c_double))
    return a


def block_dct(block):
    i = np.ascontiguousarray(block, dtype=np.int32).reshape(64)
    o = np.empty(64, dtype=np.float64)
    lib().tico_block_dct(_p(i, C.c_int32), _p(o, C.c_double))
    return o.reshape(8, 8)


def block_idct(block):
    i = np.ascontiguousarray(block, dtype=np.float64).reshape(64)
    o = np.empty(64, dtype=np.float64)
    lib().tico_block_idct(_p(i, C.c_double), _p(o, C.c_double))
    return o.reshape(8, 8)


def divisors(quality):
    d = np.empty(64, dtype=np.float64)
    rc = lib().tico_divisors(int(quality), _p(d, C.c_double))
    if rc:
        raise OracleError(rc)
    return d.reshape(8, 8)


def encode(image, quality=50):
    """-> (dc int32[N], ac int32[N,63]) exactly as reference encode() (codec.py:26-43)."""
    a = _img(image)
    h, w = a.shape
    n = nblocks(h, w)
    dc = np.zeros(max(n, 1), dtype=np.int32)
    ac = np.zeros((max(n, 1), 63), dtype=np.int32)
    if isinstance(quality, float) and quality != int(quality):  # utils.py:50-53 on a float that is not an integer
        L = lib()
        L.tico_encode_f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_ssize_t, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        rc = L.tico_encode_f(_p(a, C.c_uint8), h, w, a.strides[0] if a.size else w, float(quality), _p(dc, C.c_int32), _p(ac, C.c_int32))
    else:
        rc = lib().tico_encode(_p(a, C.c_uint8), h, w, a.strides[0] if a.size else w, int(quality), _p(dc, C.c_int32), _p(ac, C.c_int32))
    if rc:
        raise OracleError(rc)
    return dc[:n], ac[:n]


def encode_wide(image, quality=50):
    """encode() for integer images with values outside 0..255 (codec.py:29: astype(int32) - 128) -> (dc, ac)."""
    a = np.ascontiguousarray(np.asarray(image).astype(np.int32))
    if a.ndim != 2:
        raise ValueError("2-D image expected")
    h, w = a.shape
    n = nblocks(h, w)
    dc = np.zeros(max(n, 1), dtype=np.int32)
    ac = np.zeros((max(n, 1), 63), dtype=np.int32)
    rc = lib().tico_encode_i32(_p(a, C.c_int32), h, w, (a.strides[0] // 4) if a.size else w, int(quality), _p(dc, C.c_int32), _p(ac, C.c_int32))
    if rc:
        raise OracleError(rc)
    return dc[:n], ac[:n]


def encode_zz16(image, quality=50):
    """-> int16 [N,64] zig-zag coefficients, DC not differenced (the HIP kernel's output layout)."""
    a = _img(image)
    h, w = a.shape
    n = nblocks(h, w)
    zz = np.zeros((max(n, 1), 64), dtype=np.int16)
    rc = lib().tico_encode_zz16(_p(a, C.c_uint8), h, w, a.strides[0] if a.size else w, int(quality), _p(zz, C.c_int16))
    if rc:
        raise OracleError(rc)
    return zz[:n]


def rle_block(ac63):
    a = np.ascontiguousarray(ac63, dtype=np.int32)
    runs = np.zeros(64, dtype=np.int32)
    vals = np.zeros(64, dtype=np.int32)
    n = lib().tico_rle_block(_p(a, C.c_int32), _p(runs, C.c_int32), _p(vals, C.c_int32))
    return [(int(runs[i]), int(vals[i])) for i in range(n)]


def entropy_encode(dc, ac, h, w, quality):
    dc = np.ascontiguousarray(dc, dtype=np.int32)
    ac = np.ascontiguousarray(ac, dtype=np.int32)
    cap = lib().tico_compress_bound(h, w)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().tico_entropy_encode(_p(dc, C.c_int32), _p(ac, C.c_int32), h, w, int(quality), _p(out, C.c_uint8), cap, C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[: n.value].tobytes()


def compress(image, quality=50):
    a = _img(image)
    h, w = a.shape
    cap = lib().tico_compress_bound(h, w)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().tico_compress(_p(a, C.c_uint8), h, w, a.strides[0] if a.size else w, int(quality), _p(out, C.c_uint8), cap, C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[: n.value].tobytes()


def decompress(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    h, w, q = C.c_int(), C.c_int(), C.c_int()
    flag = C.c_uint32()
    rc = lib().tico_parse_header(_p(buf, C.c_uint8), buf.size, C.byref(h), C.byref(w), C.byref(q), C.byref(flag))
    if rc:
        raise OracleError(rc)
    out = np.zeros((h.value, w.value), dtype=np.uint8)
    rc = lib().tico_decompress(_p(buf, C.c_uint8), buf.size, _p(out, C.c_uint8), out.size)
    if rc:
        raise OracleError(rc)
    return out


def dump_tables():
    b = C.create_string_buffer(1 << 16)
    n = lib().tico_dump_tables(b, len(b))
    return b.raw[:n].decode()
