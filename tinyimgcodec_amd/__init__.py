"""tinyimgcodec_amd - MI355X-native drop-in for tinyimgcodec's encode/decode/compress/decompress.

Mirrors tinyimgcodec/__init__.py:1-5 of the reference (same four names); see codec.py for the mapping.
"""
from ._native import Context, NativeError, NativeUnavailable
from .codec import compress, compress_batch, dctq, decode, decompress, decompress_batch, encode, entropy_encode, parse_header

__version__ = "0.1.0"
__all__ = ["encode", "decode", "compress", "decompress"]
