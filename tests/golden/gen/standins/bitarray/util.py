"""Stand-in for `bitarray.util` (see package docstring): big-endian int <-> bitarray helpers."""
from . import bitarray


def ba2int(a):
    if len(a) == 0:
        raise ValueError("non-empty bitarray expected")
    n = 0
    for v in a:
        n = (n << 1) | v
    return n


def int2ba(n, length=None):
    n = int(n)
    if n < 0:
        raise OverflowError("unsigned integer expected")
    s = bin(n)[2:]
    if length is not None:
        if len(s) > length:
            raise OverflowError("int too large to fit in %d bits" % length)
        s = s.rjust(length, "0")
    return bitarray(s)
