"""Minimal stand-in for the third-party `bitarray` package (absent from this image, no network).

TEST INFRASTRUCTURE ONLY - used by tests/golden/gen/make_goldens.py so that the *unmodified* reference at
/root/reference imports here.  A pure big-endian bit container following bitarray's documented semantics
(`extend` of '0'/'1' strings, 0/1 iterables or other bitarrays; `frombytes`; `tobytes` zero-pads the final
byte; slicing returns a bitarray; `invert`; `to01`).  No codec arithmetic lives in it.
"""


class bitarray:
    def __init__(self, initial=None, endian="big"):
        if endian != "big":
            raise NotImplementedError("stand-in supports endian='big' only")
        self._bits = []
        if initial is not None:
            self.extend(initial)

    # -- construction -------------------------------------------------------------------------------
    def extend(self, x):
        if isinstance(x, bitarray):
            self._bits.extend(x._bits)
        elif isinstance(x, str):
            for ch in x:
                if ch == "0":
                    self._bits.append(0)
                elif ch == "1":
                    self._bits.append(1)
                else:
                    raise ValueError("expected '0' or '1', got %r" % ch)
        else:
            for v in x:
                v = int(v)
                if v not in (0, 1):
                    raise ValueError("bit must be 0 or 1, got %r" % v)
                self._bits.append(v)

    def frombytes(self, data):
        for byte in bytes(data):
            for k in range(7, -1, -1):
                self._bits.append((byte >> k) & 1)

    # -- export -------------------------------------------------------------------------------------
    def tobytes(self):
        bits = self._bits + [0] * (-len(self._bits) % 8)
        out = bytearray()
        for i in range(0, len(bits), 8):
            b = 0
            for v in bits[i : i + 8]:
                b = (b << 1) | v
            out.append(b)
        return bytes(out)

    def to01(self):
        return "".join("1" if v else "0" for v in self._bits)

    def invert(self):
        self._bits = [1 - v for v in self._bits]

    # -- container protocol ---------------------------------------------------------------------------
    def __len__(self):
        return len(self._bits)

    def __getitem__(self, i):
        if isinstance(i, slice):
            r = bitarray()
            r._bits = self._bits[i]
            return r
        return self._bits[i]

    def __iter__(self):
        return iter(self._bits)

    def __eq__(self, other):
        return isinstance(other, bitarray) and self._bits == other._bits

    def __repr__(self):
        return "bitarray('%s')" % self.to01()
