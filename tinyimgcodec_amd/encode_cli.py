#!/usr/bin/env python3
"""Command-line counterpart of the reference's encode.py (encode.py:1-19): image in, .img stream out.

    python -m tinyimgcodec_amd.encode_cli input.(gif|png|jpg|npy|raw) output.img [--quality 50] [--shape H W]

Prints "<n> bytes" and "Compression Ratio: <w*h/n>:1" exactly as the reference does.  Inputs: anything Pillow
opens (converted to "L" as the reference does), a .npy array, or headerless 8-bit gray (.raw with --shape) so that
a box without Pillow can still feed it.  Runs on the MI355X path (no CPU fallback)."""
import argparse
import sys

import numpy as np


def load_gray(path, shape=None):
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".raw"):
        if shape is None:
            raise SystemExit(".raw input needs --shape H W")
        return np.fromfile(path, dtype=np.uint8).reshape(shape)
    from PIL import Image

    return np.asarray(Image.open(path).convert("L"))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--quality", type=int, default=50)
    ap.add_argument("--shape", type=int, nargs=2, metavar=("H", "W"))
    args = ap.parse_args(argv)
    from . import compress

    im = load_gray(args.input, args.shape)
    out = compress(im, args.quality, auto_generate_huffman_table=False)
    byte_size = len(out)
    print(f"{byte_size} bytes")
    print(f"Compression Ratio: {im.shape[1] * im.shape[0] / byte_size}:1")
    with open(args.output, "wb") as f:
        f.write(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
