#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE in this container.

Run only where /root/reference exists (the build container); the GPU box uses the committed fixtures.

    python tests/golden/gen/make_goldens.py [--big]      # --big adds the 16384x16384 sha256 goldens (slow, ~10 GB RAM)

How the reference is imported: `tinyimgcodec` needs the third-party containers `bidict` and `bitarray`, which
are not installed here (no network).  The two stand-ins under tests/golden/gen/standins/ provide exactly the
container surface the reference touches; all codec arithmetic runs in the reference's own
files + numpy/scipy.  Nothing from /root/reference is copied: only inputs and the reference's outputs are
stored (arrays, byte strings, sha256 digests).

Fixture files written:
  transform_small.npz   tiny/ragged shapes + constant images: pixels, dc, ac, bitstream (full content)
  quality_sweep.npz     64x96 random frame at many qualities: dc, ac, bitstream
  dct_blocks.npz        8x8 integer blocks -> float64 bit patterns of scipy's 2-D DCT (pins SURVEY Appendix A)
                        + 1-D float vectors -> float64 bit patterns of scipy dct/idct (non-integer inputs)
  tie_blocks.npz        blocks built to sit on exact .5 quantisation ties at q=50
  lenna.npz             Lenna pixels (reference data file data/lenna.gif -> "L"), dc/ac at q=50,
                        bitstreams at q=10/50/90, decoded images
  decode_small.npz      decompress() outputs for the small streams
  manifest.json         sha256 / sizes for everything incl. big frames (512^2, 1080p, 4096^2, optional 16384^2)
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.fftpack import dct, idct  # noqa: E402

import tinyimgcodec as ref  # noqa: E402  (the unmodified reference)
from tinyimgcodec.huffman import encode_run_length  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def rand_frame(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def enc(img, q):
    info = ref.encode(img, q)
    return info["dc"].astype(np.int32), info["ac"].astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    manifest = {
        "generator": "tests/golden/gen/make_goldens.py",
        "numpy": np.__version__,
        "scipy": scipy.__version__,
        "reference_version": ref.__version__,
        "entries": {},
    }
    E = manifest["entries"]

    # ---- 1. tiny / ragged shapes and constant images (full content) ---------------------------------------
    small = {}
    shapes = [(1, 1), (1, 9), (5, 13), (8, 8), (9, 9), (15, 17), (16, 24), (7, 64), (64, 7), (24, 40), (2, 3), (3, 2)]
    names = []
    for i, (h, w) in enumerate(shapes):
        img = rand_frame(7 + i, h, w)
        for q in (50, 10, 90):
            key = f"rand_{h}x{w}_q{q}"
            dc, ac = enc(img, q)
            bs = ref.compress(img, q, auto_generate_huffman_table=False)
            small[key + "_img"] = img
            small[key + "_dc"] = dc
            small[key + "_ac"] = ac
            small[key + "_bs"] = np.frombuffer(bs, dtype=np.uint8)
            names.append(key)
    for val in (0, 127, 128, 255, 1, 254):
        img = np.full((16, 24), val, dtype=np.uint8)
        key = f"const{val}_16x24_q50"
        dc, ac = enc(img, 50)
        bs = ref.compress(img, 50, auto_generate_huffman_table=False)
        small[key + "_img"] = img
        small[key + "_dc"] = dc
        small[key + "_ac"] = ac
        small[key + "_bs"] = np.frombuffer(bs, dtype=np.uint8)
        names.append(key)
    # gradients / checkerboards / extreme content (large AC magnitudes)
    yy, xx = np.mgrid[0:32, 0:48]
    patterns = {
        "hgrad": (xx * 255 // 47).astype(np.uint8),
        "vgrad": (yy * 255 // 31).astype(np.uint8),
        "checker1": (((xx + yy) & 1) * 255).astype(np.uint8),
        "checker8": ((((xx // 8) + (yy // 8)) & 1) * 255).astype(np.uint8),
        "stripes": ((xx & 1) * 255).astype(np.uint8),
        "binary": (rand_frame(99, 32, 48) > 127).astype(np.uint8) * 255,
    }
    for pname, img in patterns.items():
        for q in (50, 90, 97):
            key = f"pat_{pname}_q{q}"
            dc, ac = enc(img, q)
            try:
                bs = ref.compress(img, q, auto_generate_huffman_table=False)
            except KeyError:
                bs = b""  # |AC| >= 1024 has no Huffman code in the reference -> KeyError (recorded as empty)
            small[key + "_img"] = img
            small[key + "_dc"] = dc
            small[key + "_ac"] = ac
            small[key + "_bs"] = np.frombuffer(bs, dtype=np.uint8)
            names.append(key)
    small["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "transform_small.npz"), **small)

    # ---- 2. quality sweep ---------------------------------------------------------------------------------
    sweep = {}
    img = rand_frame(4321, 64, 96)
    sweep["img"] = img
    qs = [1, 2, 3, 5, 7, 10, 13, 20, 25, 33, 37, 49, 50, 51, 60, 75, 80, 90, 95, 98, 99]
    sweep["qualities"] = np.array(qs)
    for q in qs:
        dc, ac = enc(img, q)
        sweep[f"q{q}_dc"] = dc
        sweep[f"q{q}_ac"] = ac
        try:
            bs = ref.compress(img, q, auto_generate_huffman_table=False)
        except KeyError:
            bs = b""
        sweep[f"q{q}_bs"] = np.frombuffer(bs, dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "quality_sweep.npz"), **sweep)

    # ---- 3. DCT bit patterns --------------------------------------------------------------------------------
    rng = np.random.default_rng(2024)
    blocks = rng.integers(-128, 128, (512, 8, 8)).astype(np.int32)
    blocks[0] = -128
    blocks[1] = 127
    blocks[2] = 0
    blocks[3] = np.where((np.arange(64).reshape(8, 8) % 2) == 0, -128, 127)
    out = dct(dct(blocks, norm="ortho", axis=-2), axis=-1, norm="ortho")
    vec = rng.standard_normal((512, 8)) * 100.0
    vout = dct(vec, norm="ortho", axis=-1)
    coef = rng.standard_normal((512, 8, 8)) * 60.0
    coef_i = np.round(coef)
    iout = idct(idct(coef_i, norm="ortho", axis=-2), axis=-1, norm="ortho")
    viout = idct(vec, norm="ortho", axis=-1)
    np.savez_compressed(
        os.path.join(GOLD, "dct_blocks.npz"),
        blocks=blocks,
        dct2_bits=out.view(np.uint64),
        vec_bits=vec.view(np.uint64),
        dct1_bits=vout.view(np.uint64),
        icoef=coef_i.astype(np.int32),
        idct2_bits=iout.view(np.uint64),
        idct1_bits=viout.view(np.uint64),
    )

    # ---- 4. tie blocks: pixel sums that put the 4 rational coefficients on exact .5 ties at q=50 ---------------
    ties = []
    trng = np.random.default_rng(55)
    while len(ties) < 256:
        b = trng.integers(0, 256, (8, 8)).astype(np.int64)
        s = int((b - 128).sum())
        # DC = s/8 ; q=50 divisor 16 -> tie when s/128 = k + .5  <=> s mod 128 == 64
        need = (64 - s) % 128
        # nudge one pixel to land on the tie if possible
        i, j = trng.integers(0, 8, 2)
        v = b[i, j] + need
        if v > 255:
            v -= 128
        if 0 <= v <= 255:
            b[i, j] = v
            if int((b - 128).sum()) % 128 == 64:
                ties.append(b.astype(np.uint8))
    ties = np.stack(ties)  # [256,8,8]
    tie_img = ties.reshape(16, 16, 8, 8).swapaxes(1, 2).reshape(128, 128)
    tdc, tac = enc(tie_img, 50)
    tbs = ref.compress(tie_img, 50, auto_generate_huffman_table=False)
    np.savez_compressed(
        os.path.join(GOLD, "tie_blocks.npz"), img=tie_img, dc=tdc, ac=tac, bs=np.frombuffer(tbs, dtype=np.uint8)
    )

    # ---- 5. Lenna (reference data file) ------------------------------------------------------------------------
    from PIL import Image

    lenna = np.asarray(Image.open("/root/reference/data/lenna.gif").convert("L"))
    L = {"img": lenna}
    E["lenna_pixels_sha256"] = sha(lenna.tobytes())
    for q in (10, 50, 90):
        t0 = time.time()
        bs = ref.compress(lenna, q, auto_generate_huffman_table=False)
        L[f"q{q}_bs"] = np.frombuffer(bs, dtype=np.uint8)
        E[f"lenna_q{q}"] = {"bytes": len(bs), "sha256": sha(bs), "ref_compress_s": round(time.time() - t0, 3)}
        dec = ref.decompress(bs)
        L[f"q{q}_dec"] = dec
        E[f"lenna_q{q}"]["decoded_sha256"] = sha(dec.tobytes())
    dc, ac = enc(lenna, 50)
    L["q50_dc"] = dc
    L["q50_ac"] = ac.astype(np.int16)
    np.savez_compressed(os.path.join(GOLD, "lenna.npz"), **L)

    # ---- 6. decode goldens for the small streams ----------------------------------------------------------------
    D = {}
    dnames = []
    for key in names:
        bs = small[key + "_bs"].tobytes()
        if not bs:
            continue
        D[key] = ref.decompress(bs)
        dnames.append(key)
    for q in qs:
        bs = sweep[f"q{q}_bs"].tobytes()
        if bs:
            D[f"sweep_q{q}"] = ref.decompress(bs)
    D["tie"] = ref.decompress(tbs)
    D["names"] = np.array(dnames)
    np.savez_compressed(os.path.join(GOLD, "decode_small.npz"), **D)

    # ---- 7. RLE known answers --------------------------------------------------------------------------------
    rle = {}
    for name, spec in {
        "all_zero": {},
        "last_only": {62: -3},
        "at16": {16: 1},
        "at15": {15: 1},
        "at33": {33: 2},
        "dense": {i: (i % 5) - 2 for i in range(63)},
        "first_last": {0: 5, 62: 7},
    }.items():
        seq = np.zeros(63, dtype=np.int32)
        for k, v in spec.items():
            seq[k] = v
        rle[name] = {"seq": seq.tolist(), "rle": [[int(a), int(b)] for a, b in encode_run_length(seq)]}
    E["rle_known_answers"] = rle

    # ---- 7b. digest of the default Huffman tables (constants.py:53-242) in a canonical text form -----------------
    from tinyimgcodec.constants import AC, DC, HUFFMAN_CATEGORY_CODEWORD

    lines = ["D,0,%d,%s" % (cat, cw) for cat, cw in HUFFMAN_CATEGORY_CODEWORD[DC].items()]
    lines += ["A,%d,%d,%s" % (cat[0], cat[1], cw) for cat, cw in HUFFMAN_CATEGORY_CODEWORD[AC].items()]
    E["huffman_table_digest"] = {"lines": len(lines), "sha256": sha(("\n".join(sorted(lines)) + "\n").encode())}

    # ---- 8. big seeded frames: sha256 only ------------------------------------------------------------------------
    for (h, w) in ((512, 512), (1080, 1920)):
        img = rand_frame(1234, h, w)
        dc, ac = enc(img, 50)
        t0 = time.time()
        bs = ref.compress(img, 50, auto_generate_huffman_table=False)
        E[f"rand1234_{h}x{w}_q50"] = {
            "bytes": len(bs),
            "sha256": sha(bs),
            "dc_i4_sha256": sha(dc.astype("<i4").tobytes()),
            "ac_i4_sha256": sha(ac.astype("<i4").tobytes()),
            "ref_compress_s": round(time.time() - t0, 2),
        }
        print("done", h, w, E[f"rand1234_{h}x{w}_q50"], flush=True)
    img = rand_frame(1234, 4096, 4096)
    for q in (10, 50, 90):
        t0 = time.time()
        dc, ac = enc(img, q)
        dt = time.time() - t0
        E[f"rand1234_4096x4096_q{q}"] = {
            "dc_i4_sha256": sha(dc.astype("<i4").tobytes()),
            "ac_i4_sha256": sha(ac.astype("<i4").tobytes()),
            "ref_encode_s": round(dt, 3),
        }
    # frames of config 3/4 (seeds 1234+i): first 4 only, coefficient digests
    for i in range(4):
        img = rand_frame(1234 + i, 1080, 1920)
        dc, ac = enc(img, 50)
        E[f"rand{1234 + i}_1080x1920_q50_coeffs"] = {
            "dc_i4_sha256": sha(dc.astype("<i4").tobytes()),
            "ac_i4_sha256": sha(ac.astype("<i4").tobytes()),
        }
    if args.big:
        img = rand_frame(1234, 16384, 16384)
        for q in (10, 50, 90):
            t0 = time.time()
            dc, ac = enc(img, q)
            E[f"rand1234_16384x16384_q{q}"] = {
                "dc_i4_sha256": sha(dc.astype("<i4").tobytes()),
                "ac_i4_sha256": sha(ac.astype("<i4").tobytes()),
                "ref_encode_s": round(time.time() - t0, 2),
            }
            del dc, ac
            print("done 16384 q", q, flush=True)
    else:
        # keep previously generated big entries
        old = os.path.join(GOLD, "manifest.json")
        if os.path.exists(old):
            prev = json.load(open(old))["entries"]
            for k, v in prev.items():
                if k.startswith("rand1234_16384x16384"):
                    E[k] = v

    # ---- 9. reference error behaviour (recorded, not asserted here) ----------------------------------------------
    errs = {}
    probe = rand_frame(3, 8, 8)
    for label, fn in {
        "quality_0": lambda: ref.compress(probe, 0),
        "quality_100": lambda: ref.compress(probe, 100),
        "quality_float": lambda: ref.compress(probe, 50.0),
        "quality_negative": lambda: ref.compress(probe, -5),
        "ndim_3": lambda: ref.compress(np.zeros((8, 8, 3), np.uint8)),
        "ndim_1": lambda: ref.compress(np.zeros((8,), np.uint8)),
        "empty_0x8": lambda: ref.compress(np.zeros((0, 8), np.uint8)),
    }.items():
        try:
            r = fn()
            errs[label] = {"ok": True, "bytes": len(r), "hex": bytes(r).hex() if len(r) <= 64 else None}
        except Exception as e:  # noqa: BLE001
            errs[label] = {"ok": False, "exc": type(e).__name__}
    E["error_behaviour"] = errs

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote goldens to", GOLD)


if __name__ == "__main__":
    main()
