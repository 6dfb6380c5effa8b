"""The CPU oracle (oracle/tic_oracle.c) against fixtures produced by the unmodified reference
(tests/golden/gen/make_goldens.py).  CPU-only; this is what pins the oracle."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, rand_frame


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_dct_bit_patterns(oracle, golden):
    """utils.py:32-37 / scipy.fftpack.dct: float64 results bit-identical."""
    d = golden("dct_blocks")
    for blk, want in zip(d["blocks"], d["dct2_bits"]):
        assert np.array_equal(oracle.block_dct(blk).view(np.uint64), want)
    vec = d["vec_bits"].view(np.float64)
    for v, want in zip(vec, d["dct1_bits"]):
        assert np.array_equal(oracle.dct8(v).view(np.uint64), want)


def test_idct_bit_patterns(oracle, golden):
    """utils.py:40-45 / scipy.fftpack.idct."""
    d = golden("dct_blocks")
    vec = d["vec_bits"].view(np.float64)
    for v, want in zip(vec, d["idct1_bits"]):
        assert np.array_equal(oracle.idct8(v).view(np.uint64), want)
    for c, want in zip(d["icoef"], d["idct2_bits"]):
        assert np.array_equal(oracle.block_idct(c.astype(np.float64)).view(np.uint64), want)


def test_huffman_table_digest(oracle, manifest):
    """constants.py:53-242 == canonical Annex-K tables built by the oracle."""
    lines = sorted(l for l in oracle.dump_tables().split("\n") if l)
    assert len(lines) == manifest["huffman_table_digest"]["lines"] == 174
    assert sha(("\n".join(lines) + "\n").encode()) == manifest["huffman_table_digest"]["sha256"]


def test_small_shapes_coefficients_and_streams(oracle, golden):
    """Ragged/tiny shapes (reflect padding), constants, patterns: dc/ac int-exact, stream byte-exact."""
    d = golden("transform_small")
    for key in d["names"]:
        img = d[key + "_img"]
        q = int(str(key).rsplit("_q", 1)[1])
        dc, ac = oracle.encode(img, q)
        assert np.array_equal(dc, d[key + "_dc"]), key
        assert np.array_equal(ac, d[key + "_ac"]), key
        want = d[key + "_bs"].tobytes()
        if want:
            assert oracle.compress(img, q) == want, key
        else:  # reference raised KeyError: |coefficient| has no Huffman code
            with pytest.raises(oracle.OracleError):
                oracle.compress(img, q)


def test_quality_sweep(oracle, golden):
    d = golden("quality_sweep")
    img = d["img"]
    for q in d["qualities"]:
        q = int(q)
        dc, ac = oracle.encode(img, q)
        assert np.array_equal(dc, d[f"q{q}_dc"]), q
        assert np.array_equal(ac, d[f"q{q}_ac"]), q
        want = d[f"q{q}_bs"].tobytes()
        if want:
            assert oracle.compress(img, q) == want, q
        else:
            with pytest.raises(oracle.OracleError):
                oracle.compress(img, q)


def test_flat_banded_and_checkerboard_content(oracle, golden):
    """Tie-dense content (flat blocks of every grey level, flat / noise / checkerboard blocks side by side, a posterised ramp) at nine
    qualities: the reference's own encode() outputs (tests/golden/gen/make_goldens_r3b.py)."""
    d = golden("flat_blocks")
    for name in ("flat", "mix", "banded"):
        for q in d["qualities"]:
            q = int(q)
            dc, ac = oracle.encode(d[name], q)
            assert np.array_equal(dc, d[f"{name}_q{q}_dc"]), (name, q)
            assert np.array_equal(ac, d[f"{name}_q{q}_ac"]), (name, q)


def test_tie_blocks(oracle, golden):
    """Blocks whose DC sits on an exact .5 tie at q=50: rounding direction follows pocketfft's last ulp."""
    d = golden("tie_blocks")
    dc, ac = oracle.encode(d["img"], 50)
    assert np.array_equal(dc, d["dc"])
    assert np.array_equal(ac, d["ac"])
    assert oracle.compress(d["img"], 50) == d["bs"].tobytes()


def test_lenna(oracle, golden, manifest):
    d = golden("lenna")
    img = d["img"]
    assert sha(img.tobytes()) == manifest["lenna_pixels_sha256"]
    dc, ac = oracle.encode(img, 50)
    assert np.array_equal(dc, d["q50_dc"])
    assert np.array_equal(ac, d["q50_ac"].astype(np.int32))
    for q in (10, 50, 90):
        bs = oracle.compress(img, q)
        assert len(bs) == manifest[f"lenna_q{q}"]["bytes"]
        assert sha(bs) == manifest[f"lenna_q{q}"]["sha256"]
        assert bs == d[f"q{q}_bs"].tobytes()
    assert manifest["lenna_q50"]["bytes"] == 20765  # CR 12.62


def test_rle_known_answers(oracle, manifest):
    """huffman.py:12-33"""
    for name, ka in manifest["rle_known_answers"].items():
        got = oracle.rle_block(np.array(ka["seq"], dtype=np.int32))
        assert [list(t) for t in got] == ka["rle"], name


def test_decompress_small(oracle, golden):
    """codec.py:167-189 + 46-70: decoded pixels identical (exact IDCT order + truncating uint8 cast)."""
    d = golden("decode_small")
    s = golden("transform_small")
    for key in d["names"]:
        got = oracle.decompress(s[key + "_bs"].tobytes())
        assert np.array_equal(got, d[key]), key
    sw = golden("quality_sweep")
    for q in sw["qualities"]:
        bs = sw[f"q{int(q)}_bs"].tobytes()
        if bs:
            assert np.array_equal(oracle.decompress(bs), d[f"sweep_q{int(q)}"]), q
    t = golden("tie_blocks")
    assert np.array_equal(oracle.decompress(t["bs"].tobytes()), d["tie"])


def test_decompress_lenna(oracle, golden, manifest):
    d = golden("lenna")
    for q in (10, 50, 90):
        got = oracle.decompress(d[f"q{q}_bs"].tobytes())
        assert np.array_equal(got, d[f"q{q}_dec"])
        assert sha(got.tobytes()) == manifest[f"lenna_q{q}"]["decoded_sha256"]


@pytest.mark.parametrize("h,w", [(512, 512), (1080, 1920)])
def test_seeded_frames_stream_digest(oracle, manifest, h, w):
    m = manifest[f"rand1234_{h}x{w}_q50"]
    img = rand_frame(1234, h, w)
    dc, ac = oracle.encode(img, 50)
    assert sha(dc.astype("<i4").tobytes()) == m["dc_i4_sha256"]
    assert sha(ac.astype("<i4").tobytes()) == m["ac_i4_sha256"]
    bs = oracle.compress(img, 50)
    assert len(bs) == m["bytes"] and sha(bs) == m["sha256"]


@pytest.mark.parametrize("q", [10, 50, 90])
def test_4096_coefficient_digest(oracle, manifest, q):
    """BASELINE config 2 (4096x4096, seed 1234): coefficient digests of the reference's encode()."""
    m = manifest[f"rand1234_4096x4096_q{q}"]
    dc, ac = oracle.encode(rand_frame(1234, 4096, 4096), q)
    assert sha(dc.astype("<i4").tobytes()) == m["dc_i4_sha256"]
    assert sha(ac.astype("<i4").tobytes()) == m["ac_i4_sha256"]


def test_error_behaviour(oracle, manifest):
    """utils.py:50 / codec.py:103-108: valid quality domain is 1..99."""
    e = manifest["error_behaviour"]
    assert not e["quality_0"]["ok"] and not e["quality_100"]["ok"]
    img = rand_frame(3, 8, 8)
    for q in (0, 100, -5):
        with pytest.raises(oracle.OracleError):
            oracle.compress(img, q)
    assert e["empty_0x8"]["bytes"] == 16
    assert oracle.compress(np.zeros((0, 8), np.uint8), 50).hex() == e["empty_0x8"]["hex"]


def test_near_ties_round2(oracle, golden):
    """Round-2 fixture from the reference: exact ties of the irrational coefficients (2,2) (2,6) (6,2) (6,6), exact ties of the
    rational four, and random blocks with an irrational coefficient within 1e-6 of a tie (q=50) - the oracle follows the
    reference's float64 through all of them, at four qualities."""
    d = golden("near_ties")
    assert (d["kinds"] == "irrational_true_tie").sum() == 24 and (d["kinds"] == "near_tie_1e-6").sum() >= 64
    for q in (50, 90, 10, 37):
        dc, ac = oracle.encode(d["img"], q)
        assert np.array_equal(dc, d[f"q{q}_dc"]), q
        assert np.array_equal(ac, d[f"q{q}_ac"]), q


def test_wide_pixels_round2(oracle, golden):
    """Integer images outside 0..255 (the reference transforms any integers, codec.py:29): int16 with negatives, full-range
    int16, 12-bit uint16, float64 (truncated by astype(int32))."""
    d = golden("wide_pixels")
    for name in d["names"]:
        key, q = str(name).rsplit("_q", 1)
        dc, ac = oracle.encode_wide(d[key + "_img"], int(q))
        assert np.array_equal(dc, d[f"{name}_dc"]), name
        assert np.array_equal(ac, d[f"{name}_ac"]), name


def test_truncated_streams_round2(oracle, golden):
    """What the reference's decompress() returns for truncated / corrupted streams (it swallows the exception of a block,
    codec.py:178-186; reads past the end follow the bit container's slicing semantics): the oracle decodes the same pixels."""
    d = golden("truncated_streams")
    for name in d["names"]:
        name = str(name)
        assert int(d[name + "_ok"]) == 1, name  # the reference raised for none of them
        got = oracle.decompress(d[name + "_bs"].tobytes())
        assert np.array_equal(got, d[name + "_out"]), name


def _same_as_fixture(d, name, img):
    if name + "_img" in d:
        return np.array_equal(img, d[name + "_img"])
    return (tuple(img.shape) == tuple(d[name + "_shape"]) and np.array_equal(img[:32], d[name + "_rows"])
            and hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == str(d[name + "_sha"]))


def test_scaled_dct_streams_round2(oracle, golden):
    """decode()'s scaled_dct branch (codec.py:59-62, 127-128): streams of the reference's C encoder at its four qualities and
    Python streams re-flagged with small exponents, decoded by the reference (tests/golden/gen/make_goldens_scaled.py)."""
    d = golden("scaled_streams")
    for name in d["names"]:
        name = str(name)
        assert _same_as_fixture(d, name, oracle.decompress(d[name + "_bs"].tobytes())), name


def test_config4_stream_digests_round2(oracle):
    """BASELINE config 4, frames 1234..1237 (1920x1080): stream sizes and digests recorded from the reference."""
    import json
    import os

    from conftest import GOLDEN, rand_frame

    m = json.load(open(os.path.join(GOLDEN, "manifest_r2.json")))["entries"]
    f = rand_frame(1235, 1080, 1920)
    bs = oracle.compress(f, 50)
    e = m["rand1235_1080x1920_q50_stream"]
    assert len(bs) == e["bytes"] and hashlib.sha256(bs).hexdigest() == e["sha256"]


def test_numpy_scipy_restatement_equals_the_c_oracle(oracle):
    """oracle/np_encode.py (bench.py's "pure-Python-stack" CPU figure) runs on scipy's pocketfft - the reference's
    own arithmetic - and must equal the C restatement coefficient for coefficient, ragged shapes and the whole quality range included."""
    scipy = pytest.importorskip("scipy")  # noqa: F841
    from oracle import np_encode
    rng = np.random.default_rng(77)
    for h, w, q in ((1, 1, 50), (5, 13, 10), (64, 72, 50), (33, 47, 90), (200, 328, 1), (136, 520, 99), (256, 256, 37)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = np_encode.encode(img, q)
        dc, ac = oracle.encode(img, q)
        assert np.array_equal(got["dc"], dc) and np.array_equal(got["ac"], ac), (h, w, q)
    assert list(np_encode.ZIGZAG[:10]) == [0, 1, 8, 16, 9, 2, 3, 10, 17, 24]


def test_decoder_edges_round3(oracle, golden):
    """decoder_edges.npz (reference-generated): streams shorter than the header and streams with an embedded-table flag make
    the reference raise (the oracle reports an error for exactly those); the stream the reference's own adaptive-table writer
    produces and the same flag bytes in front of an ordinary payload decode to the reference's pixels."""
    g = golden("decoder_edges")
    for name in [str(n) for n in g["names"]]:
        bs = g[name + "_bs"].tobytes()
        if int(g[name + "_ok"]):
            assert np.array_equal(oracle.decompress(bs), g[name + "_out"]), name
        else:
            assert str(g[name + "_exc"]) in ("error", "ValueError"), name  # struct.error / ValueError
            with pytest.raises(oracle.OracleError):
                oracle.decompress(bs)


def test_oracle_against_full_size_manifest_r4(oracle):
    """Round 4: the reference's compress() on the 1920x1080 frames of BASELINE configs 3/4 (manifest_r4.json: size + sha256 per
    frame, seeds 1234 ...).  The oracle is pinned on a spread of them here (0.1 s each); the GPU tests check every frame."""
    import hashlib
    import json

    with open(os.path.join(GOLDEN, "manifest_r4.json")) as f:
        m = json.load(f)
    fr = m["frames"]
    assert (m["height"], m["width"], m["quality"]) == (1080, 1920, 50) and len(fr) >= 256
    assert [f["seed"] for f in fr] == list(range(1234, 1234 + len(fr)))
    sizes = np.asarray([f["bytes"] for f in fr], dtype="<i8")
    assert hashlib.sha256(sizes.tobytes()).hexdigest() == m["sizes_sha256"]
    assert hashlib.sha256(sizes[:256].tobytes()).hexdigest() == m["sizes_sha256_first256"]
    for k in sorted(set([0, 1, 100, 255, 256, len(fr) // 2, len(fr) - 1]) & set(range(len(fr)))):
        bs = oracle.compress(rand_frame(fr[k]["seed"], 1080, 1920), 50)
        assert len(bs) == fr[k]["bytes"] and hashlib.sha256(bs).hexdigest() == fr[k]["sha256"], fr[k]["seed"]


def test_oracle_non_integral_qualities(oracle, golden):
    """Round 4: encode() with a quality that is not an integer (utils.py:50-53 computes with any number): the oracle's divisors and
    coefficients equal the reference's on float_quality.npz (make_goldens_r4b.py: eight qualities x four images)."""
    d = golden("float_quality")
    for name in d["names"]:
        img = d["img_" + str(name)]
        for k, q in enumerate(d["qualities"]):
            dc, ac = oracle.encode(img, float(q))
            assert np.array_equal(dc, d["dc_%s_%d" % (name, k)]) and np.array_equal(ac, d["ac_%s_%d" % (name, k)]), (name, q)
    # an integral float is the integer quality
    img = d["img_rand_40x56"]
    a, b = oracle.encode(img, 37.0), oracle.encode(img, 37)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_oracle_on_the_references_own_benchmark_set(oracle):
    """Round 5: the reference's benchmark workload (/root/reference/tests/benchmark.py:12-23: data/1..49.gif x quality 90, 80, 50, 20, 10,
    5), generated by tests/golden/gen/make_goldens_r5.py from the unmodified reference.  The oracle's compress() gives the
    reference's bytes for all 294 pairs and its decompress() the reference's pixels: natural content at q = 5 / 20 / 80 is where
    the long zero runs, ZRL codes and tie-dense blocks live."""
    import hashlib
    import json

    with open(os.path.join(GOLDEN, "benchmark_set.json")) as f:
        m = json.load(f)
    px = np.load(os.path.join(GOLDEN, "benchmark_set.npz"))["pixels"]
    assert px.shape == (49, 512, 512) and hashlib.sha256(px.tobytes()).hexdigest() == m["pixels_sha256"]
    assert len(m["entries"]) == 294 and all(e["source"] == "reference" for e in m["entries"])
    assert sorted({e["quality"] for e in m["entries"]}) == [5, 10, 20, 50, 80, 90]
    for e in m["entries"]:
        bs = oracle.compress(px[e["image"] - 1], e["quality"])
        assert len(bs) == e["bytes"] and hashlib.sha256(bs).hexdigest() == e["sha256"], (e["image"], e["quality"])
        out = oracle.decompress(bs)
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == e["decoded_sha256"], (e["image"], e["quality"])
