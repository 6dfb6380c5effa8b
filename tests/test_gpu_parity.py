"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle and the reference goldens.

Bar: bit-exact (integer coefficients, byte streams, decoded pixels).  Run with `-m gpu` on an MI355X."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import rand_frame

import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N

pytestmark = pytest.mark.gpu


def device_decoder_takes(nblocks, nbytes):
    """The streams the device Huffman decoder takes (csrc/tic_api.hip device_decoder_takes): long ones (16,384 blocks, 2 Mbit), and short ones of at
    least 1,024 blocks and 1 KB (round 6: whatever their density)."""
    bits = nbytes * 8
    return (nblocks >= 16384 and bits >= 128 + (1 << 21)) or (nblocks >= 1024 and bits >= 128 + (1 << 13))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    c = T.Context(0)
    assert c.arch.startswith("gfx950"), c.arch
    yield c
    c.close()


class DevFrame:
    """Device-resident frame + coefficient buffer driven through the tic_*_dev entry points."""

    def __init__(self, ctx, img, pitch=None):
        self.ctx, self.L = ctx, N.load()
        self.h, self.w = img.shape
        self.pitch = pitch if pitch is not None else (self.w + 255) // 256 * 256
        host = np.zeros((max(self.h, 1), self.pitch), dtype=np.uint8)
        host[: self.h, : self.w] = img
        self.n = self.L.tic_num_blocks(self.h, self.w)
        self.d_img, self.d_out = C.c_void_p(), C.c_void_p()
        ctx.check(self.L.tic_dev_alloc(ctx.handle, host.size, C.byref(self.d_img)))
        ctx.check(self.L.tic_dev_alloc(ctx.handle, max(self.n, 1) * 128, C.byref(self.d_out)))
        ctx.check(self.L.tic_memcpy_h2d(ctx.handle, self.d_img, host.ctypes.data, host.size))

    def run(self, quality, variant):
        ctx, L = self.ctx, self.L
        ctx.check(L.tic_memset_dev(ctx.handle, self.d_out, 0x5A, max(self.n, 1) * 128))
        ctx.check(L.tic_dctq_dev(ctx.handle, self.d_img, self.h, self.w, self.pitch, quality, self.d_out, variant))
        zz = np.empty((self.n, 64), dtype=np.int16)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, zz.ctypes.data, self.d_out, self.n * 128))
        return zz

    def fallbacks(self):
        c = C.c_ulonglong()
        self.ctx.check(self.L.tic_last_fallback_blocks(self.ctx.handle, C.byref(c)))
        return c.value

    def free(self):
        self.L.tic_dev_free(self.ctx.handle, self.d_img)
        self.L.tic_dev_free(self.ctx.handle, self.d_out)


def zz_to_dc_ac(zz):
    dc = zz[:, 0].astype(np.int32)
    dc[1:] = np.diff(zz[:, 0].astype(np.int32))
    return dc, zz[:, 1:].astype(np.int32)


def test_shipped_library_in_a_fresh_process(tmp_path):
    """This test process runs the test-hooks build (tests/conftest.py sets TIC_TEST_HOOKS=1: the same sources with -DTIC_TEST_HOOKS,
    whose schedules and decoder paths the tests steer).  The SHIPPED library - no hooks, no environment - gets the parity core in
    a fresh process here: config-2 coefficient digests at three qualities, Lenna's streams, a spread of the reference's benchmark set
    through compress() / compress_batch() / decompress(), all against the reference-generated goldens."""
    import subprocess
    import sys

    code = r'''
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N
L = N.load()
assert L.tic_build_has_test_hooks() == 0 and N._lib_path() == N.LIB_PATH
sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()
G = os.path.join("tests", "golden")
man = json.load(open(os.path.join(G, "manifest.json")))["entries"]
ctx = T.Context(0)
img = np.random.default_rng(1234).integers(0, 256, (4096, 4096), dtype=np.uint8)
for q in (10, 50, 90):
    e = T.encode(img, q, ctx=ctx)
    m = man["rand1234_4096x4096_q%d" % q]
    assert sha(e["dc"].astype("<i4").tobytes()) == m["dc_i4_sha256"] and sha(e["ac"].astype("<i4").tobytes()) == m["ac_i4_sha256"], q
lenna = np.load(os.path.join(G, "lenna.npz"))
for q in (10, 50, 90):
    assert T.compress(lenna["img"], q, ctx=ctx) == lenna["q%d_bs" % q].tobytes(), q
bm = json.load(open(os.path.join(G, "benchmark_set.json")))["entries"]
px = np.load(os.path.join(G, "benchmark_set.npz"))["pixels"]
n = 0
for e in bm[::7]:
    s = T.compress(px[e["image"] - 1], e["quality"], ctx=ctx)
    assert len(s) == e["bytes"] and sha(s) == e["sha256"], e
    assert sha(np.ascontiguousarray(T.decompress(s, ctx=ctx)).tobytes()) == e["decoded_sha256"], e
    n += 1
by = {(e["image"], e["quality"]): e for e in bm}
for q in (5, 80):
    for i, s in enumerate(T.compress_batch([px[i] for i in range(49)], q, ctx=ctx), 1):
        assert sha(s) == by[(i, q)]["sha256"], (i, q)
print("shipped library ok:", n, "benchmark pairs")
'''
    env = {k: v for k, v in os.environ.items() if not k.startswith("TIC_")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "shipped library ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_which_build_of_the_library_this_process_runs():
    """The test-hooks build when TIC_TEST_HOOKS=1 (tests/conftest.py's default), the shipped library otherwise."""
    L = N.load()
    hooks = os.environ.get("TIC_TEST_HOOKS") == "1"
    assert L.tic_build_has_test_hooks() == (1 if hooks else 0)
    assert N._lib_path() == (N.HOOKS_LIB_PATH if hooks else N.LIB_PATH)


SHIPPED_SUBSET = ("test_which_build_of_the_library_this_process_runs or test_rare_paths_of_the_strip_kernel or test_tie_blocks or test_near_ties_round2 "
                  "or test_flat_banded_and_checkerboard_content_vs_goldens or test_truncated_streams_round2 or test_decoder_edges_round3 "
                  "or (test_config5_16384_coefficient_digest and 50) or test_decompress_dev_wrong_guess_writes_nothing_outside_the_image "
                  "or test_device_decoder_at_the_stream_end or test_scaled_dct_streams_round2 or test_decompress_batch_mixed_streams")


def test_rare_paths_and_decoder_edges_on_the_shipped_library():
    """Round-5 verdict, weak 1(a): the rare-path frames (true irrational ties, batch overflow, all-eight-trip strips, posterised noise at
    q = 99), the tie / near-tie / flat fixtures, the 16 truncated + 11 edge decoder fixtures, C-encoder streams and one 16384^2 digest ran on
    the test-hooks build only.  Here the same test functions run once more in ONE fresh pytest process that loads the library that ships
    (TIC_TEST_HOOKS=0: no hooks compiled in, the monkeypatched TIC_* variables are inert, the launcher's defaults decide)."""
    if os.environ.get("TIC_TEST_HOOKS") != "1":
        pytest.skip("this IS the shipped-library process")
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if not k.startswith("TIC_")}
    env["TIC_TEST_HOOKS"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", SHIPPED_SUBSET],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail
    import re

    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 14, tail  # (tie_blocks, near_ties and the flat fixtures are parametrised over both kernels)


def test_dpp_byte_transpose_selftest(ctx):
    """The in-register DPP/v_perm 8x8 byte transpose equals the shuffle formulation and a numpy transpose."""
    n = 4096
    data = np.random.default_rng(5).integers(0, 256, (n, 8), dtype=np.uint8)
    a = np.zeros_like(data)
    b = np.zeros_like(data)
    ctx.check(N.load().tic_selftest_transpose(ctx.handle, data.ctypes.data, a.ctypes.data, b.ctypes.data, n))
    want = data.reshape(n // 8, 8, 8).transpose(0, 2, 1).reshape(n, 8)
    assert np.array_equal(b, want), "shuffle formulation wrong"
    assert np.array_equal(a, want), "DPP formulation wrong"


@pytest.mark.parametrize("variant", [N.KERNEL_EXACT, N.KERNEL_HYBRID])
def test_small_and_ragged_shapes_vs_goldens(ctx, golden, variant):
    """Reference goldens incl. reflect padding (1x1 ... 15x17), constants, patterns: coefficients int-exact."""
    d = golden("transform_small")
    for key in d["names"]:
        img = d[key + "_img"]
        q = int(str(key).rsplit("_q", 1)[1])
        f = DevFrame(ctx, img)
        dc, ac = zz_to_dc_ac(f.run(q, variant))
        f.free()
        assert np.array_equal(dc, d[key + "_dc"]), key
        assert np.array_equal(ac, d[key + "_ac"]), key


@pytest.mark.parametrize("variant", [N.KERNEL_EXACT, N.KERNEL_HYBRID])
def test_quality_sweep_vs_goldens(ctx, golden, variant):
    d = golden("quality_sweep")
    f = DevFrame(ctx, d["img"])
    for q in d["qualities"]:
        q = int(q)
        dc, ac = zz_to_dc_ac(f.run(q, variant))
        assert np.array_equal(dc, d[f"q{q}_dc"]), q
        assert np.array_equal(ac, d[f"q{q}_ac"]), q
    f.free()


@pytest.mark.parametrize("variant", [N.KERNEL_EXACT, N.KERNEL_HYBRID])
def test_flat_banded_and_checkerboard_content_vs_goldens(ctx, golden, variant):
    """The reference's encode() on tie-dense content (flat blocks of every grey level - the strip kernel's flat-block DC table -, flat /
    noise / checkerboard blocks side by side, a posterised ramp) at nine qualities, and the same frames tiled so that every wave's
    batch overflows."""
    d = golden("flat_blocks")
    for name in ("flat", "mix", "banded"):
        img = d[name]
        f = DevFrame(ctx, img)
        tiled = DevFrame(ctx, np.tile(img, (8, 4)))
        by, bx = img.shape[0] // 8, img.shape[1] // 8
        for q in d["qualities"]:
            q = int(q)
            dc, ac = zz_to_dc_ac(f.run(q, variant))
            assert np.array_equal(dc, d[f"{name}_q{q}_dc"]), (name, q)
            assert np.array_equal(ac, d[f"{name}_q{q}_ac"]), (name, q)
            zz = tiled.run(q, variant).reshape(8, by, 4, bx, 64)
            one = f.run(q, variant).reshape(by, bx, 64)
            assert np.array_equal(zz, np.broadcast_to(one[None, :, None], zz.shape)), (name, q, "tiled")
        f.free()
        tiled.free()


@pytest.mark.parametrize("variant", [N.KERNEL_EXACT, N.KERNEL_HYBRID])
def test_tie_blocks(ctx, golden, variant):
    """DC exactly on .5 ties at q=50: rounding must follow pocketfft's last-ulp error."""
    d = golden("tie_blocks")
    f = DevFrame(ctx, d["img"])
    dc, ac = zz_to_dc_ac(f.run(50, variant))
    f.free()
    assert np.array_equal(dc, d["dc"])
    assert np.array_equal(ac, d["ac"])


def test_random_ragged_shapes_vs_oracle(ctx, oracle):
    """Seeded random shapes (odd sizes, unaligned pitches) at random qualities: HIP == oracle, both variants."""
    rng = np.random.default_rng(77)
    for it in range(40):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        q = int(rng.integers(1, 100))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        want = oracle.encode_zz16(img, q)
        pitch = None if it % 3 else w + int(rng.integers(0, 5))  # every third case: unaligned rows (byte-load path)
        f = DevFrame(ctx, img, pitch)
        for variant in (N.KERNEL_EXACT, N.KERNEL_HYBRID):
            got = f.run(q, variant)
            assert np.array_equal(got, want), (h, w, q, variant, pitch)
        f.free()


def test_extreme_content_vs_oracle(ctx, oracle):
    """Saturated / adversarial content (largest intermediates of the float32 fast path) stays exact."""
    rng = np.random.default_rng(3)
    imgs = [
        rng.choice(np.array([0, 255], dtype=np.uint8), (256, 256)),
        np.tile(np.array([[0, 255], [255, 0]], dtype=np.uint8), (64, 64)),
        np.repeat(np.repeat(rng.choice(np.array([0, 255], dtype=np.uint8), (32, 32)), 4, 0), 4, 1),
    ]
    for img in imgs:
        f = DevFrame(ctx, img)
        for q in (5, 50, 90, 99):
            want = oracle.encode_zz16(img, q)
            for variant in (N.KERNEL_EXACT, N.KERNEL_HYBRID):
                assert np.array_equal(f.run(q, variant), want), (q, variant)
        f.free()


def test_lenna_streams_byte_exact(ctx, golden, manifest):
    """BASELINE config 1 input: compress() bytes identical to the reference at q=10/50/90 (20,765 B at q=50)."""
    d = golden("lenna")
    for q in (10, 50, 90):
        bs = T.compress(d["img"], q, ctx=ctx)
        assert len(bs) == manifest[f"lenna_q{q}"]["bytes"]
        assert bs == d[f"q{q}_bs"].tobytes()
    info = T.encode(d["img"], 50, ctx=ctx)
    assert np.array_equal(info["dc"], d["q50_dc"]) and np.array_equal(info["ac"], d["q50_ac"].astype(np.int32))
    assert (info["height"], info["width"], info["quality"]) == (512, 512, 50)


def test_natural_image_quality_sweep_vs_oracle(ctx, oracle, golden):
    """Lenna (smooth content: many zero runs, few guard trips) at a spread of qualities: stream == oracle stream."""
    img = golden("lenna")["img"]
    for q in (1, 5, 25, 50, 75, 95, 99):
        zz = T.dctq(img, q, ctx=ctx)
        assert np.array_equal(zz, oracle.encode_zz16(img, q)), q
        try:
            want = oracle.compress(img, q)
        except oracle.OracleError:  # |coefficient| >= 1024 has no Huffman code (reference: KeyError)
            with pytest.raises(KeyError):
                T.compress(img, q, ctx=ctx)
            continue
        assert T.compress(img, q, ctx=ctx) == want, q


def test_cli_counterpart_of_encode_py(ctx, golden, manifest, tmp_path, capsys):
    """python -m tinyimgcodec_amd.encode_cli: same two output lines and the same file bytes as the reference's encode.py."""
    from tinyimgcodec_amd import encode_cli as cli

    src = tmp_path / "lenna.npy"
    dst = tmp_path / "out.img"
    np.save(src, golden("lenna")["img"])
    assert cli.main([str(src), str(dst)]) == 0
    out = capsys.readouterr().out.splitlines()
    n = manifest["lenna_q50"]["bytes"]
    assert out[0] == f"{n} bytes" and out[1] == f"Compression Ratio: {512 * 512 / n}:1"
    assert dst.read_bytes() == golden("lenna")["q50_bs"].tobytes()


def test_input_dtypes_and_layouts(ctx, golden):
    """Any numeric dtype / F-order / strided input gives the same bytes; the input is not modified."""
    d = golden("lenna")
    img = d["img"][:64, :72]
    want = T.compress(img, 50, ctx=ctx)
    for variant in (img.astype(np.int16), img.astype(np.float32), np.asfortranarray(img), img.astype(np.int64)):
        assert T.compress(variant, 50, ctx=ctx) == want
    big = np.zeros((128, 144), np.uint8)
    big[::2, ::2] = img
    view = big[::2, ::2]
    keep = view.copy()
    assert T.compress(view, 50, ctx=ctx) == want
    assert np.array_equal(view, keep)


@pytest.mark.parametrize("h,w", [(512, 512), (1080, 1920)])
def test_seeded_frames_stream_digest(ctx, manifest, h, w):
    m = manifest[f"rand1234_{h}x{w}_q50"]
    img = rand_frame(1234, h, w)
    info = T.encode(img, 50, ctx=ctx)
    assert sha(info["dc"].astype("<i4").tobytes()) == m["dc_i4_sha256"]
    assert sha(info["ac"].astype("<i4").tobytes()) == m["ac_i4_sha256"]
    bs = T.compress(img, 50, ctx=ctx)
    assert len(bs) == m["bytes"] and sha(bs) == m["sha256"]


@pytest.mark.parametrize("q", [10, 50, 90])
def test_config2_4096_coefficient_digest(ctx, manifest, q):
    """BASELINE config 2: 4096x4096 seed 1234 - digests of the reference's own encode() output; both kernels."""
    m = manifest[f"rand1234_4096x4096_q{q}"]
    img = rand_frame(1234, 4096, 4096)
    f = DevFrame(ctx, img)
    N.load().tic_set_stats(ctx.handle, 1)
    f.fallbacks()
    zz_h = f.run(q, N.KERNEL_HYBRID)
    fb = f.fallbacks()
    N.load().tic_set_stats(ctx.handle, 0)
    zz_e = f.run(q, N.KERNEL_EXACT)
    f.free()
    assert np.array_equal(zz_h, zz_e)
    dc, ac = zz_to_dc_ac(zz_h)
    assert sha(dc.astype("<i4").tobytes()) == m["dc_i4_sha256"]
    assert sha(ac.astype("<i4").tobytes()) == m["ac_i4_sha256"]
    assert 0 < fb < 0.08 * 262144, fb  # guard band trips on a small fraction of blocks only


def _true_tie_block():
    """X(2,2) = (2(P+Q) + sqrt2 (P-Q+R)) / 16 with P-Q+R = 0 and (P+Q)/128 = 0.5: an exact .5 tie of an IRRATIONAL coefficient
    at q=50 (scipy gives 0.49999999999999994 -> 0); only the exact operation order settles it."""
    blk = np.full((8, 8), 128, np.int32)
    blk[0, 0] += 64
    blk[0, 1] -= 64
    return blk.astype(np.uint8)


def test_rare_paths_of_the_strip_kernel(ctx, oracle, golden, monkeypatch):
    """Every rare path of the production kernel against the exact kernel and the oracle: rational ties settled inside the loop
    (rational_quad: flat, banded, two-level and posterised content trips it in every strip, noise in one strip of six), the wave's
    batch of irrational trips (second level, exact order inside the batch, strips with ties AND trips), batch overflow and strips in
    which all eight blocks trip (whole strip redone in the exact order)."""
    tt = _true_tie_block()
    mix = rand_frame(5, 1024, 2048)
    for k in range(0, 128 * 256, 37):
        by, bx = divmod(k, 256)
        mix[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = tt
    half = rand_frame(11, 512, 1024)
    half[:, 512:] = 77  # right half flat and odd: eight DC ties per strip there, none in the random half
    rs = np.random.RandomState(77)
    levels = rs.permutation(np.arange(64 * 256, dtype=np.int64) % 256).reshape(64, 256).astype(np.uint8)  # one grey level per block
    flat_mix = np.repeat(np.repeat(levels, 8, 0), 8, 1)
    noise_at = rs.rand(64, 256) < 0.3
    flat_mix = np.where(np.repeat(np.repeat(noise_at, 8, 0), 8, 1), rand_frame(78, 512, 2048), flat_mix)
    yy, xx = np.mgrid[0:512, 0:2048]
    banded = ((xx // 3 + yy // 5) // 8 * 8 + 1).astype(np.uint8)  # 32 levels, all odd: steps cross blocks, most blocks are flat
    ab = rs.randint(0, 256, (64, 256, 2))
    checker = np.where((yy + xx) % 2 == 0, np.repeat(np.repeat(ab[..., 0], 8, 0), 8, 1), np.repeat(np.repeat(ab[..., 1], 8, 0), 8, 1)).astype(np.uint8)
    frames = {
        "flat odd grey (DC tie in every block)": (np.full((1024, 2048), 129, np.uint8), (50, 90)),
        "irrational true tie in every block": (np.tile(tt, (128, 256)), (50,)),
        "random + scattered irrational true ties": (mix, (50,)),
        "tie goldens tiled": (np.tile(golden("tie_blocks")["img"].astype(np.uint8), (8, 16)), (50, 37, 90)),
        "half random, half flat odd": (half, (50,)),
        # flat blocks of every grey level, every kind of divisor: DC ties at the odd levels
        "flat blocks of every grey level": (np.repeat(np.repeat(levels, 8, 0), 8, 1), (50, 1, 10, 25, 49, 51, 75, 90, 99)),
        "flat blocks of every grey level among noise": (flat_mix, (50, 75)),
        "posterised ramp (banded content)": (banded, (50, 30, 80)),
        # dense rational ties that are NOT flat (sum 32 (a + b) with a + b = 2 mod 4)
        "two-level checkerboards": (checker, (50, 90)),
        # many irrational trips per wave (narrow divisors): full batches, overflowing batches, ties in the same strips
        "noise at the top of the quality range": (rand_frame(12, 1024, 2048), (90, 97, 99)),
        "posterised noise (ties and trips in the same strips)": ((rand_frame(13, 1024, 2048) // 8 * 8 + 1).astype(np.uint8), (50, 90, 99)),
        # the same rare paths under the TEAM schedule (4096 x 4096: the whole grid resident at once, rows split between the rounds): whole strips
        # redone in the exact order find their pixels through the team schedule's strip index
        "irrational true tie in every block, 4096 x 4096 (team schedule)": (np.tile(tt, (512, 512)), (50,)),
        "posterised noise, 4096 x 4096 (team schedule)": ((rand_frame(14, 4096, 4096) // 8 * 8 + 1).astype(np.uint8), (50, 99)),
    }
    monkeypatch.setenv("TIC_TUNE", "1")  # re-read the knobs at every launch
    for name, (img, quals) in frames.items():
        f = DevFrame(ctx, img)
        for q in quals:
            want = oracle.encode_zz16(img, q)
            assert np.array_equal(f.run(q, N.KERNEL_EXACT), want), (name, q, "exact kernel")
            for order in ("1", "0"):  # both instantiations of the strip kernel: columns first, rows first (TIC_ORDER: csrc/tic_hooks.h)
                monkeypatch.setenv("TIC_ORDER", order)
                assert np.array_equal(f.run(q, N.KERNEL_HYBRID), want), (name, q, "production kernel", "columns first" if order == "1" else "rows first")
            monkeypatch.delenv("TIC_ORDER")
            assert np.array_equal(f.run(q, N.KERNEL_HYBRID), want), (name, q, "production kernel, order by grid")
        f.free()


def test_strip_schedules_are_equivalent(ctx, monkeypatch):
    """Team / strided / chunked / round-interleaved walks (tuning knobs of launch_dctq) produce the same coefficients."""
    monkeypatch.setenv("TIC_TUNE", "1")  # re-read the knobs at every launch
    frames = {"small grid": rand_frame(5, 1024, 1024), "large grid": rand_frame(6, 4096, 8192),
              "team schedule, 65 strips per row": rand_frame(7, 2560, 4160),
              "team schedule falls back (more than 16 rows in round 0)": rand_frame(8, 6144, 6144),
              "ragged": rand_frame(9, 1999, 4171)}
    for name, img in frames.items():
        f = DevFrame(ctx, img)
        ref = f.run(50, N.KERNEL_EXACT)
        for knobs in ({}, {"TIC_SPLIT": "0"}, {"TIC_SPLIT": "1,1,1,1,1"}, {"TIC_SCHED": "0"}, {"TIC_SCHED": "1", "TIC_CHUNK": "5"},
                      {"TIC_SCHED": "2"}, {"TIC_MAX_WGS": "512"}, {"TIC_ORDER": "0"}, {"TIC_ORDER": "1"}, {"TIC_ORDER": "0", "TIC_SCHED": "0"},
                      {"TIC_ORDER": "1", "TIC_SCHED": "1", "TIC_CHUNK": "3"}):
            for k in ("TIC_SPLIT", "TIC_SCHED", "TIC_CHUNK", "TIC_MAX_WGS", "TIC_ORDER"):
                monkeypatch.delenv(k, raising=False)
            for k, v in knobs.items():
                monkeypatch.setenv(k, v)
            assert np.array_equal(f.run(50, N.KERNEL_HYBRID), ref), (name, knobs)
        f.free()


@pytest.mark.parametrize("q", [10, 50, 90])
def test_config5_16384_coefficient_digest(ctx, manifest, q):
    """BASELINE config 5: 16384x16384 at q=10/50/90 - digest vs the reference; hybrid == exact kernel."""
    m = manifest[f"rand1234_16384x16384_q{q}"]
    img = rand_frame(1234, 16384, 16384)
    f = DevFrame(ctx, img)
    zz = f.run(q, N.KERNEL_HYBRID)
    del img
    dc, ac = zz_to_dc_ac(zz)
    assert sha(dc.astype("<i4").tobytes()) == m["dc_i4_sha256"]
    del dc
    assert sha(ac.astype("<i4").tobytes()) == m["ac_i4_sha256"]
    del ac
    h_hybrid = sha(zz.tobytes())
    del zz
    assert sha(f.run(q, N.KERNEL_EXACT).tobytes()) == h_hybrid
    f.free()


def test_decompress_goldens(ctx, golden, manifest):
    """decompress(): decoded pixels identical to the reference's (exact IDCT order, truncating cast)."""
    d = golden("decode_small")
    s = golden("transform_small")
    for key in d["names"]:
        got = T.decompress(s[key + "_bs"].tobytes(), ctx=ctx)
        assert np.array_equal(got, d[key]), key
    sw = golden("quality_sweep")
    for q in sw["qualities"]:
        bs = sw[f"q{int(q)}_bs"].tobytes()
        if bs:
            assert np.array_equal(T.decompress(bs, ctx=ctx), d[f"sweep_q{int(q)}"]), q
    L = golden("lenna")
    for q in (10, 50, 90):
        got = T.decompress(L[f"q{q}_bs"].tobytes(), ctx=ctx)
        assert np.array_equal(got, L[f"q{q}_dec"])


def test_decompress_long_stream_parallel_decoder(ctx, oracle, monkeypatch):
    """Streams of 16,384 blocks and more take the parallel host Huffman decoder (ranges measured speculatively, stitched on the true
    chain, decoded by 16 threads): same pixels as the serial decoder and as the oracle, also when the stream is damaged."""
    img = rand_frame(4321, 1536, 2048)
    bs = T.compress(img, 50, ctx=ctx)
    want = oracle.decompress(bs)
    assert np.array_equal(T.decompress(bs, ctx=ctx), want)
    monkeypatch.setenv("TIC_DECODE_SERIAL", "1")
    assert np.array_equal(T.decompress(bs, ctx=ctx), want)
    monkeypatch.delenv("TIC_DECODE_SERIAL")
    damaged = bytearray(bs)
    damaged[len(bs) // 3] ^= 0x10
    damaged = bytes(damaged[: len(bs) * 9 // 10])
    assert np.array_equal(T.decompress(damaged, ctx=ctx), oracle.decompress(damaged))


def test_decode_dict_roundtrip(ctx, golden):
    """decode(encode(x)) through the reference's dict convention == decompress(compress(x))."""
    img = golden("lenna")["img"][:120, :200]
    info = T.encode(img, 75, ctx=ctx)
    info["scaled_dct"] = False
    a = T.decode(info, ctx=ctx)
    b = T.decompress(T.compress(img, 75, ctx=ctx), ctx=ctx)
    assert np.array_equal(a, b) and a.shape == img.shape
    err = a.astype(np.int32) - img.astype(np.int32)
    assert np.sqrt((err**2).mean()) < 6.0


def test_batch_pipeline_matches_single_frame(ctx, manifest):
    """BASELINE config 3 shape (1080p, seeds 1234+i): the stream-overlapped batch == per-frame compress()."""
    frames = [rand_frame(1234 + i, 1080, 1920) for i in range(6)]
    for threads in (4, 0):  # host entropy workers / device entropy stage
        out = T.compress_batch(frames, 50, threads=threads, ctx=ctx)
        assert sha(out[0]) == manifest["rand1234_1080x1920_q50"]["sha256"]
        for i in (1, 5):
            assert out[i] == T.compress(frames[i], 50, ctx=ctx)
    many = [rand_frame(900 + i, 72, 136) for i in range(37)]  # more than two chunks, ragged size
    got = T.compress_batch(many, 30, ctx=ctx)
    for i in (0, 15, 16, 17, 36):
        assert got[i] == T.compress(many[i], 30, ctx=ctx)
    # transform-only batch: coefficient digests of the reference for frames 0..3
    L = N.load()
    n = 4
    zz = [np.empty((32400, 64), np.int16) for _ in range(n)]
    inp = (C.c_void_p * n)(*[f.ctypes.data for f in frames[:n]])
    outp = (C.c_void_p * n)(*[z.ctypes.data for z in zz])
    ctx.check(L.tic_dctq_batch(ctx.handle, inp, n, 1080, 1920, 1920, 50, outp))
    for i in range(n):
        dc, ac = zz_to_dc_ac(zz[i])
        m = manifest[f"rand{1234 + i}_1080x1920_q50_coeffs"]
        assert sha(dc.astype("<i4").tobytes()) == m["dc_i4_sha256"]
        assert sha(ac.astype("<i4").tobytes()) == m["ac_i4_sha256"]


def test_config3_and_config4_full_size_streams_against_the_reference_manifest(ctx):
    """BASELINE configs 3 and 4 at FULL size: every one of the 2,048 frames of config 4 (1920x1080, seed 1234 + i, q = 50; the
    first 256 are config 3) goes through tic_compress_batch in shards of 256, as a rank of config 4 would send them, and every
    stream's size and sha256 must be what the unmodified reference's compress() (codec.py:133-164) produced for that frame
    (tests/golden/manifest_r4.json, generator tests/golden/gen/make_goldens_r4.py)."""
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest_r4.json")) as f:
        m = json.load(f)
    frames_m = m["frames"]
    assert len(frames_m) >= 256 and frames_m[0]["seed"] == 1234 and all(f["source"] == "reference" for f in frames_m[:256])
    checked = 0
    for lo in range(0, len(frames_m), 256):
        part = frames_m[lo:lo + 256]
        frames = [rand_frame(f["seed"], 1080, 1920) for f in part]
        out = T.compress_batch(frames, 50, threads=0, ctx=ctx)
        for f, bs in zip(part, out):
            assert len(bs) == f["bytes"] and sha(bs) == f["sha256"], f["seed"]
            checked += 1
        if lo == 0:
            assert sha(np.asarray([len(b) for b in out], dtype="<i8").tobytes()) == m["sizes_sha256_first256"]
    assert checked == len(frames_m)


def test_batched_launch_matches_per_frame(ctx, oracle):
    """tic_dctq_dev_frames: several frames (ragged size -> remainder strips too) in one launch == oracle per frame."""
    L = N.load()
    n, h, w = 5, 100, 200
    pitch = 256
    frames = [rand_frame(500 + k, h, w) for k in range(n)]
    host = np.zeros((n, h + 3, pitch), np.uint8)  # frame stride larger than one frame
    for k, f in enumerate(frames):
        host[k, :h, :w] = f
    nblk = L.tic_num_blocks(h, w)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * nblk * 128 + 256 * n, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, host.ctypes.data, host.size))
    cstride = nblk * 128 + 256
    for variant in (N.KERNEL_HYBRID, N.KERNEL_EXACT):
        ctx.check(L.tic_memset_dev(ctx.handle, d_out, 0x33, n * cstride))
        ctx.check(L.tic_dctq_dev_frames(ctx.handle, d_img, n, h, w, pitch, (h + 3) * pitch, 75, d_out, cstride, variant))
        raw = np.empty(n * cstride, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, raw.ctypes.data, d_out, raw.size))
        for k in range(n):
            got = raw[k * cstride : k * cstride + nblk * 128].view(np.int16).reshape(nblk, 64)
            assert np.array_equal(got, oracle.encode_zz16(frames[k], 75)), (k, variant)
            assert (raw[k * cstride + nblk * 128 : (k + 1) * cstride] == 0x33).all()  # gap untouched
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)
    # frames that follow each other without a gap and end on a block row are walked as ONE tall frame (ragged width: the
    # remainder strips of every frame come from the exact kernel's tall launch too)
    n, h, w, pitch = 7, 96, 200, 208
    frames = [rand_frame(600 + k, h, w) for k in range(n)]
    host = np.zeros((n, h, pitch), np.uint8)
    for k, f in enumerate(frames):
        host[k, :, :w] = f
    nblk = L.tic_num_blocks(h, w)
    ctx.check(L.tic_dev_alloc(ctx.handle, host.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, n * nblk * 128, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, host.ctypes.data, host.size))
    for variant in (N.KERNEL_HYBRID, N.KERNEL_EXACT):
        ctx.check(L.tic_memset_dev(ctx.handle, d_out, 0x33, n * nblk * 128))
        ctx.check(L.tic_dctq_dev_frames(ctx.handle, d_img, n, h, w, pitch, h * pitch, 60, d_out, nblk * 128, variant))
        raw = np.empty(n * nblk * 128, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, raw.ctypes.data, d_out, raw.size))
        for k in range(n):
            got = raw[k * nblk * 128 : (k + 1) * nblk * 128].view(np.int16).reshape(nblk, 64)
            assert np.array_equal(got, oracle.encode_zz16(frames[k], 60)), (k, variant)
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)


def test_device_entropy_stage_matches_reference_streams(ctx, oracle, golden, manifest):
    """tic_entropy_encode_dev (bit counting + scan + parallel packing on the GPU) == reference compress() bytes,
    fed with oracle coefficients so that the stage is checked on its own."""
    L = N.load()

    def dev_entropy(zz, h, w, q):
        zz = np.ascontiguousarray(zz, dtype=np.int16)
        cap = L.tic_compress_bound(h, w)
        d_zz, d_out = C.c_void_p(), C.c_void_p()
        ctx.check(L.tic_dev_alloc(ctx.handle, max(zz.nbytes, 16), C.byref(d_zz)))
        ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
        if zz.nbytes:
            ctx.check(L.tic_memcpy_h2d(ctx.handle, d_zz, zz.ctypes.data, zz.nbytes))
        n = C.c_size_t()
        rc = L.tic_entropy_encode_dev(ctx.handle, d_zz, h, w, q, d_out, cap, C.byref(n))
        out = np.empty(max(n.value, 1), np.uint8)
        if rc == 0:
            ctx.check(L.tic_memcpy_d2h(ctx.handle, out.ctypes.data, d_out, n.value))
        L.tic_dev_free(ctx.handle, d_zz)
        L.tic_dev_free(ctx.handle, d_out)
        return rc, out[: n.value].tobytes()

    d = golden("lenna")
    for q in (10, 50, 90):
        rc, bs = dev_entropy(oracle.encode_zz16(d["img"], q), 512, 512, q)
        assert rc == 0 and bs == d[f"q{q}_bs"].tobytes()
    s = golden("transform_small")
    for key in s["names"]:
        img = s[key + "_img"]
        q = int(str(key).rsplit("_q", 1)[1])
        want = s[key + "_bs"].tobytes()
        rc, bs = dev_entropy(oracle.encode_zz16(img, q), img.shape[0], img.shape[1], q)
        if want:
            assert rc == 0 and bs == want, key
        else:
            assert rc == N.TIC_E_RANGE, key
    # long zero runs (ZRL), values at the size-category edges, last coefficient non-zero, all-zero blocks
    rng = np.random.default_rng(9)
    zz = np.zeros((64, 64), np.int16)
    for b in range(64):
        for _ in range(int(rng.integers(0, 6))):
            zz[b, int(rng.integers(0, 64))] = int(rng.choice([1, -1, 2, -3, 7, -8, 255, -256, 1023, -1023]))
    zz[3, 63] = -5
    zz[:, 0] = rng.integers(-1000, 1000, 64)
    rc, bs = dev_entropy(zz, 64, 64, 50)
    assert rc == 0 and bs == T.entropy_encode(zz, 64, 64, 50)
    img = rand_frame(1234, 1080, 1920)
    rc, bs = dev_entropy(oracle.encode_zz16(img, 50), 1080, 1920, 50)
    assert rc == 0 and sha(bs) == manifest["rand1234_1080x1920_q50"]["sha256"]


def test_compress_dev_resident(ctx, manifest):
    """tic_compress_dev: image and stream both resident in HBM (512^2 seed 1234 stream digest of the reference)."""
    L = N.load()
    img = rand_frame(1234, 512, 512)
    cap = L.tic_compress_bound(512, 512)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    n = C.c_size_t()
    ctx.check(L.tic_compress_dev(ctx.handle, d_img, 512, 512, 512, 50, d_out, cap, C.byref(n)))
    out = np.empty(n.value, np.uint8)
    ctx.check(L.tic_memcpy_d2h(ctx.handle, out.ctypes.data, d_out, n.value))
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)
    m = manifest["rand1234_512x512_q50"]
    assert n.value == m["bytes"] and sha(out.tobytes()) == m["sha256"]


def test_compress_dev_async_matches_the_synchronous_call(ctx, manifest):
    """tic_compress_dev_async / tic_async_result (round 5): frames queued back to back on the context's stream give the streams of
    tic_compress_dev (= the reference's, tests above), in call order, with per-frame status: a frame whose stream does not fit
    reports TIC_E_SPACE and the frames around it are unaffected; at most 64 tickets are open; an empty image is a header."""
    L = N.load()
    frames = [rand_frame(900 + k, 256 + 8 * (k % 3), 320) for k in range(6)]
    want = [T.compress(f, 40 + 10 * k, ctx=ctx) for k, f in enumerate(frames)]
    cap = L.tic_compress_bound(272, 320)
    d_imgs, d_outs = [], []
    try:
        for f in frames:
            a, b = C.c_void_p(), C.c_void_p()
            ctx.check(L.tic_dev_alloc(ctx.handle, f.size, C.byref(a)))
            ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(b)))
            ctx.check(L.tic_memcpy_h2d(ctx.handle, a, np.ascontiguousarray(f).ctypes.data, f.size))
            d_imgs.append(a)
            d_outs.append(b)
        tickets = []
        for k, f in enumerate(frames):
            t = C.c_longlong(-1)
            small = 64 if k == 3 else cap  # frame 3: a buffer that cannot hold its stream
            ctx.check(L.tic_compress_dev_async(ctx.handle, d_imgs[k], f.shape[0], f.shape[1], f.shape[1], 40 + 10 * k, d_outs[k], small, C.byref(t)))
            tickets.append(t.value)
        assert tickets == list(range(tickets[0], tickets[0] + 6))
        n = C.c_size_t()
        rc = L.tic_async_result(ctx.handle, tickets[-1], 0, C.byref(n))  # (poll: either still busy or done)
        assert rc in (N.TIC_E_BUSY, N.TIC_OK)
        if rc == N.TIC_OK:
            assert n.value == len(want[-1])
            tickets_left = tickets[:-1]
        else:
            tickets_left = tickets
        for k, t in enumerate(tickets_left):
            rc = L.tic_async_result(ctx.handle, t, 1, C.byref(n))
            if k == 3:
                assert rc == N.TIC_E_SPACE
                continue
            assert rc == N.TIC_OK and n.value == len(want[k]), (k, rc)
            got = np.empty(n.value, np.uint8)
            ctx.check(L.tic_memcpy_d2h(ctx.handle, got.ctypes.data, d_outs[k], n.value))
            assert got.tobytes() == want[k], k
        assert L.tic_async_result(ctx.handle, tickets[0], 1, C.byref(n)) == N.TIC_E_ARG  # closed
        # 64 open tickets at most; results may be collected in any order
        ts = []
        for k in range(64):
            t = C.c_longlong()
            ctx.check(L.tic_compress_dev_async(ctx.handle, d_imgs[0], frames[0].shape[0], 320, 320, 40, d_outs[0], cap, C.byref(t)))
            ts.append(t.value)
        t = C.c_longlong()
        assert L.tic_compress_dev_async(ctx.handle, d_imgs[0], frames[0].shape[0], 320, 320, 40, d_outs[0], cap, C.byref(t)) == N.TIC_E_ARG
        for t in reversed(ts):
            assert L.tic_async_result(ctx.handle, t, 1, C.byref(n)) == N.TIC_OK and n.value == len(want[0])
        # an empty image, a bad quality, a misaligned buffer
        t = C.c_longlong()
        ctx.check(L.tic_compress_dev_async(ctx.handle, None, 0, 8, 8, 50, d_outs[1], cap, C.byref(t)))
        assert L.tic_async_result(ctx.handle, t.value, 1, C.byref(n)) == N.TIC_OK and n.value == 16
        hdr = np.empty(16, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, hdr.ctypes.data, d_outs[1], 16))
        assert hdr.tobytes() == T.compress(np.zeros((0, 8), np.uint8), 50, ctx=ctx)
        assert L.tic_compress_dev_async(ctx.handle, d_imgs[0], 256, 320, 320, 100, d_outs[0], cap, C.byref(t)) == N.TIC_E_QUALITY
        assert L.tic_compress_dev_async(ctx.handle, d_imgs[0], 256, 320, 320, 50, C.c_void_p(d_outs[0].value + 4), cap - 4, C.byref(t)) == N.TIC_E_ARG
        ctx.check(L.tic_sync(ctx.handle))
    finally:
        for d in d_imgs + d_outs:
            L.tic_dev_free(ctx.handle, d)


def test_compress_dev_async_two_lanes_in_one_context(ctx, oracle):
    """Round 6: the tickets of ONE context alternate between two lanes (transform and packing on the lane's stream, placing on the
    context's): 40 different frames at alternating qualities in one burst give the synchronous call's streams, ticket by ticket; a frame
    with a coefficient that has no Huffman code reports TIC_E_RANGE through its own ticket and leaves its neighbours - one on each lane -
    alone; a synchronous call between two bursts and a decode queued behind a ticket see the tickets' results."""
    L = N.load()
    h, w = 1024, 1024
    frames = [rand_frame(7000 + k, h, w) if k % 5 else (rand_frame(7000 + k, h, w) // 32 * 32).astype(np.uint8) for k in range(40)]
    quals = [(30, 50, 85, 12)[k % 4] for k in range(40)]
    loud = np.where((np.indices((h, w))[1] % 8) < 4, 0, 255).astype(np.uint8)  # every block half black, half white: |AC| >= 1024 at q = 99
    with pytest.raises(KeyError):
        T.compress(loud, 99, ctx=ctx)
    frames[17], quals[17] = loud, 99
    cap = L.tic_compress_bound(h, w)
    want = [None if k == 17 else oracle.compress(frames[k], quals[k]) for k in range(40)]
    d_imgs, d_outs = [], []
    try:
        for f in frames:
            a, b = C.c_void_p(), C.c_void_p()
            ctx.check(L.tic_dev_alloc(ctx.handle, f.size, C.byref(a)))
            ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(b)))
            ctx.check(L.tic_memcpy_h2d(ctx.handle, a, np.ascontiguousarray(f).ctypes.data, f.size))
            d_imgs.append(a)
            d_outs.append(b)
        n = C.c_size_t()
        for burst in range(2):
            for b in d_outs:
                ctx.check(L.tic_memset_dev(ctx.handle, b, 0x11, cap))
            ts = []
            for k in range(40):
                t = C.c_longlong()
                ctx.check(L.tic_compress_dev_async(ctx.handle, d_imgs[k], h, w, w, quals[k], d_outs[k], cap, C.byref(t)))
                ts.append(t.value)
            order = range(40) if burst == 0 else reversed(range(40))  # results in any order
            for k in order:
                rc = L.tic_async_result(ctx.handle, ts[k], 1, C.byref(n))
                if k == 17:
                    assert rc == N.TIC_E_RANGE, rc
                    continue
                assert rc == N.TIC_OK and n.value == len(want[k]), (burst, k, rc, n.value)
                got = np.empty(n.value, np.uint8)
                ctx.check(L.tic_memcpy_d2h(ctx.handle, got.ctypes.data, d_outs[k], n.value))
                assert got.tobytes() == want[k], (burst, k)
            # a synchronous call between the bursts (the context's own scratch, the context's stream)
            ctx.check(L.tic_compress_dev(ctx.handle, d_imgs[3], h, w, w, quals[3], d_outs[3], cap, C.byref(n)))
            assert n.value == len(want[3])
        # a decode queued right behind a ticket, before the ticket is collected: it sees the ticket's stream (the placing kernels run on the
        # context's stream, the decoder starts behind everything queued there)
        d_pix = C.c_void_p()
        ctx.check(L.tic_dev_alloc(ctx.handle, h * w, C.byref(d_pix)))
        d_imgs.append(d_pix)
        for _ in range(3):  # (two equal headers in a row before the decoder launches on a guess)
            ctx.check(L.tic_decompress_dev(ctx.handle, d_outs[3], len(want[3]), d_pix, w, h * w, None, None))
        ctx.check(L.tic_memset_dev(ctx.handle, d_outs[3], 0, cap))
        t, td = C.c_longlong(), C.c_longlong()
        ctx.check(L.tic_compress_dev_async(ctx.handle, d_imgs[3], h, w, w, quals[3], d_outs[3], cap, C.byref(t)))
        ctx.check(L.tic_decompress_dev_async(ctx.handle, d_outs[3], len(want[3]), d_pix, w, h * w, C.byref(td)))
        ctx.check(L.tic_decompress_async_result(ctx.handle, td.value, 1, None, None))
        ctx.check(L.tic_async_result(ctx.handle, t.value, 1, C.byref(n)))
        pix = np.empty((h, w), np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, pix.ctypes.data, d_pix, pix.size))
        assert np.array_equal(pix, oracle.decompress(want[3]))
    finally:
        for d in d_imgs + d_outs:
            L.tic_dev_free(ctx.handle, d)


def test_compress_batch_over_several_contexts(ctx, manifest):
    """tic_compress_batch_multi / compress_batch(devices=[...]) (round 5): one process, a context and a host thread per listed device,
    contiguous shards, sizes in frame order - on a one-GPU box two contexts on device 0.  Streams equal the single-context batch
    (= the reference's: manifest_r4.json for the 1080p seeds)."""
    import json

    L = N.load()
    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest_r4.json")) as f:
        by_seed = {e["seed"]: e for e in json.load(f)["frames"]}
    frames = [rand_frame(1234 + k, 1080, 1920) for k in range(21)]  # 21 frames over 2 contexts: shards of 11 and 10
    for devices, threads in (([0, 0], 0), ([0, 0, 0], 0), ([0, 0], 4), ([0], 0)):
        streams = T.compress_batch(frames, 50, threads=threads, devices=devices)
        assert len(streams) == 21
        for k, s in enumerate(streams):
            e = by_seed[1234 + k]
            assert len(s) == e["bytes"] and sha(s) == e["sha256"], (devices, threads, k)
    assert T.compress_batch(frames[:2], 50, devices=[0, 0, 0]) == [T.compress(f, 50, ctx=ctx) for f in frames[:2]]  # more contexts than frames
    assert T.compress_batch([], 50, devices=[0, 0]) == []
    with pytest.raises(ValueError):
        T.compress_batch(frames[:2], 50, devices=[])
    with pytest.raises(KeyError):  # a shard's error is the call's error
        T.compress_batch([np.full((64, 64), 255, np.uint8) * (k % 2) for k in range(4)] + [(np.indices((64, 64)).sum(0) % 2 * 255).astype(np.uint8)], 99, devices=[0, 0])
    # C-ABI argument checks: the same context twice, null arrays
    c2 = T.Context(0)
    try:
        hs = (C.c_void_p * 2)(ctx.handle, ctx.handle)
        failed = C.c_int(7)
        assert L.tic_compress_batch_multi(hs, 2, None, 1, 8, 8, 8, 50, None, None, None, 0, C.byref(failed)) == N.TIC_E_ARG
        hs = (C.c_void_p * 2)(ctx.handle, c2.handle)
        assert L.tic_compress_batch_multi(hs, 2, None, 1, 8, 8, 8, 50, None, None, None, 0, C.byref(failed)) == N.TIC_E_ARG
        assert L.tic_compress_batch_multi(hs, 2, None, 0, 8, 8, 8, 50, None, None, None, 0, C.byref(failed)) == N.TIC_OK and failed.value == -1
        assert L.tic_compress_batch_multi(hs, 0, None, 0, 8, 8, 8, 50, None, None, None, 0, None) == N.TIC_E_ARG
        # placement helpers of the multi-rank bench line
        assert L.tic_get_stage_threads(ctx.handle) in range(1, 9)
        ctx.check(L.tic_set_stage_threads(ctx.handle, 3))
        assert L.tic_get_stage_threads(ctx.handle) == 3
        ctx.check(L.tic_set_stage_threads(ctx.handle, 0))
        assert L.tic_set_stage_threads(ctx.handle, -1) == N.TIC_E_ARG
        pci = L.tic_pci_bus_id(ctx.handle).decode()
        assert len(pci.split(":")) == 3, pci
    finally:
        c2.close()


def test_device_entropy_content_sweep(ctx):
    """Device entropy stage against the host coder on content that stresses the placing kernel: flat frames (48-bit partitions:
    a stream word holds the ends of two partitions), gradients, sparse spikes (long zero runs: ZRL decided per wave), noise at
    the quality extremes, and shapes whose partition count is not a multiple of the kernels' group sizes."""
    rng = np.random.default_rng(2024)
    shapes = [(1024, 1536), (8, 8), (8, 72), (136, 200), (1000, 1000), (264, 2056)]
    for h, w in shapes:
        spikes = np.full((h, w), 128, np.uint8)
        spikes[rng.integers(0, h, max(1, h * w // 4000)), rng.integers(0, w, max(1, h * w // 4000))] = 255
        contents = {
            "flat": np.full((h, w), 201, np.uint8),
            "gradient": np.tile((np.arange(w) // 3).astype(np.uint8), (h, 1)),
            "spikes": spikes,
            "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
            "soft noise": rng.integers(120, 136, (h, w), dtype=np.uint8),
        }
        for name, img in contents.items():
            for q in (1, 50, 97):
                zz = T.dctq(img, q, ctx=ctx)
                try:
                    want = T.entropy_encode(zz, h, w, q)
                except KeyError:
                    with pytest.raises(KeyError):
                        T.compress(img, q, ctx=ctx)
                    continue
                assert T.compress(img, q, ctx=ctx) == want, (h, w, name, q)


def test_device_entropy_two_level_offsets(ctx, monkeypatch):
    """Frames of very many partitions take their stream offsets through tile sums (a third small launch); the switch is moved
    down here so that a 2048x3000 frame (750 groups, 3 tiles) walks that path.  Quality 90 makes the partitions long
    enough that the placing kernel reads the staging slots directly instead of through LDS.  Checked against the host
    entropy coder."""
    L = N.load()
    img = rand_frame(77, 2048, 3000)
    cap = L.tic_compress_bound(2048, 3000)
    d_img, d_out = C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_out)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
    for q in (50, 90):
        want = T.entropy_encode(T.dctq(img, q), 2048, 3000, q)
        for direct in ("3", None):
            if direct is None:
                monkeypatch.delenv("TIC_ENT_DIRECT_GROUPS", raising=False)
            else:
                monkeypatch.setenv("TIC_ENT_DIRECT_GROUPS", direct)
            n = C.c_size_t()
            ctx.check(L.tic_compress_dev(ctx.handle, d_img, 2048, 3000, 3000, q, d_out, cap, C.byref(n)))
            out = np.empty(n.value, np.uint8)
            ctx.check(L.tic_memcpy_d2h(ctx.handle, out.ctypes.data, d_out, n.value))
            assert out.tobytes() == want, (q, direct)
    L.tic_dev_free(ctx.handle, d_img)
    L.tic_dev_free(ctx.handle, d_out)


def test_error_paths(ctx):
    L = N.load()
    img = rand_frame(1, 16, 16)
    zz = np.zeros((4, 64), np.int16)
    assert L.tic_dctq(ctx.handle, img.ctypes.data, 16, 16, 16, 0, zz.ctypes.data) == N.TIC_E_QUALITY
    assert L.tic_dctq(ctx.handle, img.ctypes.data, 16, 16, 16, 100, zz.ctypes.data) == N.TIC_E_QUALITY
    assert L.tic_dctq(ctx.handle, img.ctypes.data, 16, 16, 8, 50, zz.ctypes.data) == N.TIC_E_ARG
    assert b"stride" in L.tic_last_error(ctx.handle)
    assert T.compress(np.zeros((0, 8), np.uint8), 50, ctx=ctx).hex() == "00000000080000003200000000000000"
    # device entropy stage: a stream buffer that is too small is reported, never overrun (guard bytes stay intact)
    img = rand_frame(2, 256, 256)
    f = DevFrame(ctx, img)
    cap, guard = 4096, 4096
    d_s = C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, cap + guard, C.byref(d_s)))
    ctx.check(L.tic_memset_dev(ctx.handle, d_s, 0xA5, cap + guard))
    n = C.c_size_t()
    assert L.tic_compress_dev(ctx.handle, f.d_img, 256, 256, f.pitch, 50, d_s, cap, C.byref(n)) == N.TIC_E_SPACE
    tail = np.empty(guard, np.uint8)
    ctx.check(L.tic_memcpy_d2h(ctx.handle, tail.ctypes.data, C.c_void_p(d_s.value + cap), guard))
    assert (tail == 0xA5).all()
    L.tic_dev_free(ctx.handle, d_s)
    f.free()


def test_batch_pipeline_error_paths(ctx):
    """The three-thread batch pipeline reports a frame whose output buffer is too small (no overrun of that buffer, no hang), keeps
    working afterwards, and handles batches that are not a multiple of the chunk (40 frames: chunks of 16, 16, 8)."""
    L = N.load()
    n, h, w = 40, 128, 192
    frames = [rand_frame(100 + i, h, w) for i in range(n)]
    want = [T.compress(f, 50, ctx=ctx) for f in frames[:3]] + [None] * (n - 3)
    cap = L.tic_compress_bound(h, w)
    outs = [np.full(cap + 64, 0xA5, np.uint8) for _ in range(n)]
    caps_list = [cap] * n
    caps_list[21] = 100  # chunk 1, frame 5 of it
    imgs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * n)(*caps_list)
    lens = (C.c_size_t * n)()
    rc = L.tic_compress_batch(ctx.handle, imgs, n, h, w, w, 50, outp, caps, lens, 0)
    assert rc == N.TIC_E_SPACE and b"frame 21" in L.tic_last_error(ctx.handle)
    assert (outs[21][100:] == 0xA5).all()  # nothing written past the small buffer
    caps[21] = cap
    ctx.check(L.tic_compress_batch(ctx.handle, imgs, n, h, w, w, 50, outp, caps, lens, 0))
    for i in range(3):
        assert outs[i][: lens[i]].tobytes() == want[i]
    for i in range(n):
        assert (outs[i][cap:] == 0xA5).all() and 16 < lens[i] <= cap
    assert outs[39][: lens[39]].tobytes() == T.compress(frames[39], 50, ctx=ctx)


def test_coefficient_without_huffman_code_raises_keyerror(ctx, golden):
    """|AC| >= 1024 has no Huffman code: the reference raises KeyError (goldens with an empty stream)."""
    d = golden("transform_small")
    hit = 0
    for key in d["names"]:
        if d[key + "_bs"].size == 0:
            q = int(str(key).rsplit("_q", 1)[1])
            with pytest.raises(KeyError):
                T.compress(d[key + "_img"], q, ctx=ctx)
            hit += 1
    assert hit >= 1


def _config4_worker(rank, world, port, n_frames, out_dir):
    """One rank of the BASELINE config 4 rehearsal: its shard of 1080p frames through the real pipeline on device 0."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tinyimgcodec_amd as TT
    from tinyimgcodec_amd.distributed import TorchComm, compress_sharded

    c = TT.Context(0)  # both ranks share the one GPU of the box (RCCL wants a GPU per rank: the sizes travel over gloo here)
    comm = TorchComm()
    lo, hi, streams, sizes, offsets = compress_sharded(lambda i: rand_frame(1234 + i, 1080, 1920), n_frames, 50, comm=comm,
                                                       compress_batch_fn=lambda fr, q: TT.compress_batch(fr, q, threads=0, ctx=c))
    enc = [TT.encode(rand_frame(1234 + i, 1080, 1920), 50, ctx=c) for i in range(lo, hi)]
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, sizes=sizes, offsets=offsets,
             stream_sha=np.array([sha(s) for s in streams]), mine=np.array([len(s) for s in streams], dtype=np.int64),
             dc_sha=np.array([sha(e["dc"].astype("<i4").tobytes()) for e in enc]),
             ac_sha=np.array([sha(e["ac"].astype("<i4").tobytes()) for e in enc]))
    dist.barrier()
    dist.destroy_process_group()
    c.close()


def test_config4_rehearsal_two_ranks_on_one_gpu(tmp_path, manifest):
    """BASELINE config 4 on the hardware that exists: 2 ranks (gloo) share device 0, each sends its contiguous shard of
    1920x1080 frames (seed 1234 + i) through the real compress_batch; the gathered sizes and offsets agree on both ranks,
    frame 0's stream and the coefficients of frames 0-3 carry the reference's digests."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, n_frames = 2, 8
    mp.spawn(_config4_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    d = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    assert [(int(x["lo"]), int(x["hi"])) for x in d] == [(0, 4), (4, 8)]
    sizes = np.concatenate([x["mine"] for x in d])
    for x in d:  # every rank knows every frame's size and offset
        assert np.array_equal(x["sizes"], sizes)
        assert np.array_equal(x["offsets"], np.concatenate([[0], np.cumsum(sizes)]))
    m0 = manifest["rand1234_1080x1920_q50"]
    assert int(sizes[0]) == m0["bytes"] and str(d[0]["stream_sha"][0]) == m0["sha256"]
    for i in range(4):
        m = manifest["rand%d_1080x1920_q50_coeffs" % (1234 + i)]
        assert str(d[0]["dc_sha"][i]) == m["dc_i4_sha256"] and str(d[0]["ac_sha"][i]) == m["ac_i4_sha256"], i


@pytest.mark.parametrize("variant", [N.KERNEL_EXACT, N.KERNEL_HYBRID])
def test_near_ties_round2(ctx, golden, variant):
    """Reference-generated fixture: exact ties of the irrational coefficients (2,2) (2,6) (6,2) (6,6), of the rational four, and
    random blocks with an irrational coefficient within 1e-6 of a tie - decided by the reference's float64, never by a band."""
    d = golden("near_ties")
    f = DevFrame(ctx, d["img"])
    for q in (50, 90, 10, 37):
        dc, ac = zz_to_dc_ac(f.run(q, variant))
        assert np.array_equal(dc, d[f"q{q}_dc"]), q
        assert np.array_equal(ac, d[f"q{q}_ac"]), q
    f.free()


def test_wide_pixels_round2(ctx, golden):
    """encode() on integer images outside 0..255 (codec.py:29 transforms any integers): the exact float64 device path equals the
    reference; compress() of such content raises KeyError where the reference's Huffman table has no code."""
    d = golden("wide_pixels")
    for name in d["names"]:
        key, q = str(name).rsplit("_q", 1)
        e = T.encode(d[key + "_img"], int(q), ctx=ctx)
        assert np.array_equal(e["dc"], d[f"{name}_dc"]), name
        assert np.array_equal(e["ac"], d[f"{name}_ac"]), name
    with pytest.raises(KeyError):
        T.compress(d["int16_full_img"], 50, ctx=ctx)
    small = np.array(d["int16_mixed_img"]) // 8  # |coefficients| stay codable
    e = T.encode(small, 50, ctx=ctx)
    bs = T.compress(small, 50, ctx=ctx)
    assert bs == T.entropy_encode(np.concatenate([np.cumsum(e["dc"])[:, None], e["ac"]], axis=1).astype(np.int16), *small.shape, 50)
    # float qualities: encode() accepts integral floats as the reference does, compress() fails in the header pack
    img = rand_frame(3, 16, 16)
    assert np.array_equal(T.encode(img, 50.0, ctx=ctx)["ac"], T.encode(img, 50, ctx=ctx)["ac"])
    import struct
    with pytest.raises(struct.error):
        T.compress(img, 50.0, ctx=ctx)
    with pytest.raises(ValueError):
        T.encode(img, 0.5, ctx=ctx)  # (non-integral qualities in [1, 99] work since round 4: test_non_integral_qualities_round4)


def test_non_integral_qualities_round4(ctx, golden, oracle):
    """encode() / decode() / dctq() with a quality that is not an integer - the reference computes with whatever number it is given
    (utils.py:50-53) - against the reference's own outputs (float_quality.npz: eight qualities x four images, incl. ragged shapes and
    flat blocks): coefficients and decoded pixels, both kernels; compress() raises struct.error for a float as the reference does; an
    integer call right after a float one uses the integer's constants again."""
    import struct

    L = N.load()
    d = golden("float_quality")
    for name in d["names"]:
        img = d["img_" + str(name)]
        for k, q in enumerate(d["qualities"]):
            q = float(q)
            e = T.encode(img, q, ctx=ctx)
            assert e["quality"] == q
            assert np.array_equal(e["dc"], d["dc_%s_%d" % (name, k)]) and np.array_equal(e["ac"], d["ac_%s_%d" % (name, k)]), (name, q)
            e["scaled_dct"] = False
            assert np.array_equal(T.decode(e, ctx=ctx), d["px_%s_%d" % (name, k)]), (name, q)
            zz = T.dctq(img, q, ctx=ctx)
            dc, ac = zz_to_dc_ac(zz)
            assert np.array_equal(dc, d["dc_%s_%d" % (name, k)]) and np.array_equal(ac, d["ac_%s_%d" % (name, k)]), (name, q, "dctq")
    # both kernels on a resident frame with the custom slot (TIC_QUALITY_CUSTOM), a larger random frame against the oracle
    img = rand_frame(4242, 264, 328)
    for q in (33.3, 71.9):
        ctx.check(L.tic_set_custom_quality(ctx.handle, q))
        f = DevFrame(ctx, img)
        want_dc, want_ac = oracle.encode(img, q)
        for variant in (N.KERNEL_EXACT, N.KERNEL_HYBRID):
            dc, ac = zz_to_dc_ac(f.run(N.QUALITY_CUSTOM, variant))
            assert np.array_equal(dc, want_dc) and np.array_equal(ac, want_ac), (q, variant)
        f.free()
    # the integer qualities are untouched by the custom slot
    assert np.array_equal(T.dctq(img, 50, ctx=ctx), oracle.encode_zz16(img, 50))
    with pytest.raises(struct.error):
        T.compress(img, 37.5, ctx=ctx)
    for bad in (0.5, 99.5, -3.5, float("nan")):
        with pytest.raises((ValueError, ZeroDivisionError)):
            T.encode(img, bad, ctx=ctx)
    lens = C.c_size_t()
    out = np.empty(L.tic_compress_bound(264, 328), np.uint8)
    assert L.tic_compress(ctx.handle, img.ctypes.data, 264, 328, 328, N.QUALITY_CUSTOM, out.ctypes.data, out.size, C.byref(lens)) == N.TIC_E_QUALITY
    assert L.tic_set_custom_quality(ctx.handle, 100.0) == N.TIC_E_QUALITY


def test_rccl_single_rank_smoke(ctx, monkeypatch, tmp_path):  # (+ round 5: the library's version is read and checked before any call)
    """The RCCL leg of the C-ABI (tic_comm.hip) with ONE rank forced through the library: dlopen of librccl, unique id through
    the rendezvous file, ncclCommInitRank on the context's device, all-gather and all-reduce on its stream.  (Two ranks need
    two GPUs: that run is the driver's; the two-process flow is rehearsed with gloo in test_config4_rehearsal_two_ranks_on_one_gpu.)"""
    from tinyimgcodec_amd.distributed import RcclComm, gather_sizes

    monkeypatch.setenv("TIC_COMM_FORCE_RCCL", "1")
    comm = RcclComm(ctx, rank=0, world=1, rendezvous_path=str(tmp_path / "rdv"))
    try:
        mine = np.arange(1, 257, dtype=np.uint64) * np.uint64(1000003)
        got = comm.all_gather_u64(mine)
        assert got.shape == (1, 256) and np.array_equal(got[0], mine)
        assert 20000 <= comm.version() < 30000, comm.version()  # NCCL_VERSION_CODE of an RCCL 2.x (any other major is refused at creation)
        v = comm.allreduce_max([3.5, -2.0, 1e300])
        assert list(v) == [3.5, -2.0, 1e300]
        sizes, offsets = gather_sizes(list(range(10, 20)), 10, comm)
        assert list(sizes) == list(range(10, 20)) and offsets[-1] == sum(range(10, 20))
        assert not (tmp_path / "rdv").exists()  # rank 0 removed the rendezvous file after the first collective
    finally:
        comm.close()


def test_scaled_dct_streams_round2(ctx, golden):
    """decompress() of streams of the reference's C encoder (header flag 1<<30 -> decode()'s scaled_dct branch, codec.py:59-62)
    and of re-flagged Python streams: pixel-identical to the reference; decode() with scaled_dct=True takes the same path."""
    import hashlib
    import struct

    d = golden("scaled_streams")

    def same(name, img):
        if name + "_img" in d:
            return np.array_equal(img, d[name + "_img"])
        return (np.array_equal(img[:32], d[name + "_rows"])
                and hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == str(d[name + "_sha"]))

    for name in d["names"]:
        name = str(name)
        assert same(name, T.decompress(d[name + "_bs"].tobytes())), name
    # the dict interface: coefficients of a Python stream, handed to decode() as scaled ones
    img = rand_frame(5, 37, 53)
    info = T.encode(img, 75)
    info["scaled_dct"] = True
    info["quality"] = 2
    bs = bytearray(T.compress(img, 75))
    bs[:16] = struct.pack("IIII", 37, 53, 2, 1 << 30)
    assert np.array_equal(T.decode(info), T.decompress(bytes(bs)))
    info["quality"] = 63
    with pytest.raises(ValueError):
        T.decode(info)


def test_truncated_streams_round2(ctx, golden):
    """decompress() on truncated / corrupted streams returns what the reference returns (it swallows a block's exception,
    codec.py:178-186), including blocks that run to more than a thousand symbols without an end-of-block."""
    d = golden("truncated_streams")
    for name in d["names"]:
        name = str(name)
        got = T.decompress(d[name + "_bs"].tobytes(), ctx=ctx)
        assert np.array_equal(got, d[name + "_out"]), name


def test_module_level_api_is_reentrant(ctx, oracle):
    """The reference's compress/decompress/encode are pure functions (codec.py:26-189, no global state).  Here: 8 Python
    threads x 50 calls of the module-level API on different frames (every thread gets its own default context) equal the
    single-thread results byte for byte; and 4 threads hammering ONE explicitly shared Context (its calls serialise on the
    context's lock) do too."""
    import threading

    shapes = [(64, 64), (200, 328), (136, 520), (72, 1032), (512, 512), (33, 47), (256, 1024), (8, 8)]
    frames = [rand_frame(9000 + i, *shapes[i % len(shapes)]) for i in range(50)]
    quals = [10, 50, 90, 35, 75]
    want = [oracle.compress(f, quals[i % 5]) for i, f in enumerate(frames)]
    want_px = [oracle.decompress(b) for b in want]
    errors = []

    def worker(tid, use_ctx):
        try:
            order = list(range(50))
            np.random.default_rng(tid).shuffle(order)
            for i in order:
                bs = T.compress(frames[i], quals[i % 5], ctx=use_ctx)
                if bs != want[i]:
                    errors.append("thread %d frame %d: stream differs" % (tid, i))
                if not np.array_equal(T.decompress(bs, ctx=use_ctx), want_px[i]):
                    errors.append("thread %d frame %d: pixels differ" % (tid, i))
                e = T.encode(frames[i], quals[i % 5], ctx=use_ctx)
                if e["dc"].shape[0] != ((frames[i].shape[0] + 7) // 8) * ((frames[i].shape[1] + 7) // 8):
                    errors.append("thread %d frame %d: encode shape" % (tid, i))
        except Exception as ex:  # noqa: BLE001
            errors.append("thread %d: %r" % (tid, ex))

    seen = {}
    alive = threading.Barrier(3)

    def which_ctx(tid):
        seen[tid] = N.default_context().handle
        alive.wait(60)  # (a context freed by a finished thread could hand its address to the next one)

    for target, args_of, nthreads in ((worker, lambda t: (t, None), 8), (worker, lambda t: (t, ctx), 4), (which_ctx, lambda t: (t,), 3)):
        th = [threading.Thread(target=target, args=args_of(t)) for t in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    assert not errors, errors[:5]
    assert len(set(seen.values())) == 3, "default_context() must be per thread"


def test_batch_takes_registered_frames_in_place(ctx, oracle):
    """tic_compress_batch copies frames that lie in registered (pinned) memory to the device from where they are - no staging copy
    on the host - and stages pageable frames as before; the streams are the same either way and equal the oracle's.  The
    context reports its device's NUMA node (pinned buffers / NUMA node per GPU)."""
    L = N.load()
    n, h, w, q = 20, 136, 520, 50  # more than one chunk of 16
    block = np.stack([rand_frame(4000 + i, h, w) for i in range(n)])
    want = [oracle.compress(block[i], q) for i in range(n)]
    cap = L.tic_compress_bound(h, w)
    pool = np.empty((n, cap), dtype=np.uint8)
    outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
    caps = (C.c_size_t * n)(*([cap] * n))

    def run(ptrs):
        lens = (C.c_size_t * n)()
        inp = (C.c_void_p * n)(*ptrs)
        ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        d, s = C.c_int(), C.c_int()
        ctx.check(L.tic_last_batch_input_path(ctx.handle, C.byref(d), C.byref(s)))
        return [pool[i, : lens[i]].tobytes() for i in range(n)], d.value, s.value

    got, direct, staged = run([block[i].ctypes.data for i in range(n)])
    assert got == want and (direct, staged) == (0, n)
    ctx.check(L.tic_host_register(ctx.handle, block.ctypes.data, block.nbytes))
    try:
        got, direct, staged = run([block[i].ctypes.data for i in range(n)])          # contiguous frames: one copy per chunk
        assert got == want and (direct, staged) == (n, 0)
        order = list(range(n - 1, -1, -1))
        got, direct, staged = run([block[i].ctypes.data for i in order])               # scattered frames: one copy per frame
        assert got == [want[i] for i in order] and (direct, staged) == (n, 0)
    finally:
        ctx.check(L.tic_host_unregister(ctx.handle, block.ctypes.data))
    node, ncpus = C.c_int(-5), C.c_int(-5)
    ctx.check(L.tic_numa_info(ctx.handle, C.byref(node), C.byref(ncpus)))
    assert node.value >= -1 and ncpus.value >= 0
    ctx.check(L.tic_set_numa_binding(ctx.handle, 0))
    got, _, _ = run([block[i].ctypes.data for i in range(n)])
    ctx.check(L.tic_set_numa_binding(ctx.handle, 1))
    assert got == want


def test_batch_pins_pageable_frames_in_place(ctx, oracle, monkeypatch):
    """Round 4: pageable frames of a batch are registered IN PLACE for the duration of the call when one range covers them (frames
    allocated one after the other), and staged through the pinned slots otherwise: same streams on every route, equal to the
    oracle's; nothing stays registered behind the call; tic_set_auto_register(0) restores the staging route."""
    L = N.load()
    if not L.tic_build_has_test_hooks():
        pytest.skip("needs chunks of 16 small frames (TIC_BATCH_CHUNK, a test hook): round 6 sends up to 64 frames of this size per chunk")
    monkeypatch.setenv("TIC_BATCH_CHUNK", "16")
    n, h, w, q = 37, 520, 520, 50  # 270 KB per frame (above the 256 KB floor of the registered route), three chunks
    block = np.stack([rand_frame(5200 + i, h, w) for i in range(n)])
    want = [oracle.compress(block[i], q) for i in range(n)]
    cap = L.tic_compress_bound(h, w)
    pool = np.empty((n, cap), dtype=np.uint8)
    outp = (C.c_void_p * n)(*[pool[i].ctypes.data for i in range(n)])
    caps = (C.c_size_t * n)(*([cap] * n))

    def run(ptrs):
        lens = (C.c_size_t * n)()
        inp = (C.c_void_p * n)(*ptrs)
        pool[:] = 0
        ctx.check(L.tic_compress_batch(ctx.handle, inp, n, h, w, w, q, outp, caps, lens, 0))
        d, st, au = C.c_int(), C.c_int(), C.c_int()
        ctx.check(L.tic_last_batch_input_path(ctx.handle, C.byref(d), C.byref(st)))
        ctx.check(L.tic_last_batch_auto_registered(ctx.handle, C.byref(au)))
        return [pool[i, : lens[i]].tobytes() for i in range(n)], d.value, st.value, au.value

    dense = [block[i].ctypes.data for i in range(n)]
    got, direct, staged, auto = run(dense)                         # one range over the batch
    assert got == want and (direct, staged, auto) == (n, 0, n)
    ctx.check(L.tic_host_register(ctx.handle, block.ctypes.data, block.nbytes))   # nothing was left registered: this must succeed
    ctx.check(L.tic_host_unregister(ctx.handle, block.ctypes.data))
    ctx.check(L.tic_set_auto_register(ctx.handle, 0))
    try:
        got, direct, staged, auto = run(dense)                     # rounds 1-3: copy threads
        assert got == want and (direct, staged, auto) == (0, n, 0)
    finally:
        ctx.check(L.tic_set_auto_register(ctx.handle, 1))
    # frames that lie scattered (every second frame of a block four times the batch's size, back to front): the batch range is too
    # sparse, so are the chunks' ranges -> staged
    big = np.zeros((8 * n, h, w), dtype=np.uint8)
    for i in range(n):
        big[8 * (n - 1 - i)] = block[i]
    got, direct, staged, auto = run([big[8 * (n - 1 - i)].ctypes.data for i in range(n)])
    assert got == want and (direct, staged, auto) == (0, n, 0)
    # two dense groups far apart: the batch range is sparse, the chunks' ranges are not (chunk 2 straddles both groups -> staged)
    far = np.zeros((64 + n, h, w), dtype=np.uint8)
    far[:16] = block[:16]
    far[64 + 16:] = block[16:]
    ptrs = [far[i].ctypes.data for i in range(16)] + [far[64 + i].ctypes.data for i in range(16, n)]
    got, direct, staged, auto = run(ptrs)
    assert got == want and auto >= 16 and direct + staged == n
    # part of the batch registered by the caller already: the batch range cannot be registered over it; every frame still arrives
    ctx.check(L.tic_host_register(ctx.handle, block[:8].ctypes.data, block[:8].nbytes))
    try:
        got, direct, staged, auto = run(dense)
        assert got == want and direct + staged == n
    finally:
        ctx.check(L.tic_host_unregister(ctx.handle, block[:8].ctypes.data))
    # transform-only batch (tic_dctq_batch) takes the same route
    zz = [np.empty((L.tic_num_blocks(h, w), 64), np.int16) for _ in range(n)]
    inp = (C.c_void_p * n)(*dense)
    zp = (C.c_void_p * n)(*[z.ctypes.data for z in zz])
    ctx.check(L.tic_dctq_batch(ctx.handle, inp, n, h, w, w, q, zp))
    au = C.c_int()
    ctx.check(L.tic_last_batch_auto_registered(ctx.handle, C.byref(au)))
    assert au.value == n
    for i in (0, 17, n - 1):
        assert np.array_equal(zz[i], oracle.encode_zz16(block[i], q))


def test_bench_lines_name_their_scaling_baseline(tmp_path):
    """bench.py: --workload config4 on one GPU (the N = 1 point of a config-4 curve) and the two-rank rehearsal on the one GPU
    of the box, started AS THE DRIVER TYPES IT - `python bench.py --gpus 2 ...`, no launcher around it, no torch: bench.py spawns
    its own two rank processes (tinyimgcodec_amd/launch.py); TIC_BENCH_SHARE_GPU=1 puts both on device 0, where the exchange runs
    over the file communicator (RCCL needs a GPU per rank).  Both lines carry the same workload, per-rank rates and the name of
    the field an N > 1 value must be compared with; the gathered sizes of the two ranks are the first 16 of the one; every
    stream of both runs was checked against the reference's manifest (bench.py raises otherwise)."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--shard-frames", "8", "--steps", "2", "--warmup", "1", "--settle-ms", "1"]
    env0 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TIC_RDV_DIR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config4"] + small, capture_output=True, text=True, timeout=600, cwd=root, env=env0)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads(r.stdout.strip().splitlines()[-1])
    assert one["n_gpus"] == 1 and "config 4" in one["config"]["workload"] and one["scaling_baseline"]["value"].startswith("`config4.kernel_only_mpix_s`")
    assert one["config"]["per_rank"]["kernel_only_mpix_s"] == one["value"]
    assert one["config"]["parity"]["status"] == "ok" and one["config"]["parity"]["frames_checked"] == 8
    env = dict(env0, TIC_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + small, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([ln for ln in r.stdout.strip().splitlines() if ln.strip()]) == 1, r.stdout  # ONE JSON line on the job's stdout
    two = json.loads(r.stdout.strip())
    assert two["n_gpus"] == 2 and two["config"]["gathered_sizes"]["frames"] == 16 and "file communicator" in two["config"]["comm"]
    assert two["config"]["gathered_sizes"]["first"] == one["config"]["gathered_sizes"]["first"]
    assert abs(two["config"]["per_rank"]["kernel_only_mpix_s"] * 2 - two["value"]) < 1.0
    assert "NOT the N = 1 line's `value`" in two["scaling_baseline"]["value"]
    assert two["config"]["parity"]["status"] == "ok"
    # the torchrun form keeps working (torchrun only spawns: nothing in bench.py or the package imports torch)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + small,
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    three = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert three["config"]["gathered_sizes"] == two["config"]["gathered_sizes"]


def test_decoder_edges_round3(ctx, golden, monkeypatch):
    """decoder_edges.npz (reference-generated, round 3): same exception class as the reference for streams shorter than the header
    (struct.error) and for streams flagged as carrying a Huffman table (ValueError); same pixels for the stream the reference's
    adaptive-table writer produces; and for LONG damaged streams - the 1080p frame of BASELINE config 3 with a bit flipped in
    the middle, and cut at 60 % - the reference's pixels from BOTH host decoders (parallel and serial)."""
    import struct

    g = golden("decoder_edges")
    for name in [str(n) for n in g["names"]]:
        bs = g[name + "_bs"].tobytes()
        if int(g[name + "_ok"]):
            assert np.array_equal(T.decompress(bs, ctx=ctx), g[name + "_out"]), name
        else:
            exc = {"error": struct.error, "ValueError": ValueError}[str(g[name + "_exc"])]
            with pytest.raises(exc):
                T.decompress(bs, ctx=ctx)
    h, w = [int(v) for v in g["long_shape"]]
    s = T.compress(rand_frame(int(g["long_seed"]), h, w), 50, ctx=ctx)
    assert sha(s) == str(g["long_stream_sha"]) and len(s) == int(g["long_stream_len"])  # the stream the recipes apply to
    for name in [str(n) for n in g["long_names"]]:
        kind, at, mask = [int(v) for v in g[name + "_recipe"]]
        data = bytearray(s)
        if kind == 0:
            data[at] ^= mask
        else:
            data = data[:at]
        for serial in (False, True):
            if serial:
                monkeypatch.setenv("TIC_DECODE_SERIAL", "1")
            else:
                monkeypatch.delenv("TIC_DECODE_SERIAL", raising=False)
            out = T.decompress(bytes(data), ctx=ctx)
            assert sha(out.tobytes()) == str(g[name + "_sha"]), (name, serial)
            for (y, x), crop in zip(g["long_crops_at"], g[name + "_crops"]):
                assert np.array_equal(out[y:y + 64, x:x + 64], crop), (name, serial, int(y), int(x))
    monkeypatch.delenv("TIC_DECODE_SERIAL", raising=False)


def test_device_entropy_lane_kernel_and_its_fallback(oracle):
    """The device entropy stage packs with a lane per block (at most 512 bits per block) and runs again with the 8-lane kernel when
    a block needs more (noise at high quality): same bytes as the oracle either way, for single frames and for the batch pipeline,
    also when the overflow first shows up in the middle of a batch; sparse content (most of the walk skipped) and frames whose
    last partition is not full included."""
    c = T.Context(0)  # a fresh context: the fallback decision is remembered per context
    c.check(N.load().tic_set_entropy_lane_kernel(c.handle, 99))  # (the 8-lane kernel is the default: DESIGN.md 5.4)
    try:
        noise = rand_frame(5151, 264, 520)           # 33 x 65 blocks: the last wave holds 33 blocks
        smooth = np.add.outer(np.arange(264), np.arange(520)).astype(np.uint8)
        sparse = np.full((264, 520), 128, np.uint8)
        sparse[100:108, 200:208] = rand_frame(1, 8, 8)
        for q in (30, 50, 84, 85, 93, 96, 60):       # 93 / 96: blocks of 600-900 bits -> fallback; then back to a low quality
            for img in (noise, smooth, sparse):
                assert T.compress(img, q, ctx=c) == oracle.compress(img, q), q
        c2 = T.Context(0)
        c2.check(N.load().tic_set_entropy_lane_kernel(c2.handle, 99))
        try:
            frames = [rand_frame(6000 + i, 136, 264) for i in range(20)]
            frames[17] = np.where(rand_frame(7, 136, 264) > 127, 255, 0).astype(np.uint8)  # a few huge blocks late in the batch
            for q in (80, 84):
                got = T.compress_batch(frames, q, threads=0, ctx=c2)
                assert got == [oracle.compress(f, q) for f in frames], q
        finally:
            c2.close()
    finally:
        c.close()


def test_device_decoder_at_the_stream_end(ctx, oracle, monkeypatch):
    """The device decoder's first run takes the chain to the stream's END (no host tail on a whole stream); a stream whose end is not what
    a whole stream's is - cut inside its last blocks, cut at a block boundary, garbage or zeros behind its last block - must come out as
    the reference decodes it (codec.py:167-189: missing bits end a block's symbols, the block stays as far as it got; bytes behind the last
    block are ignored), through the second run with the 2,048-bit margin and the host's bit-serial tail.  Five frames: five different
    paddings and positions of the last block in the last range."""
    monkeypatch.setenv("TIC_TEST_HOOKS", "1")
    L = N.load()
    rbits, tries = C.c_int(), C.c_int()
    for seed, (h, w), q in ((81, (1536, 1536), 50), (82, (1536, 1544), 50), (83, (1600, 1536), 35), (84, (1536, 1536), 80), (85, (2048, 1024), 50),
                              (86, (520, 1001), 50), (87, (1024, 1024), 65)):  # (round 6: two that leave through host-mapped memory, one with ragged rows)
        img = rand_frame(seed, h, w)
        s = T.compress(img, q, ctx=ctx)
        assert len(s) * 8 >= 128 + (1 << 13)
        want = oracle.decompress(s)
        got = T.decompress(s, ctx=ctx)
        assert np.array_equal(got, want) and np.array_equal(want.shape, img.shape), (seed, q)
        assert L.tic_last_decode_path(ctx.handle) == 1 and L.tic_last_decode_giveup(ctx.handle) == 0
        assert L.tic_last_decode_range(ctx.handle, C.byref(rbits), C.byref(tries)) == 0 and tries.value == 1, (seed, q, tries.value)  # nothing left to the host, no second run
        monkeypatch.setenv("TIC_DECODE_MARGIN", "1")  # the same stream the way rounds 2-3 decoded it: margin + host tail
        assert np.array_equal(T.decompress(s, ctx=ctx), want), (seed, q, "margin")
        assert L.tic_last_decode_range(ctx.handle, C.byref(rbits), C.byref(tries)) == 0 and tries.value == 1
        monkeypatch.delenv("TIC_DECODE_MARGIN")
        rng = np.random.default_rng(seed)
        variants = {}
        for cut in (1, 2, 3, 5, 9, 17, 40, 100, 255, 256, 257, 300, 700):  # cut inside the last blocks (a block is ~27 bytes at q=50)
            variants["cut %d" % cut] = s[:-cut]
        variants["zeros behind"] = s + bytes(64)
        variants["one zero byte behind"] = s + bytes(1)
        variants["garbage behind"] = s + bytes(rng.integers(0, 256, 300, dtype=np.uint8))
        variants["ones behind"] = s + bytes([255] * 40)
        variants["last byte's padding set"] = s[:-1] + bytes([s[-1] | 0x01])
        for name, v in variants.items():
            want_v = oracle.decompress(v)
            got_v = T.decompress(v, ctx=ctx)
            assert np.array_equal(got_v, want_v), (seed, q, name)
            assert L.tic_last_decode_path(ctx.handle) in (1, 2)


def test_decompress_large_frame_device_against_host_decoder(ctx, monkeypatch):
    """A 4096 x 8192 noise frame (14 MB stream, 112,000 ranges, 524,288 blocks): the streams and the pixels of the device decoder AND of
    the host decoder against the pinned oracle's (tests/golden/big_frame_decode.json: the oracle ran once in the build container,
    tests/golden/gen/make_goldens_r5.py) - the only size at which the device decoder's two-level scans run deep."""
    import json

    L = N.load()
    with open(os.path.join(os.path.dirname(__file__), "golden", "big_frame_decode.json")) as f:
        want = {e["quality"]: e for e in json.load(f)["entries"] if (e["seed"], e["height"], e["width"]) == (8192, 4096, 8192)}
    img = rand_frame(8192, 4096, 8192)
    for q in (50, 85):
        assert (want[q]["seed"], want[q]["height"], want[q]["width"]) == (8192, 4096, 8192)
        s = T.compress(img, q, ctx=ctx)
        assert len(s) == want[q]["bytes"] and sha(s) == want[q]["sha256"], q
        monkeypatch.delenv("TIC_DECODE_HOST", raising=False)
        dev = T.decompress(s, ctx=ctx)
        assert L.tic_last_decode_path(ctx.handle) == 1 and L.tic_last_decode_giveup(ctx.handle) == 0, q
        assert sha(np.ascontiguousarray(dev).tobytes()) == want[q]["decoded_sha256"], q
        monkeypatch.setenv("TIC_DECODE_HOST", "1")
        host = T.decompress(s, ctx=ctx)
        assert L.tic_last_decode_path(ctx.handle) == 2
        monkeypatch.delenv("TIC_DECODE_HOST")
        assert sha(np.ascontiguousarray(host).tobytes()) == want[q]["decoded_sha256"], q


@pytest.mark.parametrize("q", [50, 10, 90])
def test_config5_frame_through_the_whole_codec_resident(ctx, q):
    """BASELINE config 5's frame (16384 x 16384, seed 1234) at q = 50, 10 and 90 through the whole codec with everything resident in HBM:
    tic_compress_dev's stream (115 MB at q = 50) and tic_decompress_dev's pixels against the pinned oracle's digests
    (tests/golden/big_frame_decode.json).  The only test in which the device decoder's launches exceed 4,096 workgroups without a test
    hook (16,384 and 27,000: the look-back through inclusive sums); the asynchronous decode of the same stream as well."""
    import json

    L = N.load()
    with open(os.path.join(os.path.dirname(__file__), "golden", "big_frame_decode.json")) as f:
        want = [e for e in json.load(f)["entries"] if (e["seed"], e["height"], e["width"], e["quality"]) == (1234, 16384, 16384, q)][0]
    h = w = 16384
    img = rand_frame(1234, h, w)
    cap = L.tic_compress_bound(h, w)
    d_img, d_str, d_pix = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap + 64, C.byref(d_str)))
    ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_pix)))
    try:
        ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
        del img
        n = C.c_size_t()
        ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, q, d_str, cap, C.byref(n)))
        assert n.value == want["bytes"]
        s = np.empty(n.value, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, s.ctypes.data, d_str, n.value))
        assert sha(s.tobytes()) == want["sha256"]
        del s
        pix = np.empty((h, w), np.uint8)
        for mode in ("synchronous", "synchronous again (two equal headers in a row)", "on a guess of the header", "asynchronous"):
            ctx.check(L.tic_memset_dev(ctx.handle, d_pix, 0xEE, h * w))
            if mode == "asynchronous":
                t = C.c_longlong(-1)
                ctx.check(L.tic_decompress_dev_async(ctx.handle, d_str, n.value, d_pix, w, h * w, C.byref(t)))
                hh, ww = C.c_int(), C.c_int()
                ctx.check(L.tic_decompress_async_result(ctx.handle, t.value, 1, C.byref(hh), C.byref(ww)))
                assert (hh.value, ww.value) == (h, w)
            else:
                ctx.check(L.tic_decompress_dev(ctx.handle, d_str, n.value, d_pix, w, h * w, None, None))
            assert L.tic_last_decode_path(ctx.handle) == 1 and L.tic_last_decode_giveup(ctx.handle) == 0, mode
            if mode in ("on a guess of the header", "asynchronous"):
                assert L.tic_last_decode_guess(ctx.handle) == 1, mode
            ctx.check(L.tic_memcpy_d2h(ctx.handle, pix.ctypes.data, d_pix, pix.size))
            assert sha(pix.tobytes()) == want["decoded_sha256"], mode
    finally:
        for p_ in (d_img, d_str, d_pix):
            L.tic_dev_free(ctx.handle, p_)


def test_the_references_own_benchmark_set(ctx, monkeypatch):
    """Round 5: the reference's benchmark workload - data/1..49.gif x quality 90, 80, 50, 20, 10, 5 (/root/reference/tests/benchmark.py:12-23),
    streams and decoded pixels taken from the unmodified reference (tests/golden/gen/make_goldens_r5.py; benchmark_set.json / .npz).
    All 294 pairs through compress(), through compress_batch() with the device entropy stage and with the host coder, through the
    resident tic_compress_dev, and back through decompress() (default path, serial host decoder) and the resident tic_decompress_dev."""
    import json

    L = N.load()
    gold = os.path.join(os.path.dirname(__file__), "golden")
    with open(os.path.join(gold, "benchmark_set.json")) as f:
        m = json.load(f)
    px = np.load(os.path.join(gold, "benchmark_set.npz"))["pixels"]
    assert px.shape == (49, 512, 512) and sha(px.tobytes()) == m["pixels_sha256"] and len(m["entries"]) == 294
    by_q = {}
    for e in m["entries"]:
        by_q.setdefault(e["quality"], {})[e["image"]] = e
    d_img, d_str, d_pix = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cap = L.tic_compress_bound(512, 512)
    ctx.check(L.tic_dev_alloc(ctx.handle, 512 * 512, C.byref(d_img)))
    ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_str)))
    ctx.check(L.tic_dev_alloc(ctx.handle, 512 * 512, C.byref(d_pix)))
    try:
        for q, ents in sorted(by_q.items()):
            assert sorted(ents) == list(range(1, 50))
            frames = [px[i - 1] for i in range(1, 50)]
            for threads in (4, 0):  # host coder / device entropy stage
                streams = T.compress_batch(frames, q, threads=threads, ctx=ctx)
                for i, s in enumerate(streams, 1):
                    assert len(s) == ents[i]["bytes"] and sha(s) == ents[i]["sha256"], (i, q, threads)
            zc = C.c_int(-1)  # round 6: one chunk, the mirror's pool of n x cap bytes: the read-back kernel stored all 49 streams straight into it
            ctx.check(L.tic_last_batch_zero_copy(ctx.handle, C.byref(zc)))
            assert zc.value == 49, (q, zc.value)
            # the reference's loop as two calls per quality (round 6): 49 images -> 49 streams -> 49 images, one chunk, every frame on the batch kernels
            images = T.decompress_batch(streams, ctx=ctx)
            nb, ns, nc, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            ctx.check(L.tic_last_decompress_batch(ctx.handle, C.byref(nb), C.byref(ns), C.byref(nc), C.byref(nd)))
            assert (nb.value, ns.value, nc.value, nd.value) == (49, 0, 1, 49), (q, nb.value, ns.value, nc.value, nd.value)
            for i, im in enumerate(images, 1):
                assert im.dtype == np.uint8 and im.shape == (512, 512) and sha(np.ascontiguousarray(im).tobytes()) == ents[i]["decoded_sha256"], (i, q, "decompress_batch")
            rbits, tries = C.c_int(), C.c_int()
            for i in range(1, 50):
                e, img = ents[i], px[i - 1]
                s = T.compress(img, q, ctx=ctx)
                assert len(s) == e["bytes"] and sha(s) == e["sha256"], (i, q)
                out = T.decompress(s, ctx=ctx)
                assert out.dtype == np.uint8 and sha(np.ascontiguousarray(out).tobytes()) == e["decoded_sha256"], (i, q)
                # round 6: every stream of the set - down to 8 bits per block at q = 5 - on the device decoder, in ONE run
                ctx.check(L.tic_last_decode_range(ctx.handle, C.byref(rbits), C.byref(tries)))
                assert L.tic_last_decode_path(ctx.handle) == 1 and tries.value == 1, (i, q, L.tic_last_decode_path(ctx.handle), tries.value)
                if i % 7 == 0:  # the serial host decoder and the resident pair on a spread of the images
                    monkeypatch.setenv("TIC_DECODE_SERIAL", "1")
                    assert sha(T.decompress(s, ctx=ctx).tobytes()) == e["decoded_sha256"], (i, q)
                    monkeypatch.delenv("TIC_DECODE_SERIAL")
                    n = C.c_size_t(0)
                    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, np.ascontiguousarray(img).ctypes.data, img.size))
                    ctx.check(L.tic_compress_dev(ctx.handle, d_img, 512, 512, 512, q, d_str, cap, C.byref(n)))
                    back = np.empty(n.value, dtype=np.uint8)
                    ctx.check(L.tic_memcpy_d2h(ctx.handle, back.ctypes.data, d_str, n.value))
                    assert n.value == e["bytes"] and sha(back) == e["sha256"], (i, q)
                    ctx.check(L.tic_decompress_dev(ctx.handle, d_str, n.value, d_pix, 512, 512 * 512, None, None))
                    pix = np.empty((512, 512), dtype=np.uint8)
                    ctx.check(L.tic_memcpy_d2h(ctx.handle, pix.ctypes.data, d_pix, pix.size))
                    assert sha(pix.tobytes()) == e["decoded_sha256"], (i, q)
    finally:
        for d in (d_img, d_str, d_pix):
            L.tic_dev_free(ctx.handle, d)


def test_decompress_batch_mixed_streams(ctx, oracle, golden):
    """decompress_batch() == [decompress(s) for s in streams] on a batch of everything: unequal geometries and qualities, dense and sparse
    content, ragged shapes (row pitch != width in the device buffer), an empty image, a stream too short for the device decoder, a
    C-encoder stream (scaled_dct), a cut stream and one with a flipped bit (both end on the host decoder), a 2048^2 frame of noise at
    q = 90 (the fused kernel's large window).  Through the C-ABI with scattered destinations too (the pinned route instead of the direct copy);
    and the exceptions of decompress() for a stream shorter than its header and for one flagged as carrying a table."""
    import struct

    L = N.load()
    rs = np.random.RandomState(5)
    imgs = [(rand_frame(31, 512, 512), 50), (rand_frame(32, 520, 776), 80), (np.full((512, 640), 77, np.uint8), 50), (rand_frame(33, 1080, 1920), 20),
            ((rand_frame(34, 600, 1000) // 64 * 64).astype(np.uint8), 10), (rand_frame(35, 203, 517), 90), (rand_frame(36, 64, 64), 50), (np.zeros((0, 8), np.uint8), 50),
            (rand_frame(37, 2048, 2048), 90), (np.tile(golden("lenna")["img"], (2, 2)), 5), (rand_frame(38, 512, 512), 50)]
    streams = [T.compress(im, q, ctx=ctx) for im, q in imgs]
    sc = golden("scaled_streams")
    streams.append(sc[str(sc["names"][0]) + "_bs"].tobytes())                      # a stream of the reference's C encoder
    cut = streams[0][: len(streams[0]) * 2 // 3]
    flip = bytearray(streams[3]); flip[len(flip) // 2] ^= 0x10
    streams += [cut, bytes(flip)]
    want = [oracle.decompress(s) for s in streams]
    got = T.decompress_batch(streams, ctx=ctx)
    assert len(got) == len(want)
    for k, (g, w_) in enumerate(zip(got, want)):
        assert g.dtype == np.uint8 and g.shape == w_.shape and np.array_equal(g, w_), k
    nb, ns, nc, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ctx.check(L.tic_last_decompress_batch(ctx.handle, C.byref(nb), C.byref(ns), C.byref(nc), C.byref(nd)))
    assert nb.value >= 8 and ns.value >= 3 and nb.value + ns.value == len(streams) - 1, (nb.value, ns.value)  # (the empty image takes neither)
    # scattered destinations, as a C caller might hold them: same pixels through the pinned route
    n = len(streams)
    bufs = [np.frombuffer(s, np.uint8) for s in streams]
    outs = [np.full(max(w_.size, 1) + 64, 0xCD, np.uint8) for w_ in want]
    hs, ws = (C.c_int * n)(), (C.c_int * n)()
    ctx.check(L.tic_decompress_batch(ctx.handle, (C.c_void_p * n)(*[b.ctypes.data for b in bufs]), (C.c_size_t * n)(*[b.size for b in bufs]), n,
                                     (C.c_void_p * n)(*[o.ctypes.data for o in outs]), (C.c_size_t * n)(*[w_.size for w_ in want]), hs, ws))
    for k, (o, w_) in enumerate(zip(outs, want)):
        assert (hs[k], ws[k]) == w_.shape and np.array_equal(o[: w_.size].reshape(w_.shape), w_) and np.all(o[w_.size:] == 0xCD), k
    ctx.check(L.tic_last_decompress_batch(ctx.handle, None, None, None, C.byref(nd)))
    assert nd.value == 0
    # a destination too small: the error of the first such frame, before anything is decoded
    caps = (C.c_size_t * n)(*[w_.size for w_ in want])
    caps[4] -= 1
    assert L.tic_decompress_batch(ctx.handle, (C.c_void_p * n)(*[b.ctypes.data for b in bufs]), (C.c_size_t * n)(*[b.size for b in bufs]), n,
                                  (C.c_void_p * n)(*[o.ctypes.data for o in outs]), caps, None, None) == N.TIC_E_SPACE
    assert "frame 4" in L.tic_last_error(ctx.handle).decode()
    with pytest.raises(struct.error):
        T.decompress_batch([streams[0], streams[1][:10]], ctx=ctx)
    tab = bytearray(streams[0]); tab[15] |= 0x80
    with pytest.raises(ValueError):
        T.decompress_batch([streams[0], bytes(tab)], ctx=ctx)
    assert T.decompress_batch([], ctx=ctx) == []


def test_decompress_dev_resident_round_trip(ctx, oracle):
    """tic_compress_dev -> tic_decompress_dev with image, stream and pixels resident in device memory: the pixels equal decompress() of the
    same stream (long streams through the device Huffman decoder, short ones through the host decoder), also with a padded output
    pitch and a ragged frame; error paths."""
    L = N.load()
    for (h, w), q, pad in (((2048, 2048), 50, 0), ((1500, 1999), 75, 49), ((1500, 1999), 75, 0), ((512, 512), 50, 0), ((64, 72), 30, 8), ((200, 333), 60, 4)):
        img = rand_frame(h * 7 + w, h, w)
        cap = L.tic_compress_bound(h, w)
        stride = w + pad
        d_img, d_str, d_pix = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ctx.check(L.tic_dev_alloc(ctx.handle, img.size, C.byref(d_img)))
        ctx.check(L.tic_dev_alloc(ctx.handle, cap, C.byref(d_str)))
        ctx.check(L.tic_dev_alloc(ctx.handle, h * stride, C.byref(d_pix)))
        ctx.check(L.tic_memcpy_h2d(ctx.handle, d_img, img.ctypes.data, img.size))
        n = C.c_size_t()
        ctx.check(L.tic_compress_dev(ctx.handle, d_img, h, w, w, q, d_str, cap, C.byref(n)))
        ctx.check(L.tic_memset_dev(ctx.handle, d_pix, 0xEE, h * stride))
        hh, ww = C.c_int(), C.c_int()
        ctx.check(L.tic_decompress_dev(ctx.handle, d_str, n.value, d_pix, stride, h * stride, C.byref(hh), C.byref(ww)))
        assert (hh.value, ww.value) == (h, w)
        long_enough = device_decoder_takes(L.tic_num_blocks(h, w), n.value)
        assert L.tic_last_decode_path(ctx.handle) == (1 if long_enough else 2), (h, w)
        pix = np.empty((h, stride), np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, pix.ctypes.data, d_pix, pix.size))
        stream = np.empty(n.value, np.uint8)
        ctx.check(L.tic_memcpy_d2h(ctx.handle, stream.ctypes.data, d_str, n.value))
        want = oracle.decompress(stream.tobytes())
        assert np.array_equal(pix[:, :w], want), (h, w, q)
        assert np.all(pix[:, w:] == 0xEE), "bytes between the rows stay untouched"
        # error paths: output too small, stride below the width, a stream cut inside the header
        assert L.tic_decompress_dev(ctx.handle, d_str, n.value, d_pix, stride, (h - 1) * stride + w - 1, None, None) == N.TIC_E_SPACE
        assert L.tic_decompress_dev(ctx.handle, d_str, n.value, d_pix, w - 1, h * stride, None, None) == N.TIC_E_ARG
        assert L.tic_decompress_dev(ctx.handle, d_str, 15, d_pix, stride, h * stride, None, None) == N.TIC_E_STREAM
        for p in (d_img, d_str, d_pix):
            L.tic_dev_free(ctx.handle, p)


def test_decompress_dev_launches_on_a_guess_of_the_header(ctx, oracle, monkeypatch):
    """tic_decompress_dev decodes a long stream on a guess of its header (the header of the stream the context decoded last) and reads the
    header only when the kernels' echo of it differs: same geometry and quality -> the guess holds; another quality, another geometry, a
    destination the guessed geometry does not fit, a short stream -> the stream is decoded with its own header; pixels equal the oracle's
    every time, the errors are the ones of a call that read the header first."""
    L = N.load()
    specs = [((2048, 2048), 50, 1), ((2048, 2048), 50, 2), ((2048, 2048), 50, 8), ((2048, 2048), 80, 3), ((1504, 2000), 50, 4), ((2048, 2048), 50, 5), ((200, 240), 50, 6),
             ((2048, 2048), 50, 7), ((2048, 2048), 50, 9)]
    streams = {}
    bufs = []
    for (h, w), q, seed in specs:
        s = np.frombuffer(T.compress(rand_frame(900 + seed, h, w), q, ctx=ctx), np.uint8)
        d_s, d_p = C.c_void_p(), C.c_void_p()
        ctx.check(L.tic_dev_alloc(ctx.handle, s.size + 64, C.byref(d_s)))
        ctx.check(L.tic_dev_alloc(ctx.handle, h * w, C.byref(d_p)))
        ctx.check(L.tic_memcpy_h2d(ctx.handle, d_s, s.ctypes.data, s.size))
        bufs.append((d_s, d_p, s, h, w))
    ctx2 = T.Context(0)  # a fresh context: no stream decoded yet, no guess
    # a guess is made after two equal headers in a row: calls 0 and 1 read the header, call 2 guesses and holds; then another quality (the guess
    # fails, and the streak starts over), another geometry and back (no guess: no two equal headers in a row), a short stream (200 x 240: host decoder, no
    # guess, nothing remembered), the same header again (second in a row: still read), and again (guessed, holds)
    want_guess = [0, 0, 1, -1, 0, 0, 0, 0, 1]
    for k, (d_s, d_p, s, h, w) in enumerate(bufs):
        ctx2.check(L.tic_memset_dev(ctx2.handle, d_p, 0xEE, h * w))
        hh, ww = C.c_int(), C.c_int()
        ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, C.byref(hh), C.byref(ww)))
        assert (hh.value, ww.value) == (h, w)
        assert L.tic_last_decode_guess(ctx2.handle) == want_guess[k], (k, L.tic_last_decode_guess(ctx2.handle))
        pix = np.empty((h, w), np.uint8)
        ctx2.check(L.tic_memcpy_d2h(ctx2.handle, pix.ctypes.data, d_p, pix.size))
        assert np.array_equal(pix, oracle.decompress(s.tobytes())), k
    # a destination too small for the guessed geometry (which is this stream's): the error of a call that read the header
    d_s, d_p, s, h, w = bufs[0]
    assert L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w - 1, None, None) == N.TIC_E_SPACE
    assert L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w - 1, h * w, None, None) == N.TIC_E_ARG
    # ... and a larger stream after two smaller ones: the guessed geometry fits the destination, the echo differs
    d_s4, d_p4, s4, h4, w4 = bufs[4]
    for _ in range(2):
        ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s4, s4.size, d_p4, w4, h4 * w4, None, None))
    ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, None, None))
    assert L.tic_last_decode_guess(ctx2.handle) == -1
    pix = np.empty((h, w), np.uint8)
    ctx2.check(L.tic_memcpy_d2h(ctx2.handle, pix.ctypes.data, d_p, pix.size))
    assert np.array_equal(pix, oracle.decompress(s.tobytes()))
    for _ in range(2):
        ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, None, None))
    assert L.tic_last_decode_guess(ctx2.handle) == 1
    ctx2.check(L.tic_set_decode_guess(ctx2.handle, 0))  # the switch: no guessing on this context
    ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, None, None))
    assert L.tic_last_decode_guess(ctx2.handle) == 0
    ctx2.check(L.tic_set_decode_guess(ctx2.handle, 1))
    ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, None, None))
    assert L.tic_last_decode_guess(ctx2.handle) == 1
    if L.tic_build_has_test_hooks():
        monkeypatch.setenv("TIC_DECODE_NO_GUESS", "1")
        ctx2.check(L.tic_decompress_dev(ctx2.handle, d_s, s.size, d_p, w, h * w, None, None))
        assert L.tic_last_decode_guess(ctx2.handle) == 0
        monkeypatch.delenv("TIC_DECODE_NO_GUESS")
    for d_s, d_p, *_ in bufs:
        L.tic_dev_free(ctx.handle, d_s); L.tic_dev_free(ctx.handle, d_p)
    ctx2.close()


def test_decompress_dev_wrong_guess_writes_nothing_outside_the_image(ctx, oracle):
    """A decode launched on a guessed header that turns out wrong must not touch a byte outside the real image (round-5 advice: the fused
    kernel used to write the GUESSED h x w before the header was known): two large frames, then a small one, into a window of a
    sentinel-filled surface whose stride and capacity admit the guessed geometry - synchronously and through the asynchronous tickets."""
    L = N.load()
    H, W = 1024, 1536          # the surface (and the guessed geometry: two frames of this size come first)
    h, w = 520, 776            # the frame that follows
    big = np.frombuffer(T.compress(rand_frame(4711, H, W), 50, ctx=ctx), np.uint8)
    small = np.frombuffer(T.compress(rand_frame(4712, h, w), 50, ctx=ctx), np.uint8)
    want_big, want_small = oracle.decompress(big.tobytes()), oracle.decompress(small.tobytes())
    d_big, d_small, d_surf = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx.check(L.tic_dev_alloc(ctx.handle, big.size + 64, C.byref(d_big)))
    ctx.check(L.tic_dev_alloc(ctx.handle, small.size + 64, C.byref(d_small)))
    ctx.check(L.tic_dev_alloc(ctx.handle, H * W, C.byref(d_surf)))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_big, big.ctypes.data, big.size))
    ctx.check(L.tic_memcpy_h2d(ctx.handle, d_small, small.ctypes.data, small.size))
    ctx2 = T.Context(0)
    surf = np.empty((H, W), np.uint8)
    try:
        for mode in ("sync", "async"):
            for _ in range(2):  # two equal headers in a row: the next call guesses H x W
                ctx2.check(L.tic_decompress_dev(ctx2.handle, d_big, big.size, d_surf, W, H * W, None, None))
            ctx2.check(L.tic_memset_dev(ctx2.handle, d_surf, 0xA5, H * W))
            hh, ww = C.c_int(), C.c_int()
            if mode == "sync":
                ctx2.check(L.tic_decompress_dev(ctx2.handle, d_small, small.size, d_surf, W, H * W, C.byref(hh), C.byref(ww)))
            else:
                t = C.c_longlong()
                ctx2.check(L.tic_decompress_dev_async(ctx2.handle, d_small, small.size, d_surf, W, H * W, C.byref(t)))
                ctx2.check(L.tic_decompress_async_result(ctx2.handle, t.value, 1, C.byref(hh), C.byref(ww)))
            assert (hh.value, ww.value) == (h, w)
            assert L.tic_last_decode_guess(ctx2.handle) == -1, mode  # the guess was made, and was wrong
            ctx2.check(L.tic_memcpy_d2h(ctx2.handle, surf.ctypes.data, d_surf, surf.size))
            assert np.array_equal(surf[:h, :w], want_small), mode
            assert np.all(surf[h:, :] == 0xA5) and np.all(surf[:h, w:] == 0xA5), "%s: bytes outside the %dx%d image were written" % (mode, h, w)
        ctx2.check(L.tic_decompress_dev(ctx2.handle, d_big, big.size, d_surf, W, H * W, None, None))
        ctx2.check(L.tic_memcpy_d2h(ctx2.handle, surf.ctypes.data, d_surf, surf.size))
        assert np.array_equal(surf, want_big)
    finally:
        ctx2.close()
        for p in (d_big, d_small, d_surf):
            L.tic_dev_free(ctx.handle, p)


def test_decompress_dev_async_matches_the_synchronous_call(ctx, oracle):
    """tic_decompress_dev_async / tic_decompress_async_result: frames launched on the guess of their header on streams of their own,
    four tickets open at a time - every ticket's outcome (pixels, geometry, return code) is the synchronous call's: same geometry and
    quality (the launch stands), another quality and a damaged stream (decoded again at collection), a short stream and a first call
    without any guess (run synchronously at once); a fifth open ticket and a closed ticket are refused."""
    L = N.load()
    ctx2 = T.Context(0)
    A = ((2048, 2048), 50)
    # (a launch on the guess needs two equal headers in a row before it: the first two frames, and the ones right behind a change, run synchronously)
    specs = [A + (0,), A + (0,), A + (0,), A + (0,), ((2048, 2048), 80, 0), A + (0,), A + (0,), A + (0,), A + (1,), ((200, 240), 50, 0), A + (0,), A + (0,), A + (2,),
             ((1504, 2000), 60, 0), A + (0,)]
    jobs = []
    for k, ((h, w), q, damage) in enumerate(specs):
        s = bytearray(T.compress(rand_frame(1300 + k, h, w), q, ctx=ctx))
        if damage == 1:
            s[len(s) // 2 + 11] ^= 0x10  # a flipped bit in the middle
        elif damage == 2:
            s = s[: len(s) * 6 // 10]     # cut short
        s = np.frombuffer(bytes(s), np.uint8)
        d_s, d_p = C.c_void_p(), C.c_void_p()
        ctx2.check(L.tic_dev_alloc(ctx2.handle, s.size + 64, C.byref(d_s)))
        ctx2.check(L.tic_dev_alloc(ctx2.handle, h * w, C.byref(d_p)))
        ctx2.check(L.tic_memcpy_h2d(ctx2.handle, d_s, s.ctypes.data, s.size))
        jobs.append((d_s, d_p, s, h, w, oracle.decompress(s.tobytes())))

    def collect(k, ticket):
        d_s, d_p, s, h, w, want = jobs[k]
        hh, ww = C.c_int(), C.c_int()
        rc = L.tic_decompress_async_result(ctx2.handle, ticket, 0, C.byref(hh), C.byref(ww))
        if rc == N.TIC_E_BUSY:
            rc = L.tic_decompress_async_result(ctx2.handle, ticket, 1, C.byref(hh), C.byref(ww))
        assert rc == 0, (k, rc, L.tic_last_error(ctx2.handle))
        assert (hh.value, ww.value) == (h, w), k
        pix = np.empty((h, w), np.uint8)
        ctx2.check(L.tic_memcpy_d2h(ctx2.handle, pix.ctypes.data, d_p, pix.size))
        assert np.array_equal(pix, want), k
        assert L.tic_decompress_async_result(ctx2.handle, ticket, 1, None, None) == N.TIC_E_ARG  # closed

    open_tickets = []
    for k, (d_s, d_p, s, h, w, want) in enumerate(jobs):
        ctx2.check(L.tic_memset_dev(ctx2.handle, d_p, 0xEE, h * w))
        t = C.c_longlong(-1)
        ctx2.check(L.tic_decompress_dev_async(ctx2.handle, d_s, s.size, d_p, w, h * w, C.byref(t)))
        open_tickets.append((k, t.value))
        if len(open_tickets) == 4:
            if k == 3:  # four tickets are open: a fifth is refused, and refusing it changes nothing
                t5 = C.c_longlong(-1)
                assert L.tic_decompress_dev_async(ctx2.handle, d_s, s.size, d_p, w, h * w, C.byref(t5)) == N.TIC_E_ARG
            collect(*open_tickets.pop(0))
    ctx2.check(L.tic_sync(ctx2.handle))  # (waits for the asynchronous decodes too)
    while open_tickets:
        collect(*open_tickets.pop(0))
    assert L.tic_decompress_async_result(ctx2.handle, 10 ** 6, 1, None, None) == N.TIC_E_ARG
    for d_s, d_p, *_ in jobs:
        L.tic_dev_free(ctx2.handle, d_s); L.tic_dev_free(ctx2.handle, d_p)
    ctx2.close()


def test_small_frames_leave_compress_through_host_mapped_memory(ctx, oracle, monkeypatch):
    """tic_compress of a frame whose stream is at most 2 MB long lets the placing kernel write the stream into host-mapped memory and copies
    it out with memcpy (no device-to-host DMA copy); larger frames, and every frame under TIC_NO_SMALL_PATH, go through the device stream
    buffer: the same bytes either way, the reference's bytes, also right at the line and with a destination that is too small."""
    L = N.load()
    for (h, w), q in (((512, 512), 50), ((200, 333), 75), ((8, 8), 50), ((840, 768), 90), ((848, 768), 90), ((1080, 1920), 50)):  # (840 x 768: the last frame below the line, 848 x 768 the first above)
        img = rand_frame(h * 3 + w, h, w)
        want = oracle.compress(img, q)
        monkeypatch.delenv("TIC_NO_SMALL_PATH", raising=False)
        a = T.compress(img, q, ctx=ctx)
        monkeypatch.setenv("TIC_NO_SMALL_PATH", "1")
        b = T.compress(img, q, ctx=ctx)
        monkeypatch.delenv("TIC_NO_SMALL_PATH")
        assert a == want and b == want, (h, w, q, len(a), len(b), len(want))
        out = np.zeros(len(want) - 1, np.uint8)
        n = C.c_size_t()
        assert L.tic_compress(ctx.handle, img.ctypes.data, h, w, w, q, out.ctypes.data, out.size, C.byref(n)) == N.TIC_E_SPACE


def test_frames_beyond_the_32bit_walk_are_transformed_in_bands(ctx, monkeypatch):
    """The strip walk uses 32-bit pixel offsets; a frame of 4 GiB or more is cut into bands of whole block rows, one launch each
    (round 2 ran such frames on the exact kernel only).  TIC_BAND_BYTES lowers the limit so that a 3000 x 2112 frame is cut into
    several bands, the last one with bottom padding: same coefficients as the exact kernel."""
    img = rand_frame(31, 3003, 2112)
    f = DevFrame(ctx, img)
    ref = f.run(50, N.KERNEL_EXACT)
    for limit in (str(2304 * 1000), str(2304 * 8), str(2304 * 2999)):
        monkeypatch.setenv("TIC_BAND_BYTES", limit)
        assert np.array_equal(f.run(50, N.KERNEL_HYBRID), ref), limit
    monkeypatch.delenv("TIC_BAND_BYTES")
    assert np.array_equal(f.run(50, N.KERNEL_HYBRID), ref)
    f.free()


def test_decompress_long_streams_on_the_device_decoder(ctx, oracle, golden, monkeypatch):
    """Streams of >= 16,384 blocks are Huffman-decoded on the device (speculative measuring per range, parallel stitch, block
    positions, a lane per block decoding, DC prefix sum): same pixels as the host decoders and as the oracle on noise at three qualities, natural, smooth and
    sparse content, a frame whose block count is not a multiple of anything convenient, and streams damaged in the middle or cut
    short (there the device decoder either succeeds on the true chain or gives up and the host's bit-serial path takes over)."""
    L = N.load()
    lenna = golden("lenna")["img"]
    frames = {
        "noise 2048x2048": (rand_frame(71, 2048, 2048), (10, 50, 90)),
        "noise ragged 1500x1999": (rand_frame(72, 1500, 1999), (50,)),
        "noise ragged 517x1003": (rand_frame(74, 517, 1003), (20, 90)),  # (round 6: at most 1 MB of pixels - out through host-mapped memory, row by row)
        "noise 1024x1032": (rand_frame(75, 1024, 1032), (50,)),  # (just above that bound)
        "lenna tiled 2048x2048": (np.ascontiguousarray(np.tile(lenna, (4, 4))), (50, 90)),
        "smooth": (np.add.outer(np.arange(1600) // 3, np.arange(1800) // 5).astype(np.uint8), (50,)),
        "smooth, large": (np.add.outer(np.arange(4096) // 3, np.arange(6144) // 5).astype(np.uint8), (90,)),
        "sparse": (np.where(rand_frame(73, 1536, 1536) > 253, 255, 128).astype(np.uint8), (50,)),
    }
    for name, (img, quals) in frames.items():
        for q in quals:
            s = T.compress(img, q, ctx=ctx)
            want = oracle.decompress(s)
            monkeypatch.delenv("TIC_DECODE_HOST", raising=False)
            got = T.decompress(s, ctx=ctx)
            long_enough = device_decoder_takes(((img.shape[0] + 7) // 8) * ((img.shape[1] + 7) // 8), len(s))  # (shorter streams decode serially on the host)
            assert L.tic_last_decode_path(ctx.handle) == (1 if long_enough else 2), (name, q, len(s))
            if long_enough:
                assert L.tic_last_decode_giveup(ctx.handle) == 0, (name, q)
                rbits, tries = C.c_int(), C.c_int()
                assert L.tic_last_decode_range(ctx.handle, C.byref(rbits), C.byref(tries)) == 0
                avg = (len(s) * 8) / (((img.shape[0] + 7) // 8) * ((img.shape[1] + 7) // 8))
                floor = 1056 if avg < 7 else 288  # (round 6: two average blocks, at least 288 bits; nearly flat content gets 1,056)
                assert tries.value == 1 and rbits.value % 64 == 32 and max(floor, 2 * avg - 32) <= rbits.value <= max(floor, 2 * avg + 64), \
                    (name, q, rbits.value, tries.value, avg)  # the first choice of range held: no second run
            assert np.array_equal(got, want), (name, q)
            # the sums inside the two kernels (block counts, DC differences): launches above TIC_DECODE_FLAT_GRID workgroups look back through
            # inclusive sums (a 16384^2 stream in production; here forced on every launch, and on the larger part of them)
            for fg in ("0", "100"):
                monkeypatch.setenv("TIC_DECODE_FLAT_GRID", fg)
                assert np.array_equal(T.decompress(s, ctx=ctx), want), (name, q, "flat grid", fg)
                assert L.tic_last_decode_path(ctx.handle) == (1 if long_enough else 2) and L.tic_last_decode_giveup(ctx.handle) == 0
            monkeypatch.delenv("TIC_DECODE_FLAT_GRID")
            if name == "noise 2048x2048":  # forced range lengths, and the second try with the longest range when 288 or 544 bits are
                # shorter than the blocks (q=90: 404 bits)
                for rb in ("288", "544", "928", "1056", "2016"):
                    monkeypatch.setenv("TIC_DECODE_RANGE", rb)
                    assert np.array_equal(T.decompress(s, ctx=ctx), want), (name, q, rb)
                    assert L.tic_last_decode_path(ctx.handle) == 1, (name, q, rb)
                    assert L.tic_last_decode_range(ctx.handle, C.byref(rbits), C.byref(tries)) == 0
                    assert (rbits.value, tries.value) in ((int(rb), 1), (2016, 2)), (name, q, rb, rbits.value, tries.value)
                    if q == 90 and rb == "288":  # blocks of 404 bits pass over whole ranges of 288 bits: rounds 2-5 needed the second run with the
                        assert tries.value == 1, (name, q, rb)  # longest range here, round 6's stitch follows the chain from range to range
                monkeypatch.delenv("TIC_DECODE_RANGE")
            monkeypatch.setenv("TIC_DECODE_HOST", "1")
            assert np.array_equal(T.decompress(s, ctx=ctx), want), (name, q, "host")
            assert L.tic_last_decode_path(ctx.handle) == 2
            monkeypatch.delenv("TIC_DECODE_HOST")
            if q == 50:  # damage: a flipped bit in the middle, a cut at 70 %, garbage behind a cut
                rng = np.random.default_rng(len(s))
                for k in range(3):
                    bad = bytearray(s)
                    if k == 0:
                        bad[len(s) // 2 + 7] ^= 0x04
                    elif k == 1:
                        bad = bad[: len(s) * 7 // 10]
                    else:
                        bad = bad[: len(s) // 3] + bytes(rng.integers(0, 256, 4096, dtype=np.uint8))
                    assert np.array_equal(T.decompress(bytes(bad), ctx=ctx), oracle.decompress(bytes(bad))), (name, q, k)
                    # (when the device decoder left the stream to the host it says why: a non-zero set of DecStatus::giveup bits)
                    assert (L.tic_last_decode_path(ctx.handle) == 1) == (L.tic_last_decode_giveup(ctx.handle) == 0) or not device_decoder_takes(((img.shape[0] + 7) // 8) * ((img.shape[1] + 7) // 8), len(bad))
