"""CPU-side tests of the product's host logic and of the C-ABI surface (no GPU compute)."""
import ctypes
import hashlib
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, rand_frame

import tinyimgcodec_amd as T
from tinyimgcodec_amd import _native as N


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/*.h must be exported by the shared library and bound in _native."""
    hdr = open(os.path.join(ROOT, "include", "tinyimgcodec_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tic_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for path, hooks in ((N.LIB_PATH, 0), (N.HOOKS_LIB_PATH, 1)):
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), "%s does not export %s" % (os.path.basename(path), name)
        # the shipped library reads no environment: the hook gate is compiled out of it (csrc/tic_hooks.h), and with it every
        # variable name; the test-hooks build of the same sources says what it is
        assert lib.tic_build_has_test_hooks() == hooks
        blob = open(path, "rb").read()
        for var in (b"TIC_SCHED", b"TIC_SPLIT", b"TIC_DECODE_HOST", b"TIC_DECODE_SERIAL", b"TIC_COMM_FORCE_RCCL", b"TIC_BAND_BYTES", b"TIC_TEST_HOOKS"):
            assert (var in blob) == bool(hooks) or var == b"TIC_TEST_HOOKS", (os.path.basename(path), var)
        assert b"TIC_TEST_HOOKS" not in blob  # (neither build reads that one: _native.py picks the build by it)
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    assert N.load().tic_build_has_test_hooks() == 1  # tests/conftest.py sets TIC_TEST_HOOKS=1: this process runs the test-hooks build
    L = N.load()
    assert b"gfx950" in L.tic_version()
    assert L.tic_num_blocks(1080, 1920) == 32400 and L.tic_num_blocks(9, 9) == 4 and L.tic_num_blocks(0, 8) == 0
    assert L.tic_compress_bound(0, 8) >= 16


def test_no_silent_cpu_fallback():
    """Without a device the codec must fail loudly, never compute on the CPU."""
    if N.load().tic_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(T.NativeUnavailable):
        T.compress(np.zeros((8, 8), np.uint8))
    with pytest.raises(T.NativeUnavailable):
        T.encode(np.zeros((8, 8), np.uint8))


def test_product_does_not_import_oracle():
    """The product package must not reference oracle/ (checked textually over its sources)."""
    pkg = os.path.join(ROOT, "tinyimgcodec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "tic_oracle" not in txt and "libtic_oracle" not in txt, f


def test_entropy_stage_matches_reference_streams(oracle, golden, manifest):
    """Host entropy stage (tic_entropy_encode) fed with oracle coefficients == reference compress() bytes."""
    d = golden("lenna")
    for q in (10, 50, 90):
        bs = T.entropy_encode(oracle.encode_zz16(d["img"], q), 512, 512, q)
        assert bs == d[f"q{q}_bs"].tobytes()
    s = golden("transform_small")
    for key in s["names"]:
        img = s[key + "_img"]
        q = int(str(key).rsplit("_q", 1)[1])
        want = s[key + "_bs"].tobytes()
        zz = oracle.encode_zz16(img, q)
        if want:
            assert T.entropy_encode(zz, img.shape[0], img.shape[1], q) == want, key
        else:
            with pytest.raises(KeyError):
                T.entropy_encode(zz, img.shape[0], img.shape[1], q)
    sw = golden("quality_sweep")
    for q in sw["qualities"]:
        q = int(q)
        want = sw[f"q{q}_bs"].tobytes()
        zz = oracle.encode_zz16(sw["img"], q)
        if want:
            assert T.entropy_encode(zz, 64, 96, q) == want, q
    img = rand_frame(1234, 1080, 1920)
    bs = T.entropy_encode(oracle.encode_zz16(img, 50), 1080, 1920, 50)
    assert sha(bs) == manifest["rand1234_1080x1920_q50"]["sha256"]


def test_entropy_rle_known_answers(manifest):
    """huffman.py:12-33 known answers through the product's bit stream (decoded with the table digests' codes)."""
    ka = manifest["rle_known_answers"]
    # {62: -3} -> 3 x ZRL, (14,-3), EOB ; stream = header + DC(0)="00" + 3x"11111111001" + code(14,2)+"00" + "1010"
    zz = np.zeros((1, 64), np.int16)
    zz[0, 63] = -3
    bs = T.entropy_encode(zz, 8, 8, 50)
    bits = "".join(f"{b:08b}" for b in bs[16:])
    assert ka["last_only"]["rle"] == [[15, 0]] * 3 + [[14, -3], [0, 0]]
    assert bits.startswith("00" + "11111111001" * 3 + "1111111111101100" + "00" + "1010")
    # all-zero block -> DC "00" + EOB "1010", zero padded
    bs = T.entropy_encode(np.zeros((1, 64), np.int16), 8, 8, 50)
    assert bs[16:] == bytes([0b00101000])


def test_header_layout_and_parse():
    bs = T.entropy_encode(np.zeros((0, 64), np.int16), 0, 8, 50)
    assert bs == struct.pack("<IIII", 0, 8, 50, 0)
    hdr = T.parse_header(struct.pack("<IIII", 512, 300, 77, 0) + b"\x00")
    assert hdr == {"height": 512, "width": 300, "quality": 77, "flag": 0}
    with pytest.raises(struct.error):
        T.parse_header(b"\x00" * 5)


def test_quality_argument_errors_mirror_reference(manifest):
    """Exception types for invalid quality follow the reference (recorded in manifest['error_behaviour'])."""
    e = manifest["error_behaviour"]
    img = rand_frame(3, 8, 8)
    assert e["quality_0"]["exc"] == "ZeroDivisionError"
    with pytest.raises(ZeroDivisionError):
        T.compress(img, 0)
    assert e["quality_100"]["exc"] == "KeyError"
    with pytest.raises(KeyError):
        T.compress(img, 100)
    assert e["quality_float"]["exc"] == "error" and e["quality_negative"]["exc"] == "error"
    with pytest.raises(struct.error):
        T.compress(img, 50.0)
    with pytest.raises(struct.error):
        T.compress(img, -5)
    assert e["ndim_3"]["exc"] == "ValueError" and e["ndim_1"]["exc"] == "ValueError"
    with pytest.raises(ValueError):
        T.compress(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(ValueError):
        T.compress(np.zeros((8,), np.uint8))
    with pytest.raises(NotImplementedError):
        T.compress(img, 50, auto_generate_huffman_table=True)


def test_entropy_errors():
    zz = np.zeros((1, 64), np.int16)
    zz[0, 5] = 1024  # size 11 has no AC code
    with pytest.raises(KeyError):
        T.entropy_encode(zz, 8, 8, 50)
    zz[0, 5] = 1023
    assert len(T.entropy_encode(zz, 8, 8, 50)) > 16


def test_host_entropy_coder_under_sanitizers(tmp_path):
    """The product's host Huffman/RLE coder and decoder, built with AddressSanitizer + UBSan, against the oracle's coder
    on dense / sparse / maximal / long-run coefficient blocks: same streams, exact-size and undersized buffers, round
    trips, truncated and corrupted streams (GPU sanitizers are not available on the pool; this is the CPU build); and the device
    decoder's chain tables (a chain of symbols per look-up) against the one-symbol walk: the same block ends from 1,200 random bits of
    dense, short-block and random streams."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None or shutil.which("gcc") is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "host_selftest"
    obj = tmp_path / "tic_oracle.o"
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    subprocess.run(["gcc", "-c", *flags, "-o", str(obj), os.path.join(root, "oracle", "tic_oracle.c")], check=True)
    subprocess.run(["g++", "-std=c++17", *flags, "-o", str(exe), os.path.join(root, "tests", "native", "host_selftest.cpp"),
                    os.path.join(root, "tinyimgcodec_amd", "csrc", "tic_entropy.cpp"), str(obj), "-lm", "-lpthread"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_selftest ok" in r.stdout


def test_guard_band_of_the_fast_path_is_a_bound(tmp_path):
    """kGuard (tic_math.h) is what makes the float32 fast path of the strip kernel 'identical for every input'.  Pinned here:
    (1) the table is at least the COMPLETE forward-error bound of tools/fastpath_error_bound.py - systematic error of the
        float32 AAN constants (exact maximum of a linear functional over the pixel box), float32 roundings of both passes,
        representation error of the quantiser multiplier, the reference's own float64 error - times its margin, for every
        coefficient, and not needlessly wider (the trip rate scales with it);
    (2) the systematic term is real: on the block that maximises it, the float-constant algorithm evaluated in float64
        differs from the true DCT by exactly that much;
    (3) the kernel's own arithmetic (tic_math.h compiled for the host: dct8_aan<float> columns then rows - the strip kernel's order -, quant_fma) stays
        inside kGuard against the exact-order float64 DCT on the adversarial blocks of tools/fastpath_error_search.py
        (tests/golden/adversarial_blocks.npz), extreme patterns and 2,000,000 random blocks of four kinds, in coefficient
        units and in quantised units at q = 1, 10, 50, 90, 99; and wherever the kernel's accept test (float thresholds of
        build_consts, float distance) accepts a rounding, it is the reference's."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "guard_selftest"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), os.path.join(root, "tests", "native", "guard_selftest.cpp")],
                   check=True)
    adv = np.load(os.path.join(root, "tests", "golden", "adversarial_blocks.npz"))["blocks"]
    assert adv.shape[1:] == (8, 8) and adv.dtype == np.uint8 and len(adv) >= 64
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import fastpath_error_bound as feb
    finally:
        sys.path.pop(0)
    sysb, worst = feb.systematic_bound(want_blocks=True)
    raw = tmp_path / "adv.bin"
    # the maximisers of the systematic term join the adversarial set; both sets were searched for the rows-first algorithm, the strip
    # kernel now runs columns first: the transposed blocks are ITS adversaries (both orientations are fed)
    raw.write_bytes(adv.tobytes() + worst.tobytes() + np.ascontiguousarray(adv.transpose(0, 2, 1)).tobytes()
                    + np.ascontiguousarray(worst.reshape(-1, 8, 8).transpose(0, 2, 1)).tobytes())
    r = subprocess.run([str(exe), str(raw), "2000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "guard_selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 accepted roundings differ" in r.stdout
    guard = np.array([float(x) for x in r.stdout.splitlines()[0].split()[1:]]).reshape(8, 8)
    # (1)
    need = feb.guard_matrix()
    assert (guard >= need).all(), (guard / need).min()
    assert (guard <= need * 1.001).all(), (guard / need).max()  # (the table is written with five significant digits, rounded up)
    t = feb.terms()
    assert (t["systematic"][:, 7] > 1.2e-4).all() and t["systematic"][7, 7] > 2e-4  # the term round 2's bound lacked is not small
    assert (feb.bound_matrix() >= t["systematic"] + t["rounding"] + t["multiplier"]).all()
    # (2) float64 evaluation of the float-constant algorithm (rows, then columns) against the true scaled DCT
    k = np.arange(8)
    T1 = np.where(k[:, None] == 0, 1.0, np.sqrt(2) * np.cos(k[:, None] * np.pi / 16) * np.sqrt(8) * 0.5 * np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16))

    class V:  # float64 values with dct8_aan's operation set
        def __init__(self, a): self.a = a
        def __add__(self, o): return V(self.a + o.a)
        def __sub__(self, o): return V(self.a - o.a)
        def mulc(self, c): return V(self.a * c)
        def fma(self, c, o): return V(self.a * c + o.a)

    for (u, v) in [(7, 7), (0, 7), (7, 0), (1, 1), (3, 5)]:
        x = worst[u, v].astype(np.float64) - 128.0
        rows = np.stack([o.a for o in feb.aan([V(x[:, n]) for n in range(8)], feb.C_F32)], 1)     # along the pixel rows
        z = np.stack([o.a for o in feb.aan([V(rows[r, :]) for r in range(8)], feb.C_F32)], 0)      # down the columns
        true = T1 @ x @ T1.T
        assert abs(abs(z[u, v] - true[u, v]) - sysb[u, v]) < 1e-9 * max(1.0, sysb[u, v]) + 1e-9, (u, v, z[u, v] - true[u, v], sysb[u, v])




def test_division_of_the_tie_path_is_the_ieee_division(tmp_path):
    """rational_quad / rational_slim of the strip kernel divide with tic_math.h div_rn since round 6 (five multiply-adds from the correctly
    rounded reciprocal instead of the compiler's ~13-instruction division): the same function compiled for the host gives the compiler's
    a / b bit for bit on every divisor of every quality (and 2,000 non-integral ones) times every tie point (k + 1/2) b with its neighbours
    within 8 ulps, every multiple of 1/8 and random numerators - 8 x 10^8 quotients - and build_consts hands it RN(1 / b)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = tmp_path / "div_selftest"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe), os.path.join(root, "tests", "native", "div_selftest.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith(": ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_strip_kernel_binary_keeps_its_landing_registers_private():
    """The production kernel's pixel loads land in v72..v79, registers the compiler may not allocate (amdgpu_num_vgpr(72)):
    a load in flight must never share a register with anything the compiler placed.  Checked on the SHIPPED binary by
    tinyimgcodec_amd/csrc/lint_strip_kernel.py - the same script csrc/Makefile runs behind the link, where a violation fails the
    build, for BOTH instantiations of dctq_strip_kernel (columns first / rows first): the only instructions that name v72..v79 are the
    hand-written loads into them, the hand-off out of them directly behind an s_waitcnt vmcnt (columns first: the ds_write_b64 to the
    byte-transpose buffer; rows first: the group of eight byte-to-float conversions) and plain moves out of them; no scratch, no
    accumulator registers, exactly 80 vector registers (six waves per SIMD)."""
    import importlib.util
    from tinyimgcodec_amd import _native as N
    if not os.path.exists(N.LIB_PATH):
        pytest.skip("library not built")
    spec = importlib.util.spec_from_file_location("lint_strip_kernel", os.path.join(ROOT, "tinyimgcodec_amd", "csrc", "lint_strip_kernel.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    try:
        summary = lint.check(N.LIB_PATH)
        assert "80 VGPRs" in lint.check(N.HOOKS_LIB_PATH)
    except lint.ToolsMissing as e:
        pytest.skip(str(e))
    assert "80 VGPRs" in summary


def test_rendezvous_file_is_created_exclusively_and_stale_files_are_ignored(tmp_path):
    """tic_comm_create's rendezvous (tic_rdv_publish / tic_rdv_wait, no GPU involved): rank 0 replaces whatever an earlier launch
    left under the name and never writes through a symlink; a reader ignores a complete file that is older than its own
    process (a crashed launch with the same MASTER_PORT / launcher pid) and takes the fresh one once it appears."""
    import ctypes as C
    import struct
    import threading
    import time
    from tinyimgcodec_amd import _native as N
    L = N.load()
    path = str(tmp_path / "rdv")
    ident = bytes(range(128))
    # a leftover of an earlier launch: complete, right size, right magic, but published long ago
    old = struct.pack("<8sQQ", b"TICRDV1\0", int((time.time() - 3600) * 1e9), 128) + bytes(128)
    with open(path, "wb") as f:
        f.write(old)
    buf = C.create_string_buffer(128)
    assert L.tic_rdv_wait(path.encode(), buf, 128, 300, 0) == N.TIC_E_ARG
    assert b"stale" in L.tic_comm_last_error(None)
    # rank 0 publishes 0.3 s after the reader started to poll: the reader must return the new payload, not the leftover
    t = threading.Timer(0.3, lambda: L.tic_rdv_publish(path.encode(), ident, 128))
    t.start()
    assert L.tic_rdv_wait(path.encode(), buf, 128, 5000, 0) == N.TIC_OK
    t.join()
    assert buf.raw == ident
    assert oct(os.stat(path).st_mode & 0o777) == "0o600"
    # never through a symlink: the temporary name is removed first, then created with O_EXCL|O_NOFOLLOW
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.unlink(path)
    os.symlink(str(victim), path + ".tmp")
    assert L.tic_rdv_publish(path.encode(), ident, 128) == N.TIC_OK
    assert victim.read_bytes() == b"precious" and not os.path.islink(path)
    # a file of the wrong size or magic is not a rendezvous file
    with open(path, "wb") as f:
        f.write(b"x" * 152)
    assert L.tic_rdv_wait(path.encode(), buf, 128, 100, 0) == N.TIC_E_ARG
    assert L.tic_rdv_wait(path.encode(), buf, 128, 100, 1) == N.TIC_E_ARG  # (an explicit not-before time does not help a bad file)
    # communicators of one job get distinct names
    from tinyimgcodec_amd import distributed as D
    assert D.default_rendezvous_path() != D.default_rendezvous_path()
