"""numpy/scipy restatement of the reference's transform stage, encode() (tinyimgcodec/codec.py:26-43 with utils.py:13-20,
32-37, 48-61 and constants.py:9-35) - the same library stack the reference runs on, written from its description, not its files.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (oracle/): bench.py times it as the "pure-Python-stack" figure of SURVEY.md section
8d(ii) where scipy is importable, and tests/test_oracle_golden.py checks it against the C oracle.  Because scipy's pocketfft IS
the reference's arithmetic, this function is bit-identical to the reference wherever the same scipy runs."""
import numpy as np

Q_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112,
                   100, 103, 99], dtype=np.int64).reshape(8, 8)  # JPEG Annex K luminance table (constants.py:9-20)


def _zigzag_order():
    """Natural index u*8+v of scan position k (constants.py:23-34): anti-diagonals, direction alternating."""
    order = []
    for s in range(15):
        diag = [(u, s - u) for u in range(8) if 0 <= s - u < 8]
        order += diag if s % 2 else diag[::-1]
    return np.array([u * 8 + v for u, v in order])


ZIGZAG = _zigzag_order()


def encode(image, quality=50):
    """-> dict(height, width, quality, dc int32[N] (DPCM'd), ac int32[N, 63]) as codec.py:26-43."""
    from scipy.fftpack import dct

    image = np.asarray(image)
    h, w = image.shape
    ph, pw = (-h) % 8, (-w) % 8
    if ph or pw:
        image = np.pad(image, ((0, ph), (0, pw)), "reflect")               # utils.py:56-61
    x = image.astype(np.int32) - 128                                       # codec.py:29
    bh, bw = x.shape[0] // 8, x.shape[1] // 8
    blocks = x.reshape(bh, 8, bw, 8).swapaxes(1, 2)                        # utils.py:13-20
    coef = dct(dct(blocks, norm="ortho", axis=-2), norm="ortho", axis=-1)  # utils.py:32-37
    factor = 5000 / quality if quality < 50 else 200 - 2 * quality        # utils.py:50-51
    q = np.round(coef / (Q_LUMA * factor / 100)).astype(np.int32)         # utils.py:52-53
    zz = q.reshape(bh, bw, 64)[:, :, ZIGZAG]                               # codec.py:32-33
    dc = zz[:, :, 0].reshape(-1).copy()
    dc[1:] = np.diff(dc)                                                   # codec.py:34-35
    return {"height": h, "width": w, "quality": quality, "dc": dc, "ac": zz[:, :, 1:].reshape(-1, 63)}
