"""numpy twin of libtamtr_host.so's 8-bit image kernels (include/tamtr_host.h).  TEST INFRASTRUCTURE: imported by tests/ only.

Same arithmetic as tam-tr_amd/csrc/host/imgproc.c, written array-at-a-time: OpenCV's published 8-bit algorithms for
cv2.resize(INTER_LINEAR) (ultralytics/data/base.py:156-164), cv2.warpAffine (data/augment.py:415-420) and the
BGR2HSV -> LUT -> HSV2BGR chain of RandomHSV (data/augment.py:590-609).  Parity with cv2 itself is UNPINNED: cv2 is not in the
image and the reference holds no vectors for these calls; the closed-form tests in tests/test_data.py are what anchors them.
"""
import numpy as np


def _rint(x):
    return np.rint(x).astype(np.int64)


def _linear_taps(sn, dn):
    scale = sn / dn
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return f - s.astype(np.float32), s


def resize_linear_u8(src, dw, dh):
    """OpenCV's INTER_LINEAR for 8-bit images: pixel-centre mapping, 11-bit fixed-point taps, horizontal then vertical pass;
    an exact 2x decimation takes the 2x2 mean path.  src [h, w, c] uint8 -> [dh, dw, c] uint8."""
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    fx, sx = _linear_taps(sw, dw)
    low, high = sx < 0, sx >= sw - 1
    fx = np.where(low | high, np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    a0, a1 = _rint((1 - fx) * np.float32(2048)), _rint(fx * np.float32(2048))
    fy, sy = _linear_taps(sh, dh)
    b0, b1 = _rint((1 - fy) * np.float32(2048)), _rint(fy * np.float32(2048))
    r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int64)
    rows = np.unique(np.concatenate([r0, r1]))
    lut = np.zeros(sh, np.int64)
    lut[rows] = np.arange(len(rows))
    hp = s[rows][:, sx] * a0[None, :, None] + s[rows][:, np.minimum(sx + 1, sw - 1)] * a1[None, :, None]
    top, bot = hp[lut[r0]], hp[lut[r1]]
    out = (((b0[:, None, None] * (top >> 4)) >> 16) + ((b1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


_WARP_TAB = None


def _warp_table():
    """[32*32, 4] int weights (sum 32768) for the 1/32-pixel bilinear remap."""
    global _WARP_TAB
    if _WARP_TAB is None:
        t = np.arange(32, dtype=np.float32) / np.float32(32)
        wy, wx = np.stack([1 - t, t], 1), np.stack([1 - t, t], 1)
        tab = (wy[:, None, :, None] * wx[None, :, None, :]).reshape(32 * 32, 4)
        it = np.minimum(_rint(tab * np.float32(32768)), 32767)
        it[np.arange(len(it)), it.argmax(1)] += 32768 - it.sum(1)
        _WARP_TAB = it
    return _WARP_TAB


def warp_affine_u8(src, M, dw, dh, border=114):
    """OpenCV's warpAffine (bilinear, constant border) for 8-bit images: M [2, 3] maps source -> destination; the inverse map is
    evaluated in 10-bit fixed point, positions quantised to 1/32 pixel, taps blended with 15-bit weights."""
    M = np.asarray(M, np.float64)[:2]
    det = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    det = 1.0 / det if det != 0 else 0.0
    m0, m1, m3, m4 = M[1, 1] * det, -M[0, 1] * det, -M[1, 0] * det, M[0, 0] * det
    b1, b2 = -m0 * M[0, 2] - m1 * M[1, 2], -m3 * M[0, 2] - m4 * M[1, 2]
    x, y = np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64)
    X = (_rint((m1 * y + b1) * 1024) + 16)[:, None] + _rint(m0 * x * 1024)[None]
    Y = (_rint((m4 * y + b2) * 1024) + 16)[:, None] + _rint(m3 * x * 1024)[None]
    X, Y = X >> 5, Y >> 5
    sx, sy, alpha = X >> 5, Y >> 5, (Y & 31) * 32 + (X & 31)
    sh, sw = src.shape[:2]
    wts = _warp_table()[alpha]                                                  # [dh, dw, 4]
    acc = np.zeros((dh, dw, src.shape[2]), np.int64)
    for k, (oy, ox) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + oy, sx + ox
        inside = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        tap = np.where(inside[..., None], src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)].astype(np.int64), border)
        acc += tap * wts[..., k, None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def rgb_to_hsv_u8(img):
    """OpenCV's 8-bit RGB->HSV (H in [0, 180)): integer arithmetic with 12-bit reciprocal tables.  -> three uint8 planes."""
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.concatenate([[0], _rint((255 << 12) / i)])
    hdiv = np.concatenate([[0], _rint((180 << 12) / (6.0 * i))])
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    diff = v - np.minimum(np.minimum(r, g), b)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return h.astype(np.uint8), s.astype(np.uint8), v.astype(np.uint8)


def hsv_to_rgb_u8(h, s, v):
    """OpenCV's 8-bit HSV->RGB: float sextant formula, result * 255 rounded and saturated."""
    hf = h.astype(np.float32) * np.float32(6.0 / 180.0)
    sf, vf = s.astype(np.float32) * np.float32(1 / 255.0), v.astype(np.float32) * np.float32(1 / 255.0)
    hf = np.where(hf >= 6, hf - 6, hf)
    sector = np.floor(hf).astype(np.int64)
    frac = hf - sector.astype(np.float32)
    frac = np.where(sector >= 6, np.float32(0), frac)
    sector = np.where(sector >= 6, 0, sector)
    tab = np.stack([vf, vf * (1 - sf), vf * (1 - sf * frac), vf * (1 - sf * (1 - frac))], -1)
    order = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sector]     # (b, g, r) picks per sextant
    bgr = np.take_along_axis(tab, order, -1)
    bgr = np.where((s == 0)[..., None], vf[..., None], bgr)
    return np.clip(np.rint(bgr[..., ::-1] * np.float32(255)), 0, 255).astype(np.uint8)


def hsv_lut_u8(img, lut_h, lut_s, lut_v):
    """RGB -> HSV, one look-up table per plane, HSV -> RGB."""
    h, s, v = rgb_to_hsv_u8(img)
    return hsv_to_rgb_u8(lut_h[h], lut_s[s], lut_v[v])
