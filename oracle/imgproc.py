"""TEST INFRASTRUCTURE (CPU oracle) -- NumPy restatement of the OpenCV operations the mask-free sampler wraps around the
MBD / GDT transforms (radet/ops/bbox2distance/bbox2distance_wrapper.py:80-93, 118-130, 170-181): cv2.resize (INTER_LINEAR),
cv2.GaussianBlur 9x9 (sigma 0), and GDT_box2distance.sobel_extract_edge (GaussianBlur 3x3 -> RGB2GRAY -> Sobel x / y ->
addWeighted -> abs -> / max).

cv2 is not installed here and its source is not part of /root/reference (pip dependency opencv-python, unpinned in
requirements.txt): PARITY UNPINNED against cv2.  The formulas follow OpenCV's generic C++ implementation (imgproc
resize.cpp / smooth / color / deriv) as documented; tests cross-check them against scipy.ndimage and analytic cases."""
import numpy as np


def _lin_coords(dst_n, src_n, clamp_frac):
    scale = 1.0 / (np.float64(dst_n) / np.float64(src_n))
    f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= src_n - 1
        f[hi], s[hi] = 0, src_n - 1
    return s, f


def _fix(c):
    return np.clip(np.rint(c.astype(np.float32) * np.float32(2048.0)), -32768, 32767).astype(np.int64)


def resize_linear_u8(img, dsize):
    """cv2.resize(img, (dw, dh)) for uint8 [h, w(, c)], INTER_LINEAR, 11-bit fixed point"""
    dw, dh = int(dsize[0]), int(dsize[1])
    a = img.reshape(img.shape[0], img.shape[1], -1).astype(np.int64)
    sh, sw = a.shape[:2]
    sx, fx = _lin_coords(dw, sw, True)
    sy, fy = _lin_coords(dh, sh, False)
    a0, a1 = _fix(np.float32(1) - fx), _fix(fx)
    b0, b1 = _fix(np.float32(1) - fy), _fix(fy)
    x1 = np.minimum(sx + 1, sw - 1)
    hor = a[:, sx, :] * a0[None, :, None] + a[:, x1, :] * a1[None, :, None]          # [sh, dw, c]
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    v = (((b0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8).reshape((dh, dw) + img.shape[2:])


def resize_linear_float(img, dsize):
    """cv2.resize for float32 / float64 [h, w]: float coefficients, sums in the element type"""
    dw, dh = int(dsize[0]), int(dsize[1])
    T = img.dtype.type
    sh, sw = img.shape
    sx, fx = _lin_coords(dw, sw, True)
    sy, fy = _lin_coords(dh, sh, False)
    a0, a1 = (np.float32(1) - fx).astype(T), fx.astype(T)
    b0, b1 = (np.float32(1) - fy).astype(T), fy.astype(T)
    x1 = np.minimum(sx + 1, sw - 1)
    hor = img[:, sx] * a0[None, :] + img[:, x1] * a1[None, :]
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    return (hor[y0] * b0[:, None] + hor[y1] * b1[:, None]).astype(T)


def gauss9_taps():
    sigma = 0.3 * ((9 - 1) * 0.5 - 1) + 0.8
    x = np.arange(9, dtype=np.float64) - 4.0
    cf = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    s = 0.0
    for v in cf:
        s += float(v)
    return (cf.astype(np.float64) * (1.0 / s)).astype(np.float32)


def _reflect101(idx, n):
    idx = np.asarray(idx).copy()
    if n == 1:
        return np.zeros_like(idx)
    while ((idx < 0) | (idx >= n)).any():
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
    return idx


def gaussian_blur9_u8(img):
    """cv2.GaussianBlur(img, (9, 9), 0, borderType=BORDER_DEFAULT) for uint8 [h, w, 3]: float taps, symmetric passes"""
    k = gauss9_taps()
    h, w = img.shape[:2]
    a = img.astype(np.float32)
    xs = np.arange(w)
    row = k[4] * a
    for i in range(1, 5):
        row = row + k[4 + i] * (a[:, _reflect101(xs + i, w)] + a[:, _reflect101(xs - i, w)])
    ys = np.arange(h)
    col = k[4] * row
    for i in range(1, 5):
        col = col + k[4 + i] * (row[_reflect101(ys + i, h)] + row[_reflect101(ys - i, h)])
    return np.clip(np.rint(col), 0, 255).astype(np.uint8)


def sobel_edge(img):
    """GDT_box2distance.sobel_extract_edge (bbox2distance_wrapper.py:118-130) for uint8 [h, w, 3] -> float32 [h, w]"""
    h, w = img.shape[:2]
    a = img.astype(np.int64)
    ys, xs = np.arange(h), np.arange(w)
    acc = np.zeros_like(a)
    for j, wy in ((-1, 1), (0, 2), (1, 1)):
        for i, wx in ((-1, 1), (0, 2), (1, 1)):
            acc += wy * wx * a[_reflect101(ys + j, h)][:, _reflect101(xs + i, w)]
    b = (acc + 8) >> 4
    gray = ((b[..., 0] * 4899 + b[..., 1] * 9617 + b[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.float32)

    def at(j, i):
        return gray[_reflect101(ys + j, h)][:, _reflect101(xs + i, w)]

    gx = (at(-1, 1) - at(-1, -1)) + np.float32(2) * (at(0, 1) - at(0, -1)) + (at(1, 1) - at(1, -1))
    gy = (at(1, -1) - at(-1, -1)) + np.float32(2) * (at(1, 0) - at(-1, 0)) + (at(1, 1) - at(-1, 1))
    e = np.abs(gx * np.float32(0.5) + gy * np.float32(0.5)).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (e / e.max()).astype(np.float32)
