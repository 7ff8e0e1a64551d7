"""ORACLE (test infrastructure, never imported by radet_amd/): NumPy restatement of the instance-mask transforms of
the reference's training pipeline.

Follows radet/core/mask/structures.py:253-303 (BitmapMasks.rescale / resize / flip / pad -> mmcv.imrescale /
imresize / imflip / impad per mask) and radet/datasets/pipelines/loading.py:419-422 (mask / mask.max()).
flip = np.flip, pad = right/bottom constant pad (mmcv.impad(shape=...)), both pinned by NumPy itself.
resize: mmcv's 'nearest' is cv2.INTER_NEAREST.  cv2 is NOT installed here and the reference holds no mask fixtures,
so this one rule is "parity unpinned": it restates OpenCV's published resizeNN,
    inv_scale = dsize / ssize;  src = min(floor(dst * (1. / inv_scale)), ssize - 1)      (double arithmetic)."""
import numpy as np


def rescale_size(old_size, scale):
    w, h = old_size
    if isinstance(scale, (float, int)):
        factor = scale
    else:
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def _nn_index(dst_n, src_n):
    inv = 1.0 / (np.float64(dst_n) / np.float64(src_n))
    return np.minimum(np.floor(np.arange(dst_n, dtype=np.float64) * inv).astype(np.int64), src_n - 1)


def resize_nearest(masks, out_hw):
    masks = np.asarray(masks)
    Hr, Wr = out_hw
    if masks.shape[0] == 0:
        return np.empty((0, Hr, Wr), masks.dtype)
    return masks[:, _nn_index(Hr, masks.shape[1])][:, :, _nn_index(Wr, masks.shape[2])]


def flip(masks, direction="horizontal"):
    axis = {"horizontal": 2, "vertical": 1, "diagonal": (1, 2)}[direction]
    return np.flip(np.asarray(masks), axis=axis)


def pad(masks, out_hw, pad_val=0):
    masks = np.asarray(masks)
    out = np.full((masks.shape[0], out_hw[0], out_hw[1]), pad_val, masks.dtype)
    out[:, :masks.shape[1], :masks.shape[2]] = masks
    return out


def normalize(masks):
    out = []
    for m in np.asarray(masks):
        with np.errstate(divide="ignore", invalid="ignore"):
            q = m / m.max()
        out.append(np.nan_to_num(q, nan=0.0).astype(m.dtype))
    return np.stack(out) if out else np.asarray(masks)


def transform(masks, resized_hw=None, flip_dir=None, out_hw=None, pad_val=0, norm=False):
    m = np.asarray(masks)
    if norm:
        m = normalize(m)
    if resized_hw is not None:
        m = resize_nearest(m, resized_hw)
    if flip_dir:
        m = flip(m, flip_dir)
    if out_hw is not None:
        m = pad(m, out_hw, pad_val)
    return np.ascontiguousarray(m)
