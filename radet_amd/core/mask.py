"""Device-resident instance masks with the reference's `BitmapMasks` interface
(radet/core/mask/structures.py:215-303, 473-480): the masks of one image live in HBM as u8[G,H,W] and every
geometric transform of the training pipeline (Resize -> RandomFlip -> Pad, transforms.py) is one HIP pass, so the
label assigner (`LabelAssignment.assign_batch`) consumes them without a host round trip.

Resize semantics: `mmcv.imresize(..., interpolation='nearest')` = cv2.INTER_NEAREST, whose published rule is
src = min(floor(dst * (1 / (dst_size / src_size))), src_size - 1); cv2 is not installed in the build image, so
this one rule is restated, not pinned (oracle/masks.py says the same)."""
import numpy as np
import torch

from .. import kernels as K


def rescale_size(old_size, scale):
    """mmcv.rescale_size: (w, h), scale = float factor or (long_edge, short_edge) bound -> new (w, h)"""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f"Invalid scale {scale}, must be positive.")
        factor = scale
    else:
        max_long, max_short = max(scale), min(scale)
        factor = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


class BitmapMasks:
    def __init__(self, masks, height, width, device=None):
        self.height, self.width = int(height), int(width)
        dev = device or torch.device("cuda", torch.cuda.current_device())
        if isinstance(masks, torch.Tensor):
            m = masks.to(device=dev, dtype=torch.uint8)
        elif len(masks) == 0:
            m = torch.empty(0, self.height, self.width, dtype=torch.uint8, device=dev)
        else:
            arr = np.stack(masks) if isinstance(masks, (list, tuple)) else np.asarray(masks)
            assert arr.ndim == 3 and arr.shape[1:] == (self.height, self.width), arr.shape
            m = torch.from_numpy(np.ascontiguousarray(arr.astype(np.uint8))).to(dev)
        self.masks = m.reshape(-1, self.height, self.width).contiguous()

    def __len__(self):
        return self.masks.shape[0]

    def __repr__(self):
        return f"BitmapMasks(num_masks={len(self)}, height={self.height}, width={self.width})"

    # --- the reference's per-op interface -------------------------------------------------------------------------
    def transform(self, resized_hw=None, flip=None, out_hw=None, pad_val=0, normalize=False):
        """Resize -> flip -> pad (-> loader normalisation) fused into one pass"""
        Hr, Wr = resized_hw or (self.height, self.width)
        Hd, Wd = out_hw or (Hr, Wr)
        out = K.mask_transform(self.masks, (Hd, Wd), (Hr, Wr), flip, pad_val, normalize)
        return BitmapMasks(out, Hd, Wd, device=out.device)

    def rescale(self, scale, interpolation="nearest"):
        assert interpolation == "nearest"
        new_w, new_h = rescale_size((self.width, self.height), scale)
        return self.transform(resized_hw=(new_h, new_w))

    def resize(self, out_shape, interpolation="nearest"):
        assert interpolation == "nearest"
        return self.transform(resized_hw=(int(out_shape[0]), int(out_shape[1])))

    def flip(self, flip_direction="horizontal"):
        assert flip_direction in ("horizontal", "vertical", "diagonal")
        return self.transform(flip=flip_direction)

    def pad(self, out_shape, pad_val=0):
        return self.transform(out_hw=(int(out_shape[0]), int(out_shape[1])), pad_val=pad_val)

    def normalized(self):
        """LoadAnnotations._load_bop_masks: (mask / mask.max()).astype(dtype), per mask (loading.py:419-422)"""
        return self.transform(normalize=True)

    @property
    def areas(self):
        return self.masks.sum((1, 2), dtype=torch.int64).cpu().numpy()

    def to_ndarray(self):
        return self.masks.cpu().numpy()

    def to_tensor(self, dtype, device):
        return self.masks.to(dtype=dtype, device=device)
