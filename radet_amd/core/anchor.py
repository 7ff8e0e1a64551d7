"""AnchorGenerator (core/anchor/anchor_generator.py:58-329 of the reference) for the single-square-anchor
configuration of the RADet head: one anchor per cell, side = octave_base_scale * stride, centred at
(j*stride, i*stride) (center_offset = 0).  Grids are produced by the radet_grid_anchors HIP kernel
and cached per (sizes, device)."""
import numpy as np
import torch

from .. import kernels as K
from ..utils import Registry, build_from_cfg

ANCHOR_GENERATORS = Registry("Anchor generator")


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


@ANCHOR_GENERATORS.register_module()
class AnchorGenerator:
    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.0):
        if list(ratios) != [1.0] or scales is not None or base_sizes is not None or centers is not None \
                or center_offset != 0.0 or scales_per_octave != 1 or octave_base_scale is None:
            raise NotImplementedError("AnchorGenerator: only ratios=[1.0], scales_per_octave=1, center_offset=0 "
                                      "(the RADet head configuration) is implemented")
        self.strides = [(s, s) if isinstance(s, int) else tuple(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides]
        self.octave_base_scale = int(octave_base_scale)
        self.scales = torch.tensor([float(octave_base_scale)])
        self.ratios = torch.tensor(ratios)
        self.center_offset = center_offset
        self._cache = {}

    @property
    def num_base_anchors(self):
        return [1 for _ in self.strides]

    @property
    def num_levels(self):
        return len(self.strides)

    def grid_anchors(self, featmap_sizes, device="cuda"):
        assert self.num_levels == len(featmap_sizes)
        device = torch.device(device)
        key = (tuple((int(h), int(w)) for h, w in featmap_sizes), str(device))
        if key not in self._cache:
            if device.type != "cuda":
                # host-side grid for data-pipeline callers (pure index arithmetic, exact integers)
                outs = []
                for (h, w), (s, _) in zip(key[0], self.strides):
                    jj, ii = np.meshgrid(np.arange(w), np.arange(h))
                    c = np.stack([jj.reshape(-1) * s, ii.reshape(-1) * s], 1).astype(np.float32)
                    half = 0.5 * self.octave_base_scale * s
                    outs.append(torch.from_numpy(np.concatenate([c - half, c + half], 1)))
                self._cache[key] = outs
            else:
                lv = K.Levels(key[0], 1)
                d, n = K.level_desc(lv, [s[0] for s in self.strides])
                flat = torch.empty(lv.rows, 4, device=device)
                K.grid_anchors(flat, d, n, self.octave_base_scale)
                self._cache[key] = [flat[lv.offsets[i]:lv.offsets[i] + h * w] for i, (h, w) in enumerate(key[0])]
        return list(self._cache[key])

    def valid_flags(self, featmap_sizes, pad_shape, device="cuda"):
        """Computed-but-unused by RADetHead.loss in the reference; kept for API parity."""
        flags = []
        for (fh, fw), (sw, sh) in zip(featmap_sizes, self.strides):
            h, w = pad_shape[:2]
            vh, vw = min(int(np.ceil(h / sh)), fh), min(int(np.ceil(w / sw)), fw)
            vx = np.zeros(fw, bool)
            vy = np.zeros(fh, bool)
            vx[:vw] = True
            vy[:vh] = True
            flags.append(torch.from_numpy((vy[:, None] & vx[None, :]).reshape(-1)).to(device))
        return flags
