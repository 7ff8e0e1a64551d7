"""Deterministic synthetic inputs and weights shared by the golden generator, the tests,
smoke() and bench.py.  Pure numpy / torch-CPU generators so that this
container and the GPU box produce identical bytes.  Our own code (not from the reference).
"""
import zlib

import numpy as np
import torch

IMG_H, IMG_W = 480, 640


def synth_objects(seed, G, H=IMG_H, W=IMG_W, tiny_visible=False):
    """G boxes (int-valued float32 xyxy), labels in [0,21), visible masks u8[G,H,W]:
    ellipse inscribed in the box; odd-indexed objects lose their left half (occlusion).
    tiny_visible: object 0 keeps only a 3x3-pixel visible patch at its box centre region."""
    rng = np.random.RandomState(seed)
    boxes = np.zeros((G, 4), np.float32)
    masks = np.zeros((G, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for g in range(G):
        w = rng.randint(30, 300)
        h = rng.randint(30, 300)
        x = rng.randint(0, W - w)
        y = rng.randint(0, H - h)
        boxes[g] = (x, y, x + w, y + h)
        cx, cy = x + w / 2, y + h / 2
        m = (((xx - cx) / (w / 2)) ** 2 + ((yy - cy) / (h / 2)) ** 2 <= 1)
        if g % 2:
            m[:, :int(cx)] = False
        masks[g] = m
    labels = rng.randint(0, 21, G).astype(np.int64)
    if tiny_visible and G > 0:
        x1, y1, x2, y2 = boxes[0].astype(int)
        masks[0] = 0
        # a small visible patch that covers only a couple of stride-8 cell centres
        px, py = ((x1 + x2) // 2) // 8 * 8, ((y1 + y2) // 2) // 8 * 8
        masks[0, py:py + 9, px:px + 17] = 1
    return boxes, labels, masks


def synth_images(seed, B, H=IMG_H, W=IMG_W):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, H, W, generator=g)


def img_metas(B, H=IMG_H, W=IMG_W):
    return [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3),
                 scale_factor=np.ones(4, np.float32), flip=False) for _ in range(B)]


def _name_seed(seed, name):
    return (seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1)


@torch.no_grad()
def fill_state_dict(sd, seed=0):
    """Overwrite every tensor of a (reference-named) state dict with seeded, non-trivial
    values, keyed by NAME (so module construction order does not matter):
      conv / linear weights : N(0, sqrt(2/fan_out))          (fan_out = Cout*kh*kw)
      predictor convs (atss_*): N(0, 0.02), atss_cls bias -2.0 (so that some scores pass 0.05)
      biases                : N(0, 0.05)
      BN weight             : U(0.5, 1.5)   (norm3 / downsample BN: U(0.15, 0.45) to tame residual growth)
      BN bias, running_mean : N(0, 0.1);  running_var: U(0.5, 1.5)
      GN weight             : U(0.5, 1.5);  GN bias: N(0, 0.1)
      Scale.scale           : U(0.8, 1.2)
    """
    for name in sorted(sd.keys()):
        t = sd[name]
        g = torch.Generator().manual_seed(_name_seed(seed, name))
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue

        def normal(std, mean=0.0):
            t.copy_(torch.randn(t.shape, generator=g) * std + mean)

        def uniform(lo, hi):
            t.copy_(torch.rand(t.shape, generator=g) * (hi - lo) + lo)

        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
        is_bn = parent.startswith("bn") or (parent.isdigit() and "downsample" in name and t.dim() == 1)
        is_gn = parent == "gn"
        if leaf == "scale":
            uniform(0.8, 1.2)
        elif is_bn:
            if leaf == "weight":
                if parent == "bn3" or "downsample" in name:
                    uniform(0.15, 0.45)
                else:
                    uniform(0.5, 1.5)
            elif leaf in ("bias", "running_mean"):
                normal(0.1)
            elif leaf == "running_var":
                uniform(0.5, 1.5)
        elif is_gn:
            if leaf == "weight":
                uniform(0.5, 1.5)
            else:
                normal(0.1)
        elif leaf == "weight" and t.dim() == 4:
            if ".atss_" in name or name.startswith("atss_"):
                normal(0.02)
            else:
                fan_out = t.shape[0] * t.shape[2] * t.shape[3]
                normal((2.0 / fan_out) ** 0.5)
        elif leaf == "bias":
            if "atss_cls" in name:
                normal(0.05, mean=-2.0)
            else:
                normal(0.05)
        else:
            raise KeyError(f"unclassified tensor {name} {tuple(t.shape)}")
    return sd


def graded_maps(masks):
    """float32 [G, H, W] sampling maps with values in {0} u [0.25, 1]: the visible masks times a fixed spatial pattern -- inputs
    of the assigner tests that need NON-binary map values (multiply_samplepro_for_weight; the mask-free sampler's maps)."""
    m = np.asarray(masks)
    H, W = m.shape[-2:]
    yy, xx = np.mgrid[0:H, 0:W]
    pat = (0.25 + 0.75 * (((xx * 7 + yy * 13) % 32) / 31.0)).astype(np.float32)
    return (m.astype(np.float32) * pat[None]).astype(np.float32)
