"""GPU instance-mask path: fused nearest-resize / flip / pad / normalise of u8 mask stacks through
the C ABI, bit-exact against the NumPy oracle (oracle/masks.py), and the masks -> assigner hand-over on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _masks(rng, G, H, W, vals=(0, 255)):
    m = np.zeros((G, H, W), np.uint8)
    for g in range(G):
        w, h = rng.randint(4, W), rng.randint(4, H)
        x, y = rng.randint(0, W - w + 1), rng.randint(0, H - h + 1)
        yy, xx = np.mgrid[0:H, 0:W]
        e = (((xx - (x + w / 2)) / (w / 2)) ** 2 + ((yy - (y + h / 2)) / (h / 2)) ** 2) <= 1
        m[g][e] = vals[1]
        m[g][rng.rand(H, W) < 0.02] = rng.randint(0, 256)        # speckle: exercises every byte value
    return m


CASES = [
    # G, Hs, Ws, resized (Hr, Wr), flip, out (Hd, Wd), pad_val, normalise
    (3, 480, 640, None, None, None, 0, False),
    (3, 480, 640, (480, 640), "horizontal", (480, 640), 0, True),
    (5, 480, 640, (600, 800), "vertical", (608, 800), 0, True),
    (2, 480, 640, (333, 517), "diagonal", (352, 544), 7, False),     # odd sizes, Wd % 4 == 0
    (2, 97, 131, (200, 263), "horizontal", (201, 263), 255, True),   # Wd % 4 != 0 -> byte stores
    (1, 64, 64, (7, 5), None, (8, 8), 0, False),                     # strong down-scale
    (4, 33, 47, (33, 47), None, (64, 64), 1, False),                 # pad only
]


@pytest.mark.parametrize("case", CASES)
def test_mask_transform_bit_exact(case):
    from oracle import masks as om
    from radet_amd import kernels as K
    G, Hs, Ws, rhw, flip, ohw, pv, norm = case
    rng = np.random.RandomState(G * 1000 + Hs + Ws)
    m = _masks(rng, G, Hs, Ws)
    if norm:
        m[0] = 0                                             # an all-zero mask: 0 / 0 -> 0
    got = K.mask_transform(torch.from_numpy(m).cuda(), ohw, rhw, flip, pv, norm).cpu().numpy()
    ref = om.transform(m, rhw, flip, ohw, pv, norm)
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert (got == ref).all(), np.argwhere(got != ref)[:5]


def test_bitmapmasks_api_and_assigner_handover():
    """Resize(keep_ratio) -> RandomFlip -> Pad on the device, then LabelAssignment straight from the device masks ==
    oracle transforms + oracle assigner on the host (bit-exact indices and weights, same RNG consumption)."""
    from oracle import assigner as oa, masks as om
    from radet_amd.core import BitmapMasks, rescale_size
    from radet_amd.datasets import LabelAssignment
    rng = np.random.RandomState(3)
    H0, W0, G = 480, 640, 4
    m = _masks(rng, G, H0, W0)
    bm = BitmapMasks(list(m), H0, W0)
    assert len(bm) == G and (bm.to_ndarray() == m).all() and (bm.areas == m.astype(np.int64).sum((1, 2))).all()
    scale = (400, 300)                                       # (long, short) bound like Resize(img_scale=..., keep_ratio)
    new_w, new_h = rescale_size((W0, H0), scale)
    assert (new_w, new_h) == om.rescale_size((W0, H0), scale)
    pad_hw = ((new_h + 31) // 32 * 32, (new_w + 31) // 32 * 32)
    step = bm.normalized().rescale(scale).flip("horizontal").pad(pad_hw)
    fused = bm.transform(resized_hw=(new_h, new_w), flip="horizontal", out_hw=pad_hw, normalize=True)
    ref = om.transform(m, (new_h, new_w), "horizontal", pad_hw, 0, True)
    assert (step.to_ndarray() == ref).all() and (fused.to_ndarray() == ref).all()
    assert (step.height, step.width) == pad_hw
    # boxes of the transformed masks
    boxes = np.zeros((G, 4), np.float32)
    for g in range(G):
        ys, xs = np.nonzero(ref[g])
        boxes[g] = (xs.min(), ys.min(), xs.max() + 1, ys.max() + 1)
    labels = rng.randint(0, 21, G).astype(np.int64)
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
    r1, r2 = np.random.RandomState(11), np.random.RandomState(11)
    p2g, pw = la.assign_batch([boxes], [fused.masks], (pad_hw[0], pad_hw[1], 3), rngs=[r1])
    a, w = oa.assign_points(boxes, labels, ref, (pad_hw[0], pad_hw[1], 3), rng=r2)
    assert (p2g[0].cpu().numpy() == a).all() and (pw[0].cpu().numpy() == w).all()
    assert r1.random_sample() == r2.random_sample()
    # empty stack
    e = BitmapMasks([], H0, W0).rescale(scale).flip().pad(pad_hw)
    assert len(e) == 0 and e.to_ndarray().shape == (0,) + pad_hw
