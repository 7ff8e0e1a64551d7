"""MBD / GDT box-to-distance transforms on the GPU: skewed-wavefront kernels, bit-identical to the
oracle (oracle/dist.c, itself pinned to the reference's bbox2distance_ext.cpp compiled in place)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [(150, 200), (37, 53), (64, 41), (2, 2), (90, 90), (151, 233), (3, 300)]


def _crops(rs):
    out = []
    for h, w in SIZES:
        img = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        img[h // 4:3 * h // 4, w // 4:3 * w // 4] //= 3
        out.append(img)
    return out


@pytest.mark.parametrize("niter,base", [(4, 300), (3, 40), (1, 300)])
def test_mbd_batch_bit_exact(niter, base):
    from oracle import dist
    from radet_amd import ops
    rs = np.random.RandomState(niter)
    imgs = _crops(rs)
    seeds = [dist.border_seeds(*i.shape[:2]) for i in imgs]
    got = ops.mbd_batch(imgs, seeds, 0.1, niter, base)
    for i, s, g in zip(imgs, seeds, got):
        ref = dist.mbd(i, s[0], s[1], 0.1, niter, base)
        assert g.dtype == torch.float64 and np.array_equal(g.cpu().numpy(), ref), i.shape


def test_gdt_batch_bit_exact():
    from oracle import dist
    from radet_amd import ops
    rs = np.random.RandomState(7)
    costs = [rs.rand(h, w).astype(np.float32) for h, w in SIZES]
    seeds = [dist.border_seeds(h, w) for h, w in SIZES]
    got = ops.gdt_batch(costs, seeds)
    for c, s, g in zip(costs, seeds, got):
        assert np.array_equal(g.cpu().numpy(), dist.gdt(c, s[0], s[1])), c.shape


def test_box2distance_classes():
    """reference wrapper semantics (bbox2distance_wrapper.py:58-95,155-185): disabled crops -> ones, crop to the
    un-padded region, seeds on the border every `interval` pixels."""
    from oracle import dist
    from radet_amd import ops
    rs = np.random.RandomState(1)
    imgs = _crops(rs)[:3]
    xy = [(2, 3, 100, 90), (0, 0, 53, 37), (5, 1, 30, 60)]
    m = ops.MBD_box2distance(mode="mean", alpha=0.1, niter=4, base_size=300, interval=3)
    out = m(imgs, [True, False, True], xy)
    for i, (img, e, b, o) in enumerate(zip(imgs, [True, False, True], xy, out)):
        if e:
            ref = dist.mbd(img, *dist.border_seeds(*img.shape[:2]), 0.1, 4, 300)[b[1]:b[3], b[0]:b[2]]
            assert np.array_equal(o.cpu().numpy(), ref)
        else:
            assert o.shape == (b[3] - b[1], b[2] - b[0]) and bool((o == 1).all())
    sx, sy = ops.border_seeds(37, 53)
    osx, osy = dist.border_seeds(37, 53)
    assert np.array_equal(sx.numpy(), osx) and np.array_equal(sy.numpy(), osy)
    edge = lambda im: (np.asarray(im, np.float32).mean(2) / 255.0).astype(np.float32)   # noqa: E731  (stand-in cost map)
    g = ops.GDT_box2distance(mode="mean", extract_edge_func=edge)
    out = g(imgs, [True, True, False], xy)
    ref = dist.gdt(edge(imgs[0]), *dist.border_seeds(*imgs[0].shape[:2]))[3:90, 2:100]
    assert np.array_equal(out[0].cpu().numpy(), ref)
    with pytest.raises(NotImplementedError):
        ops.MBD_box2distance(mode="center")(imgs, [True] * 3, xy)
    # single-crop entry points with the pybind signatures
    d = ops.MBD(torch.from_numpy(imgs[1]), sx, sy, 0.1, 4, 300)
    assert np.array_equal(d.cpu().numpy(), dist.mbd(imgs[1], osx, osy, 0.1, 4, 300))
