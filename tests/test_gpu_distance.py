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
        ops.GDT_box2distance()                       # edge_mode='sed' (the reference default) needs cv2.ximgproc's model file
    # single-crop entry points with the pybind signatures
    d = ops.MBD(torch.from_numpy(imgs[1]), sx, sy, 0.1, 4, 300)
    assert np.array_equal(d.cpu().numpy(), dist.mbd(imgs[1], osx, osy, 0.1, 4, 300))


def test_imgproc_kernels_bit_exact():
    """resize (uint8 fixed point, float32, float64), 9x9 Gaussian and the Sobel edge map, batched over ragged crops ==
    oracle/imgproc.py bit for bit (up- and down-scaling, 2-pixel crops, odd sizes)."""
    from oracle import imgproc as ip
    from radet_amd import ops
    rs = np.random.RandomState(3)
    imgs = _crops(rs)
    dsz = [(150, 112), (53, 37), (20, 33), (5, 7), (45, 45), (300, 196), (150, 2)]       # (w, h) like cv2.resize
    for g, i, d in zip(ops.resize_batch(imgs, dsz), imgs, dsz):
        assert g.dtype == torch.uint8 and np.array_equal(g.cpu().numpy(), ip.resize_linear_u8(i, d)), (i.shape, d)
    for dt in (np.float32, np.float64):
        maps = [rs.rand(h, w).astype(dt) * 50 for h, w in SIZES]
        for g, m, d in zip(ops.resize_batch(maps, dsz), maps, dsz):
            assert np.array_equal(g.cpu().numpy(), ip.resize_linear_float(m, d)), (m.shape, d, dt)
    big = [i for i in imgs if min(i.shape[:2]) >= 2]
    for g, i in zip(ops.gaussian_blur9_batch(big), big):
        assert np.array_equal(g.cpu().numpy(), ip.gaussian_blur9_u8(i)), i.shape
    for g, i in zip(ops.sobel_edge_batch(big), big):
        assert g.dtype == torch.float32 and np.array_equal(g.cpu().numpy(), ip.sobel_edge(i), equal_nan=True), i.shape
    flat = np.full((6, 9, 3), 10, np.uint8)                                               # no edges: 0 / 0 = nan like np
    assert bool(torch.isnan(ops.sobel_edge_batch([flat])[0]).all())


def _oracle_center(img, transform, interval=3):
    """bbox2distance_wrapper.py:80-88 / 170-177 on the host: resize -> blur -> transform -> resize back"""
    from oracle import dist, imgproc as ip
    h, w = img.shape[:2]
    ratio = 150 / min(w, h)
    nw, nh = int(w * ratio), int(h * ratio)
    blurred = ip.gaussian_blur9_u8(ip.resize_linear_u8(img, (nw, nh)))
    sx, sy = dist.border_seeds(nh, nw, interval)
    if transform == "mbd":
        d = dist.mbd(blurred, sx, sy, 0.1, 4, 300)
    else:
        d = dist.gdt(ip.sobel_edge(blurred), sx, sy)
    return ip.resize_linear_float(d, (w, h))


def test_box2distance_center_mode():
    """mode='center' (the reference default) of both wrappers == the host restatement, bit for bit"""
    from radet_amd import ops
    rs = np.random.RandomState(2)
    imgs = [i for i in _crops(rs) if min(i.shape[:2]) >= 30]
    xy = [(2, 3, i.shape[1] - 4, i.shape[0] - 2) for i in imgs]
    en = [True] * len(imgs)
    en[1] = False
    out = ops.MBD_box2distance(mode="center")(imgs, en, xy)
    for img, e, b, o in zip(imgs, en, xy, out):
        if e:
            ref = _oracle_center(img, "mbd")[b[1]:b[3], b[0]:b[2]]
            assert o.dtype == torch.float64 and np.array_equal(o.cpu().numpy(), ref), img.shape
        else:
            assert bool((o == 1).all()) and o.shape == (b[3] - b[1], b[2] - b[0])
    out = ops.GDT_box2distance(edge_mode="sobel", mode="center")(imgs, en, xy)
    for img, e, b, o in zip(imgs, en, xy, out):
        if e:
            assert o.dtype == torch.float32 and np.array_equal(o.cpu().numpy(), _oracle_center(img, "gdt")[b[1]:b[3], b[0]:b[2]])
    m = ops.GDT_box2distance(edge_mode="sobel", mode="mean")(imgs[:2], [True, True], xy[:2])
    from oracle import dist, imgproc as ip
    ref = dist.gdt(ip.sobel_edge(imgs[0]), *dist.border_seeds(*imgs[0].shape[:2]))
    assert np.array_equal(m[0].cpu().numpy(), ref[xy[0][1]:xy[0][3], xy[0][0]:xy[0][2]])


@pytest.mark.parametrize("kind", ["mbd", "gdt"])
def test_mask_free_sampler_end_to_end(kind):
    """GenerateDistanceMap(with_gt_mask=False) -> LabelAssignment on float distance maps (loading.py:586-645,
    label_assignment.py:85-201): crops (incl. boxes touching the image border and a box too small to be transformed),
    GPU transform, maps pasted into the image, GPU assigner == the host restatement with the same RNG streams."""
    import random
    from oracle import assigner as oa
    from radet_amd.datasets import GenerateDistanceMap, LabelAssignment
    rs = np.random.RandomState(5)
    H, W = 480, 640
    img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
    img[100:300, 150:420] //= 4
    boxes = np.array([[150.3, 100.8, 420.2, 300.9], [0.0, 380.5, 130.7, 479.0], [500.2, 10.1, 639.0, 120.6],
                      [300.0, 300.0, 320.0, 325.0]], np.float32)                          # the last one: area < 32^2
    labels = np.array([3, 7, 1, 5], np.int64)
    kw = dict(distance_transform="mbd") if kind == "mbd" else dict(distance_transform="gdt", edge_mode="sobel")
    gdm = GenerateDistanceMap(with_gt_mask=False, **kw)
    random.seed(11)
    res = gdm(dict(img=img, img_shape=(H, W, 3), gt_bboxes=boxes, gt_labels=labels))
    dm = res["distance_maps"]
    assert dm.dtype == torch.float32 and tuple(dm.shape) == (4, H, W) and dm.is_cuda
    # host restatement of the same pipeline
    random.seed(11)
    crops, enable, regions = gdm.crop_boxes(img, (H, W), boxes)
    assert enable.tolist() == [True, True, True, False]
    ref = np.zeros((4, H, W), np.float32)
    for k, (c, e, r, b) in enumerate(zip(crops, enable, regions, boxes)):
        m = _oracle_center(c, kind) if e else np.ones(c.shape[:2], np.float32)
        bi = b.astype(np.int_)
        ref[k, bi[1]:bi[3], bi[0]:bi[2]] = m[r[1]:r[3], r[0]:r[2]].astype(np.float32)
    assert np.array_equal(dm.cpu().numpy(), ref, equal_nan=True)
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
    np.random.seed(21)
    out = la(dict(res))
    p2g, pw = oa.assign_points(boxes, labels, ref, (H, W, 3), rng=np.random.RandomState(21))
    assert np.array_equal(out["points_to_gt_index"], p2g) and np.array_equal(out["points_weight"], pw)
    assert (p2g > 0).sum() > 10
