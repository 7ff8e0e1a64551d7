"""Randomised parity sweeps on the GPU: many seeded cases of the index-deciding kernels against the oracle
(bit-exact), including ties, degenerate boxes, crowded labels and ragged batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOTE_CFG = dict(type="vote", iou_threshold=0.65, cluster_score=["cls", "iou"], vote_score=["iou", "cls"],
                iou_enable=False, sima=0.025)


def random_boxes(rs, n, n_labels, crowd):
    nb = max(1, n // crowd)
    base = rs.rand(nb, 2) * np.array([500.0, 350.0])
    wh = rs.rand(nb, 2) * 150 + 5
    b = np.concatenate([base, base + wh], 1).astype(np.float32)
    idx = rs.randint(0, nb, n)
    boxes = (b[idx] + rs.randn(n, 4).astype(np.float32) * rs.choice([0.5, 3.0, 8.0])).astype(np.float32)
    labels = rs.randint(0, n_labels, nb)[idx].astype(np.int64)
    return boxes, labels


@pytest.mark.parametrize("seed", range(12))
def test_nms_family_random(seed):
    from oracle import nms as onms
    from radet_amd import ops
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([1, 2, 7, 64, 65, 500, 1777, 4096]))
    boxes, labels = random_boxes(rs, n, int(rs.choice([1, 3, 21])), int(rs.choice([1, 4, 30])))
    cls = (rs.rand(n) * 0.9 + 0.05).astype(np.float32)
    ctr = (rs.rand(n) * 0.9 + 0.05).astype(np.float32)
    if seed % 3 == 0:                      # degenerate (zero-area) and duplicated boxes
        boxes[: max(1, n // 10), 2:] = boxes[: max(1, n // 10), :2]
        if n > 3:
            boxes[-2] = boxes[-3]
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    for thr in (0.3, 0.65):
        cfg = dict(VOTE_CFG, iou_threshold=thr)
        for fn, ofn in ((ops.vote_nms, onms.vote_nms), (ops.global_vote_nms, onms.global_vote_nms)):
            for max_num in (0, 100):
                b, l = fn(t(boxes), t(cls), t(labels), cfg, score_factor=t(ctr), max_num=max_num)
                ob, ol = ofn(boxes, cls, labels, cfg, score_factor=ctr, max_num=max_num)
                assert np.array_equal(l.numpy(), ol), (seed, thr, fn.__name__)
                # NaN-safe bit comparison (a zero-weight cluster yields 0/0 in the reference too)
                assert np.array_equal(b.numpy().view(np.uint32), ob.view(np.uint32)), (seed, thr, fn.__name__)
        ids, num = ops.cluster_nms(boxes, cls * ctr, labels, thr)
        oids, onum = onms.cluster_nms(boxes, cls * ctr, labels, thr)
        assert np.array_equal(ids.numpy(), oids) and np.array_equal(num.numpy(), onum)
        dets, keep = ops.batched_nms(t(boxes), t(cls * ctr), t(labels), dict(type="nms", iou_threshold=thr))
        odets, okeep = onms.batched_nms(boxes, cls * ctr, labels, thr)
        assert np.array_equal(keep.numpy(), okeep) and np.array_equal(dets.numpy(), odets)


def test_nms_score_ties_follow_index_order():
    """Equal scores: the reference's torch::sort is unstable; ours (and the oracle's) is index-stable."""
    from oracle import nms as onms
    from radet_amd import ops
    rs = np.random.RandomState(5)
    boxes, labels = random_boxes(rs, 300, 3, 6)
    cls = np.round(rs.rand(300) * 8) / 10 + 0.1          # heavy ties
    cls = cls.astype(np.float32)
    one = np.ones(300, np.float32)
    b, l = ops.vote_nms(torch.from_numpy(boxes), torch.from_numpy(cls), torch.from_numpy(labels), VOTE_CFG,
                        score_factor=torch.from_numpy(one))
    ob, ol = onms.vote_nms(boxes, cls, labels, VOTE_CFG, score_factor=one)
    assert np.array_equal(l.numpy(), ol) and np.array_equal(b.numpy(), ob)


@pytest.mark.parametrize("seed", range(6))
def test_assigner_random_batches_with_options(seed):
    """The constructor options of label_assignment.py:30-46 on ragged random batches with graded float maps, against the
    oracle (which the reference's goldens pin): flags cycle through the combinations, incl. the uniform integer draw."""
    from oracle import assigner as oa, synth
    from radet_amd.datasets import LabelAssignment
    rs = np.random.RandomState(700 + seed)
    H, W = [(480, 640), (200, 264), (320, 320)][seed % 3]
    f = (seed * 3 + 2) % 8 + (8 if seed % 3 == 1 else 0)
    opts = dict(balance_sample=bool(f & 1), multiply_samplepro_for_weight=bool(f & 2), adapt_positive_num=bool(f & 4),
                random_sample_by_distance=not (f & 8))
    B = 4
    boxes, maps, rngs, orngs = [], [], [], []
    for i in range(B):
        G = int(rs.randint(1, 11))
        bx = np.zeros((G, 4), np.float32)
        mk = np.zeros((G, H, W), np.uint8)
        for g in range(G):
            w, h = rs.randint(4, W // 2), rs.randint(4, H // 2)
            x, y = rs.randint(0, W - w), rs.randint(0, H - h)
            bx[g] = (x, y, x + w, y + h)
            kind = rs.randint(0, 3)
            if kind == 0:
                mk[g, y:y + h, x:x + w] = 1
            elif kind == 1:
                mk[g, y:y + h, x + w // 2:x + w] = 1
            else:
                mk[g, y + h // 2:y + h // 2 + 3, x + w // 2:x + w // 2 + 3] = 1
        boxes.append(bx); maps.append(synth.graded_maps(mk) if seed % 2 else mk)
        rngs.append(np.random.RandomState(1900 + seed * 10 + i)); orngs.append(np.random.RandomState(1900 + seed * 10 + i))
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, **opts)
    p2g, pw = la.assign_batch(boxes, maps, (H, W, 3), rngs=rngs)
    p2g, pw = p2g.cpu().numpy(), pw.cpu().numpy()
    for i in range(B):
        rp, rw = oa.assign_points(boxes[i], np.zeros(len(boxes[i]), np.int64), maps[i], (H, W, 3), rng=orngs[i], **opts)
        assert np.array_equal(p2g[i], rp) and np.array_equal(pw[i], rw), (seed, i, opts)
        assert rngs[i].random_sample() == orngs[i].random_sample()


@pytest.mark.parametrize("seed", range(6))
def test_assigner_random_batches(seed):
    """Ragged batches (0..12 gts per image, overlapping / tiny / fully occluded objects), 200x264 .. 480x640."""
    from oracle import assigner as oa
    from radet_amd.datasets import LabelAssignment
    rs = np.random.RandomState(300 + seed)
    H, W = [(480, 640), (200, 264), (320, 320)][seed % 3]
    B = 5
    boxes, masks, rngs, orngs = [], [], [], []
    for i in range(B):
        G = int(rs.randint(0, 13))
        bx = np.zeros((G, 4), np.float32)
        mk = np.zeros((G, H, W), np.uint8)
        for g in range(G):
            w, h = rs.randint(4, W // 2), rs.randint(4, H // 2)
            x, y = rs.randint(0, W - w), rs.randint(0, H - h)
            bx[g] = (x, y, x + w, y + h)
            kind = rs.randint(0, 4)
            if kind == 0:
                mk[g, y:y + h, x:x + w] = 1
            elif kind == 1:
                mk[g, y:y + h, x + w // 2:x + w] = 1
            elif kind == 2:
                mk[g, y + h // 2:y + h // 2 + 3, x + w // 2:x + w // 2 + 3] = 1
            # kind 3: fully occluded (all-zero mask)
        boxes.append(bx); masks.append(mk)
        rngs.append(np.random.RandomState(900 + seed * 10 + i)); orngs.append(np.random.RandomState(900 + seed * 10 + i))
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, balance_sample=True)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3), rngs=rngs)
    p2g, pw = p2g.cpu().numpy(), pw.cpu().numpy()
    for i in range(B):
        rp, rw = oa.assign_points(boxes[i], np.zeros(len(boxes[i]), np.int64), masks[i], (H, W, 3), rng=orngs[i])
        assert np.array_equal(p2g[i], rp) and np.array_equal(pw[i], rw), (seed, i)
        assert rngs[i].random_sample() == orngs[i].random_sample()      # stream position preserved


def test_assigner_shared_rng_consumes_one_stream_in_order():
    """The reference's loader runs the sampler image after image on NumPy's GLOBAL RandomState: image i starts where
    image i-1 stopped.  assign_batch with no rngs (or one RNG object for several images) must reproduce exactly that --
    same assignments AND the same final stream position."""
    from oracle import assigner as oa
    from radet_amd.datasets import LabelAssignment
    rs = np.random.RandomState(77)
    H, W, B = 320, 320, 4
    boxes, masks = [], []
    for i in range(B):
        G = int(rs.randint(1, 7))
        bx = np.zeros((G, 4), np.float32)
        mk = np.zeros((G, H, W), np.uint8)
        for g in range(G):
            w, h = rs.randint(20, W // 2), rs.randint(20, H // 2)
            x, y = rs.randint(0, W - w), rs.randint(0, H - h)
            bx[g] = (x, y, x + w, y + h)
            mk[g, y:y + h, x + w // 3:x + w] = 1
        boxes.append(bx); masks.append(mk)
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, balance_sample=True)
    np.random.seed(4242)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3))                       # rngs=None: the global np.random, shared
    after = np.random.random_sample()
    ref = np.random.RandomState(4242)
    for i in range(B):
        rp, rw = oa.assign_points(boxes[i], np.zeros(len(boxes[i]), np.int64), masks[i], (H, W, 3), rng=ref)
        assert np.array_equal(p2g[i].cpu().numpy(), rp) and np.array_equal(pw[i].cpu().numpy(), rw), i
    assert after == ref.random_sample()
    shared = np.random.RandomState(9)                                       # one explicit RandomState for every image
    ref = np.random.RandomState(9)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3), rngs=[shared] * B)
    for i in range(B):
        rp, rw = oa.assign_points(boxes[i], np.zeros(len(boxes[i]), np.int64), masks[i], (H, W, 3), rng=ref)
        assert np.array_equal(p2g[i].cpu().numpy(), rp), i
    assert shared.random_sample() == ref.random_sample()
    # None, the np.random module and np.random.mtrand._rand are ONE generator, whatever mixture names it
    np.random.seed(4242)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3), rngs=[None, np.random, np.random.mtrand._rand, None])
    ref = np.random.RandomState(4242)
    for i in range(B):
        rp, rw = oa.assign_points(boxes[i], np.zeros(len(boxes[i]), np.int64), masks[i], (H, W, 3), rng=ref)
        assert np.array_equal(p2g[i].cpu().numpy(), rp) and np.array_equal(pw[i].cpu().numpy(), rw), i
    assert np.random.random_sample() == ref.random_sample()
    with pytest.raises(ValueError):
        la.assign_batch([np.zeros((300, 4), np.float32)], [np.zeros((300, 8, 8), np.uint8)], (8, 8, 3))


@pytest.mark.parametrize("n,n_labels,crowd", [(700, 1, 4), (1024, 1, 30), (1025, 1, 4), (4096, 1, 30), (4097, 1, 30),
                                              (8192, 1, 30), (6000, 2, 4), (5000, 3, 1),
                                              # above the LDS sorts (8192): global-memory sorts; a single label of > 8192
                                              # boxes also takes the general greedy pass over the mask rows
                                              (8193, 1, 30), (12000, 2, 4), (20000, 21, 30), (20000, 1, 30), (30000, 5, 8)])
def test_nms_long_label_segments(n, n_labels, crowd):
    """Crowded classes: every clustering path of the kernel -- register-resident segments (<= 64 boxes), the mask +
    register-resident greedy pass (<= 8192 per label), and for N > 8192 (the reference ops take any N: cluster_ext.cpp:4-87,
    vote_ext.cpp:70-207) the global-memory sorts and the general greedy pass -- bit-exact, all modes."""
    from oracle import nms as onms
    from radet_amd import ops
    rs = np.random.RandomState(n + n_labels)
    boxes, labels = random_boxes(rs, n, n_labels, crowd)
    cls = (rs.rand(n) * 0.9 + 0.05).astype(np.float32)
    ctr = (rs.rand(n) * 0.9 + 0.05).astype(np.float32)
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    for iou_enable in (False, True):
        cfg = dict(VOTE_CFG, iou_threshold=0.5, iou_enable=iou_enable)
        for fn, ofn in ((ops.vote_nms, onms.vote_nms), (ops.global_vote_nms, onms.global_vote_nms)):
            b, l = fn(t(boxes), t(cls), t(labels), cfg, score_factor=t(ctr), max_num=100)
            ob, ol = ofn(boxes, cls, labels, cfg, score_factor=ctr, max_num=100)
            assert np.array_equal(l.numpy(), ol), (fn.__name__, iou_enable)
            if not iou_enable:
                assert np.array_equal(b.numpy().view(np.uint32), ob.view(np.uint32)), (fn.__name__, iou_enable)
            else:
                # iou_enable rescales vote scores by expf(-(1-iou)^2/sigma): the device expf and glibc's differ by
                # <= 1 ulp, so the voted fp32 boxes agree to rounding (north_star: 1e-4), not bit for bit; a 1-ulp
                # weight can also flip a member in / out of the 1-sigma vote window (rare)
                close = np.isclose(b.numpy(), ob, rtol=1e-4, atol=1e-3) | (np.isnan(b.numpy()) & np.isnan(ob))
                assert close.all(axis=1).mean() >= 0.98, (fn.__name__, close.all(axis=1).mean())
    ids, num = ops.cluster_nms(boxes, cls * ctr, labels, 0.5)
    oids, onum = onms.cluster_nms(boxes, cls * ctr, labels, 0.5)
    assert np.array_equal(ids.numpy(), oids) and np.array_equal(num.numpy(), onum)
    dets, keep = ops.batched_nms(t(boxes), t(cls * ctr), t(labels), dict(type="nms", iou_threshold=0.5))
    odets, okeep = onms.batched_nms(boxes, cls * ctr, labels, 0.5)
    assert np.array_equal(keep.numpy(), okeep) and np.array_equal(dets.numpy(), odets)
