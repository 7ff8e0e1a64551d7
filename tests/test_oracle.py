"""The oracle (oracle/) pinned against golden vectors produced by running the reference
itself in the build container (tests/golden/gen_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from oracle import assigner, model as om, nms as onms, synth

VOTE_CFG = dict(type="vote", iou_threshold=0.65, cluster_score=["cls", "iou"], vote_score=["iou", "cls"],
                iou_enable=False, sima=0.025)
LEVEL_HW = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
ASSIGN_TAGS = ["g0", "g1", "g8", "g8b", "g20", "g3"]


def unpack_masks(g, tag):
    G = g[tag + "_boxes"].shape[0]
    if G == 0:
        return np.zeros((0, 480, 640), np.uint8)
    return np.unpackbits(g[tag + "_masks"], axis=1).reshape(G, 480, 640)


def test_anchors(golden):
    g = golden("anchors")
    for tag in ("a480x640", "a800x800"):
        sizes = [tuple(int(v) for v in s) for s in g[tag + "_sizes"]]
        mine = torch.cat(om.grid_anchors(sizes)).numpy()
        assert np.array_equal(mine, g[tag])
    assert g["a480x640"].shape == (6400, 4) and g["a800x800"].shape == (13343, 4)
    assert np.array_equal(g["a480x640"][0], [-32, -32, 32, 32])


@pytest.mark.parametrize("tag", ASSIGN_TAGS)
def test_assigner_numpy_rng(golden, tag):
    g = golden("assigner")
    rs = np.random.RandomState(int(g[tag + "_npseed"]))
    p2g, w = assigner.assign_points(g[tag + "_boxes"], g[tag + "_labels"], unpack_masks(g, tag), (480, 640, 3), rng=rs)
    assert np.array_equal(p2g, g[tag + "_p2g"].astype(np.int64))
    assert np.array_equal(w, g[tag + "_w"])


@pytest.mark.parametrize("tag", ASSIGN_TAGS)
def test_assigner_explicit_uniform_stream(golden, tag):
    """legacy_choice restatement == numpy's RandomState.choice, including stream position."""
    g = golden("assigner")
    u = np.random.RandomState(int(g[tag + "_npseed"])).random_sample(2048)
    p2g, w, used = assigner.assign_points_explicit(g[tag + "_boxes"], g[tag + "_labels"], unpack_masks(g, tag),
                                                   (480, 640, 3), u)
    assert np.array_equal(p2g, g[tag + "_p2g"].astype(np.int64))
    assert np.array_equal(w, g[tag + "_w"])
    assert used == int(g[tag + "_used"])


def test_legacy_choice_matches_numpy():
    rs = np.random.RandomState(99)
    for trial in range(200):
        n = int(rs.randint(1, 40))
        p = rs.rand(n).astype(np.float32) + 1e-3
        p = p / np.sum(p)
        seed = int(rs.randint(0, 2 ** 31 - 1))
        a = np.random.RandomState(seed)
        ref = a.choice(n, 10, p=p, replace=n < 10)
        after = a.random_sample()
        b = np.random.RandomState(seed)
        u = b.random_sample(512)
        mine, used = assigner.legacy_choice(p, 10, n < 10, u)
        assert np.array_equal(ref, mine)
        assert u[used] == after


def test_coder_and_overlaps(golden):
    g = golden("ops")
    pri, gts, pred = (torch.from_numpy(g[k]) for k in ("priors", "gts", "pred"))
    assert np.array_equal(om.tblr_encode(pri, gts).numpy(), g["enc"])
    assert np.array_equal(om.tblr_decode(pri, pred).numpy(), g["dec"])
    assert np.array_equal(om.tblr_decode(pri, pred, max_shape=(480, 640, 3)).numpy(), g["dec_clip"])
    b1, b2 = torch.from_numpy(g["b1"]), torch.from_numpy(g["b2"])
    assert np.allclose(om.overlaps_aligned(b1, b2).numpy(), g["iou_aligned"], rtol=0, atol=1e-7)
    assert np.allclose(om.overlaps_aligned(b1, b2, "giou").numpy(), g["giou_aligned"], rtol=0, atol=1e-7)
    assert np.allclose(om.overlaps_matrix(b1, b2).numpy(), g["iou_matrix"], rtol=0, atol=1e-7)
    assert np.allclose(om.overlaps_matrix(b1, b2, "giou").numpy(), g["giou_matrix"], rtol=0, atol=1e-7)
    f = om.focal_elementwise(torch.from_numpy(g["logits"]), torch.from_numpy(g["labels"]))
    assert np.allclose(f.numpy(), g["focal"], rtol=1e-6, atol=1e-7)


def test_overlaps_second_golden_set(golden):
    """the oracle's M x N / aligned overlaps against the larger reference fixtures (degenerate boxes, 700 x 1300 GIoU)"""
    g = golden("ops2")
    deg = torch.from_numpy(g["deg"])
    for mode in ("iou", "giou"):
        assert np.array_equal(om.overlaps_matrix(deg, deg, mode).numpy(), g["deg_" + mode])
        assert np.array_equal(om.overlaps_aligned(deg, deg.roll(1, 0), mode).numpy(), g["deg_al_" + mode])
    A, B = torch.from_numpy(g["big_a"]), torch.from_numpy(g["big_b"])
    big = om.overlaps_matrix(A, B, "giou")
    assert np.array_equal(big.reshape(-1)[torch.from_numpy(g["big_idx"])].numpy(), g["big_val"])
    assert abs(big.double().sum().item() - float(g["big_sum"])) < 1e-9 * abs(float(g["big_sum"]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_nms_ops(golden, tag):
    g = golden("nms")
    b, l = onms.vote_nms(g[tag + "_boxes"], g[tag + "_cls"], g[tag + "_labels"], VOTE_CFG, score_factor=g[tag + "_ctr"])
    assert np.array_equal(b, g[tag + "_vote_b"]) and np.array_equal(l, g[tag + "_vote_l"])
    b, l = onms.global_vote_nms(g[tag + "_boxes"], g[tag + "_cls"], g[tag + "_labels"], VOTE_CFG,
                                score_factor=g[tag + "_ctr"])
    assert np.array_equal(b, g[tag + "_gvote_b"]) and np.array_equal(l, g[tag + "_gvote_l"])
    ids, num = onms.cluster_nms(g[tag + "_boxes"], g[tag + "_cls"] * g[tag + "_ctr"], g[tag + "_labels"])
    assert np.array_equal(ids, g[tag + "_cl_ids"]) and np.array_equal(num, g[tag + "_cl_num"])


def test_nms_against_reference_binary(golden):
    """oracle/_ref = the reference's own vote_ext.cpp compiled here; also covers iou_enable=True."""
    from oracle import build_ref
    ref = build_ref.load("ref_vote_ext")
    if ref is None:
        pytest.skip("oracle/_ref not built")
    g = golden("nms")
    for tag in "abc":
        bx = torch.from_numpy(g[tag + "_boxes"])
        sc = torch.from_numpy(g[tag + "_cls"] * g[tag + "_ctr"])
        lb = torch.from_numpy(g[tag + "_labels"])
        for mine, theirs in ((onms.vote_nms_raw, ref.vote_nms), (onms.global_vote_nms_raw, ref.global_vote_nms)):
            for iou_enable in (False, True):
                rb, rl, rs = theirs(bx, sc, sc.clone(), lb, 0.65, iou_enable, 0.025)
                ob, ol, os_ = mine(bx.numpy(), sc.numpy(), sc.numpy(), lb.numpy(), 0.65, iou_enable, 0.025)
                assert np.array_equal(rb.numpy(), ob) and np.array_equal(rl.numpy(), ol) and np.array_equal(rs.numpy(), os_)


def test_nms_empty_and_single():
    b, l = onms.vote_nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), VOTE_CFG,
                         score_factor=np.zeros(0, np.float32))
    assert b.shape == (0, 5) and l.shape == (0,)
    b, l = onms.vote_nms(np.array([[1, 2, 30, 40]], np.float32), np.array([0.5], np.float32), np.array([3]), VOTE_CFG,
                         score_factor=np.array([0.5], np.float32))
    assert np.allclose(b, [[1, 2, 30, 40, 0.25]]) and l.tolist() == [3]
    dets, keep = onms.batched_nms(np.array([[0, 0, 10, 10], [1, 1, 10, 10], [0, 0, 10, 10]], np.float32),
                                  np.array([0.9, 0.8, 0.7], np.float32), np.array([0, 0, 1]), 0.5)
    assert keep.tolist() == [0, 2]


def synth_head_outputs(seed, B, cls_mean=-2.0):
    g = torch.Generator().manual_seed(seed)
    cls, reg, iou = [], [], []
    for (h, w) in LEVEL_HW:
        cls.append(torch.randn(B, 21, h, w, generator=g) * 1.5 + cls_mean)
        reg.append(torch.relu(torch.randn(B, 4, h, w, generator=g) * 2.0 + 2.5))
        iou.append(torch.randn(B, 1, h, w, generator=g))
    return cls, reg, iou


def assign_inputs(golden, tags=("g8", "g3")):
    a = golden("assigner")
    return ([torch.from_numpy(a[t + "_boxes"]) for t in tags], [torch.from_numpy(a[t + "_labels"]) for t in tags],
            [torch.from_numpy(a[t + "_p2g"].astype(np.int64)) for t in tags], [torch.from_numpy(a[t + "_w"]) for t in tags])


def test_head_loss_and_grads(golden):
    g = golden("head_loss")
    gt_b, gt_l, p2g, pw = assign_inputs(golden)
    cls, reg, iou = synth_head_outputs(7, 2)
    for t in cls + reg + iou:
        t.requires_grad_(True)
    losses, (labels, tgts, weights, pos) = om.head_loss(cls, reg, iou, gt_b, gt_l, p2g, pw)
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert np.allclose(losses[k].item(), g[k], rtol=1e-5), k
    assert np.array_equal(labels.numpy(), g["labels"].astype(np.int64))
    assert np.array_equal(pos.numpy(), g["pos"])
    assert np.array_equal(tgts[pos].numpy(), g["bbox_targets_pos"])
    assert np.array_equal(weights.numpy(), g["weights"])
    om.parse_losses(losses).backward()
    gc, gr, gi = (om.flatten_levels([t.grad for t in ts]) for ts in (cls, reg, iou))
    assert np.allclose(gc[::7].numpy(), g["g_cls_rows"], rtol=1e-4, atol=1e-9)
    assert np.allclose(gc[pos].numpy(), g["g_cls_pos"], rtol=1e-4, atol=1e-9)
    assert np.allclose(gr[pos].numpy(), g["g_reg_pos"], rtol=1e-4, atol=1e-8)
    assert np.allclose(gi.reshape(-1)[pos].numpy(), g["g_iou_pos"].reshape(-1), rtol=1e-4, atol=1e-9)
    assert np.isclose(gr.double().abs().sum().item(), g["g_reg_abs"], rtol=1e-5)
    assert np.isclose(gc.double().abs().sum().item(), g["g_cls_abs"], rtol=1e-5)


def test_head_loss_no_gt(golden):
    g = golden("head_loss")
    cls, reg, iou = synth_head_outputs(8, 2)
    e_b = [torch.zeros(0, 4)] * 2
    e_l = [torch.zeros(0, dtype=torch.long)] * 2
    e_p = [torch.full((6400,), -1, dtype=torch.long)] * 2
    e_w = [torch.ones(6400)] * 2
    losses, _ = om.head_loss(cls, reg, iou, e_b, e_l, e_p, e_w)
    assert np.allclose(losses["loss_cls"].item(), g["e_loss_cls"], rtol=1e-5)
    assert losses["loss_bbox"].item() == 0.0 and losses["loss_iou"].item() == 0.0
    assert g["e_loss_bbox"] == 0.0 and g["e_loss_iou"] == 0.0


@pytest.mark.parametrize("nms_type", ["vote", "global_vote"])
def test_get_bboxes(golden, nms_type):
    g = golden("get_bboxes")
    cls, reg, iou = synth_head_outputs(9, 2, cls_mean=-4.0)
    cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, max_per_img=100, nms=dict(VOTE_CFG, type=nms_type))
    dets = om.get_bboxes(cls, reg, iou, synth.img_metas(2), cfg, rescale=True)
    assert int(g["n_candidates"]) > 2000
    for i, (db, dl) in enumerate(dets):
        assert np.array_equal(dl, g[f"{nms_type}_{i}_l"])
        assert np.array_equal(db, g[f"{nms_type}_{i}_b"])


@pytest.mark.timeout(600)
def test_trainable_stem_against_reference(golden):
    """ResNet(frozen_stages=-1) (conv1 / bn1 / layer1 trainable, resnet.py:572-588): losses and sampled gradient elements of
    all 210 trainable tensors as written by the reference (tests/golden/model_grads_stem.npz, gen_golden.py stem)."""
    g = golden("model_grads_stem")
    torch.set_num_threads(8)
    det = om.OracleDetector(50, seed=0, frozen_stages=-1)
    img = synth.synth_images(0, 2)
    gt_b, gt_l, p2g, pw = assign_inputs(golden)
    losses = det.forward_train(img, gt_b, gt_l, p2g, pw)
    for k, ref in zip(("loss_cls", "loss_bbox", "loss_iou"), g["losses"]):
        assert np.isclose(losses[k].item(), float(ref), rtol=1e-4), k
    om.parse_losses(losses).backward()
    grads = det.named_grads()
    assert sorted(grads) == sorted(str(n) for n in g["names"]) and "backbone.conv1.weight" in grads
    from _grads import assert_sampled_grads
    assert_sampled_grads(grads, g, atol_total=1e-7, total=float(g["total_grad_norm"]))


def test_state_dict_inventory():
    sd = om.make_state_dict(50)
    n_all = sum(t.numel() for n, t in sd.items() if t.is_floating_point() and "running" not in n)
    n_train = sum(t.numel() for n, t in sd.items() if t.is_floating_point() and om.is_trainable(n))
    assert n_all == 32159327 and n_train == 31933983


@pytest.mark.timeout(600)
def test_full_model_against_reference(golden):
    """Seeded weights by NAME -> same features / head outputs / losses / per-parameter grad norms /
    detections as the reference model (tests/golden/model.npz)."""
    g = golden("model")
    torch.set_num_threads(8)
    det = om.OracleDetector(50, seed=0)
    img = synth.synth_images(0, 2)
    gt_b, gt_l, p2g, pw = assign_inputs(golden)
    cf = om.backbone(det.sd, img)
    pf = om.neck(det.sd, cf)
    outs = om.head(det.sd, pf)
    for i, f in enumerate(cf):
        v = f.detach().reshape(-1)[torch.from_numpy(g[f"c{i + 2}_idx"])].numpy()
        assert np.allclose(v, g[f"c{i + 2}_val"], rtol=1e-4, atol=1e-4 * float(g[f"c{i + 2}_absmean"]))
    for i, f in enumerate(pf):
        v = f.detach().reshape(-1)[torch.from_numpy(g[f"p{i + 3}_idx"])].numpy()
        assert np.allclose(v, g[f"p{i + 3}_val"], rtol=1e-4, atol=1e-4 * float(g[f"p{i + 3}_absmean"]))
    for nm, ts in zip(["cls", "reg", "iou"], outs):
        v = om.flatten_levels(ts).detach().reshape(-1)[torch.from_numpy(g[f"{nm}_idx"])].numpy()
        assert np.allclose(v, g[f"{nm}_val"], rtol=1e-4, atol=1e-4 * float(g[f"{nm}_absmean"]))
    losses, _ = om.head_loss(*outs, gt_b, gt_l, p2g, pw)
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert np.isclose(losses[k].item(), float(g[k]), rtol=1e-4), k
    om.parse_losses(losses).backward()
    grads = det.named_grads()
    names = [str(n) for n in g["grad_names"]]
    assert sorted(names) == sorted(grads.keys())
    mine = np.array([grads[n].double().norm().item() for n in names])
    assert np.allclose(mine, g["grad_norms"], rtol=2e-4, atol=1e-6 * float(g["total_grad_norm"]))
    # ... and the gradient tensors themselves: sampled elements of every parameter's gradient, written by the reference
    from _grads import assert_sampled_grads
    assert_sampled_grads(grads, golden("model_grads"), atol_total=1e-7, total=float(g["total_grad_norm"]))
    dets = det.simple_test(img, synth.img_metas(2))
    for i, (db, dl) in enumerate(dets):
        ref = g[f"det_{i}"]
        order = np.argsort(-db[:, 4], kind="stable")
        rorder = np.argsort(-ref[:, 4], kind="stable")
        assert np.array_equal(dl[order], ref[rorder, 5].astype(np.int64))
        assert np.allclose(db[order], ref[rorder, :5], rtol=1e-4, atol=1e-3)


def test_mask_oracle_identities():
    """oracle/masks.py (parity unpinned for the cv2 nearest rule): structural identities that any correct
    restatement satisfies."""
    from oracle import masks as om
    rng = np.random.RandomState(0)
    m = (rng.rand(3, 37, 53) * 256).astype(np.uint8)
    assert (om.resize_nearest(m, (37, 53)) == m).all()                       # identity resize
    up = om.resize_nearest(m, (74, 106))                                     # x2: every pixel replicated 2x2
    assert (up[:, ::2, ::2] == m).all() and (up[:, 1::2, 1::2] == m).all()
    for d in ("horizontal", "vertical", "diagonal"):
        assert (om.flip(om.flip(m, d), d) == m).all()
    assert (om.flip(m, "diagonal") == om.flip(om.flip(m, "horizontal"), "vertical")).all()
    p = om.pad(m, (40, 64), 9)
    assert (p[:, :37, :53] == m).all() and (p[:, 37:] == 9).all() and (p[:, :, 53:] == 9).all()
    n = om.normalize(np.stack([m[0], np.zeros_like(m[0]), (m[2] > 128).astype(np.uint8) * 255]))
    assert set(np.unique(n)) <= {0, 1} and n[1].max() == 0 and (n[2] == (m[2] > 128)).all()
    assert om.rescale_size((640, 480), (1333, 800)) == (1067, 800) and om.rescale_size((640, 480), 0.5) == (320, 240)


def test_distance_oracle_matches_reference_build():
    """oracle/dist.c (MBD / GDT raster scans) == the reference's bbox2distance_ext.cpp compiled in place (oracle/_ref),
    bit for bit, on ragged crop sizes incl. crops larger than base_size^2 (integer size factor) and odd niter."""
    import torch
    from oracle import build_ref, dist
    ref = build_ref.load("ref_bbox2distance_ext")
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    rs = np.random.RandomState(0)
    for (h, w, niter, base) in ((150, 200, 4, 300), (37, 53, 4, 300), (64, 41, 3, 40), (2, 2, 4, 300), (90, 90, 5, 30)):
        img = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        img[h // 4:3 * h // 4, w // 4:3 * w // 4] //= 3
        sx, sy = dist.border_seeds(h, w)
        a = dist.mbd(img, sx, sy, 0.1, niter, base)
        b = ref.MBD(torch.from_numpy(img), torch.from_numpy(sx), torch.from_numpy(sy), 0.1, niter, base).numpy()
        assert np.array_equal(a, b), (h, w)
        cost = rs.rand(h, w).astype(np.float32)
        c = dist.gdt(cost, sx, sy)
        d = ref.GDT(torch.from_numpy(cost), torch.from_numpy(sx), torch.from_numpy(sy)).numpy()
        assert np.array_equal(c, d), (h, w)


def test_imgproc_oracle_against_scipy():
    """oracle/imgproc.py (the OpenCV operations around MBD / GDT, cv2 absent -> restated) against scipy.ndimage and
    analytic cases: Gaussian / Sobel by correlation with mirror borders (= BORDER_REFLECT_101), bilinear resize on the
    half-pixel grid, identity / constant images."""
    import scipy.ndimage as ndi
    from oracle import imgproc as ip
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    k = ip.gauss9_taps().astype(np.float64)
    assert abs(k.sum() - 1) < 1e-7 and np.allclose(k, k[::-1]) and k.argmax() == 4
    ref = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(ip.gaussian_blur9_u8(img).astype(float) - np.rint(ref)).max() <= 1           # float32 passes vs float64
    blur = np.stack([ndi.correlate(img[..., c].astype(np.int64), np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]]), mode="mirror")
                     for c in range(3)], -1)
    b = (blur + 8) >> 4
    gray = ((b[..., 0] * 4899 + b[..., 1] * 9617 + b[..., 2] * 1868 + 8192) >> 14).astype(np.float64)
    gx = ndi.correlate(gray, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], float), mode="mirror")
    gy = ndi.correlate(gray, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], float), mode="mirror")
    e = np.abs(0.5 * gx + 0.5 * gy)
    assert np.array_equal(ip.sobel_edge(img), (e / e.max()).astype(np.float32))                 # integer-valued: exact
    assert np.array_equal(ip.resize_linear_u8(img, (53, 37)), img)                              # identity
    assert (ip.resize_linear_u8(np.full((20, 30, 3), 77, np.uint8), (45, 31)) == 77).all()      # constants survive
    up = ip.resize_linear_u8(img, (150, 105))
    f = ip.resize_linear_float(img[..., 0].astype(np.float64), (150, 105))
    assert np.abs(up[..., 0].astype(float) - f).max() < 1.0                                     # fixed point vs float: < 1 LSB
    yy, xx = (np.arange(105) + 0.5) * (37 / 105) - 0.5, (np.arange(150) + 0.5) * (53 / 150) - 0.5
    ref = ndi.map_coordinates(img[..., 0].astype(np.float64), np.meshgrid(yy, xx, indexing="ij"), order=1, mode="nearest")
    assert np.abs(f - ref).max() < 2e-3                                                         # float32 coordinates in cv2's rule
    dn = ip.resize_linear_float(img[..., 1].astype(np.float32), (20, 11))
    assert dn.dtype == np.float32 and dn.shape == (11, 20) and img[..., 1].min() <= dn.min() and dn.max() <= img[..., 1].max()
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (9, 1, 3))              # linear ramp stays linear
    r2 = ip.resize_linear_u8(ramp, (100, 9))[4, 2:-2, 0].astype(int)
    assert set(np.diff(r2)) <= {1, 2, 3}


ASSIGN_OPT_TAGS = ["nobal", "mulpro", "mulpro_tiny", "adapt", "adapt20", "adapt_nobal", "all3", "g_mulpro", "g_all3", "g_plain",
                   "unif", "unif_tiny", "unif_nobal", "unif20_all"]


def assigner_opt_case(g, tag):
    """inputs of an assigner_opts.npz case (regenerated by oracle/synth.py) and the reference's constructor options"""
    sseed, G, tiny, npseed, graded = (int(v) for v in g[tag + "_synth"])
    boxes, labels, masks = synth.synth_objects(sseed, G, tiny_visible=bool(tiny))
    maps = synth.graded_maps(masks) if graded else masks
    f = int(g[tag + "_flags"])
    opts = dict(balance_sample=bool(f & 1), multiply_samplepro_for_weight=bool(f & 2), adapt_positive_num=bool(f & 4),
                random_sample_by_distance=not (f & 8))
    return boxes, labels, maps, npseed, opts


@pytest.mark.parametrize("tag", ASSIGN_OPT_TAGS)
def test_assigner_constructor_options(golden, tag):
    """balance_sample=False, multiply_samplepro_for_weight, adapt_positive_num, random_sample_by_distance=False
    (label_assignment.py:30-46, 88-131), alone and together, on binary masks and on graded float maps: the oracle against
    outputs of the reference itself, with NumPy's own RandomState and with the explicit stream of the RandomState's raw 32-bit
    outputs (= what the HIP kernel consumes), incl. the stream position."""
    g = golden("assigner_opts")
    boxes, labels, maps, npseed, opts = assigner_opt_case(g, tag)
    p2g, w = assigner.assign_points(boxes, labels, maps, (480, 640, 3), rng=np.random.RandomState(npseed), **opts)
    assert np.array_equal(p2g, g[tag + "_p2g"].astype(np.int64))
    assert np.array_equal(w, g[tag + "_w"])
    words = np.random.RandomState(npseed)._bit_generator.random_raw(65536)
    p2g, w, used = assigner.assign_points_explicit(boxes, labels, maps, (480, 640, 3), words=words, **opts)
    assert np.array_equal(p2g, g[tag + "_p2g"].astype(np.int64)) and np.array_equal(w, g[tag + "_w"])
    assert used == int(g[tag + "_used_words"])


def test_legacy_integer_draws_match_numpy():
    """choice(n, size, replace) WITHOUT p of the legacy RandomState (randint / permutation: masked rejection on raw 32-bit
    outputs) restated over an explicit word stream == numpy itself, values and stream position."""
    rs = np.random.RandomState(17)
    for trial in range(300):
        n = int(rs.randint(1, 3000))
        size = int(rs.randint(0, 40))
        replace = bool(rs.randint(0, 2)) or size > n
        seed = int(rs.randint(0, 2 ** 31 - 1))
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        ref = a.choice(a=n, size=size, replace=replace)
        words = b._bit_generator.random_raw(8192)
        mine, used = assigner.legacy_choice_uniform(n, size, replace, words)
        assert np.array_equal(mine, ref), (trial, n, size, replace)
        assert a.random_sample() == assigner.uniforms_from_words(words[used:used + 2])[0]
