"""Reference-API surface on the GPU: the stand-alone box / loss operators behind the registered classes
(bbox_overlaps / BboxOverlaps2D, TBLRBBoxCoder, FocalLoss, GIoULoss, CrossEntropyLoss, multiclass_nms) and the
RADetHead methods (loss / get_bboxes / forward_train), each against goldens produced by running the reference
(tests/golden/gen_golden.py).  Everything executes in libradet_hip.so through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVEL_HW = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
TOL = 1e-4   # BASELINE.json north_star: fp32 boxes / scores within 1e-4


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ------------------------------------------------------------------------------------------------ overlaps
def test_bbox_overlaps_golden_bit_exact(golden):
    from radet_amd.core import BboxOverlaps2D, bbox_overlaps
    g = golden("ops")
    b1, b2 = T(g["b1"]), T(g["b2"])
    assert np.array_equal(bbox_overlaps(b1, b2, is_aligned=True).numpy(), g["iou_aligned"])
    assert np.array_equal(bbox_overlaps(b1, b2, mode="giou", is_aligned=True, eps=1e-6).numpy(), g["giou_aligned"])
    assert np.array_equal(bbox_overlaps(b1, b2).numpy(), g["iou_matrix"])
    assert np.array_equal(bbox_overlaps(b1, b2, mode="giou").numpy(), g["giou_matrix"])
    calc = BboxOverlaps2D()
    with5 = torch.cat([b1, torch.ones(32, 1)], 1)
    assert np.array_equal(calc(with5.cuda(), b2.cuda()).cpu().numpy(), g["iou_matrix"])     # device in -> device out
    g2 = golden("ops2")
    a, b = T(g2["ov_a"]), T(g2["ov_b"])
    assert np.array_equal(bbox_overlaps(a, b, mode="iof").numpy(), g2["iof_matrix"])
    assert np.array_equal(bbox_overlaps(a, b[:37], mode="iof", is_aligned=True).numpy(), g2["iof_aligned"])
    ab, bb = T(g2["ovb_a"]), T(g2["ovb_b"])                                                  # batch dims (2, 3)
    assert np.array_equal(bbox_overlaps(ab, bb, mode="giou").numpy(), g2["ovb_giou"])
    assert np.array_equal(bbox_overlaps(ab, ab.flip(2), is_aligned=True).numpy(), g2["ovb_iou_aligned"])
    deg = T(g2["deg"])                                                                       # zero-area / identical / touching
    for mode in ("iou", "iof", "giou"):
        assert np.array_equal(bbox_overlaps(deg, deg, mode=mode).numpy(), g2["deg_" + mode])
        assert np.array_equal(bbox_overlaps(deg, deg.roll(1, 0), mode=mode, is_aligned=True).numpy(), g2["deg_al_" + mode])


def test_bbox_overlaps_big_matrix_and_empty(golden):
    from radet_amd.core import bbox_overlaps
    g = golden("ops2")
    A, B = T(g["big_a"]).cuda(), T(g["big_b"]).cuda()                   # 700 x 1300: 11 x 6 workgroup tiles, ragged edges
    out = bbox_overlaps(A, B, mode="giou")
    assert out.is_cuda and tuple(out.shape) == (700, 1300)
    assert np.array_equal(out.reshape(-1)[T(g["big_idx"]).cuda()].cpu().numpy(), g["big_val"])
    assert abs(out.double().sum().item() - float(g["big_sum"])) <= 1e-9 * abs(float(g["big_sum"]))
    iou = bbox_overlaps(A, B)
    assert abs(iou.double().sum().item() - float(g["big_iou_sum"])) <= 1e-9 * abs(float(g["big_iou_sum"]))
    # properties at size: symmetry of IoU, transposition, diagonal of the matrix == aligned
    assert torch.equal(iou, bbox_overlaps(B, A).t())
    n = 700
    assert torch.equal(bbox_overlaps(A, B[:n]).diagonal(), bbox_overlaps(A, B[:n], is_aligned=True))
    empty = torch.empty(0, 4)
    assert tuple(bbox_overlaps(empty, A.cpu()).shape) == (0, 700) and tuple(bbox_overlaps(A.cpu(), empty).shape) == (700, 0)
    assert tuple(bbox_overlaps(empty, empty, is_aligned=True).shape) == (0,)
    with pytest.raises(AssertionError):
        bbox_overlaps(A, B, mode="diou")


# ------------------------------------------------------------------------------------------------ TBLR coder
def test_tblr_coder_golden_bit_exact(golden):
    from radet_amd.core import bboxes2tblr, build_bbox_coder, tblr2bboxes
    g = golden("ops")
    coder = build_bbox_coder(dict(type="TBLRBBoxCoder", normalizer=1 / 8))
    pri, gts, pred = T(g["priors"]), T(g["gts"]), T(g["pred"])
    assert np.array_equal(coder.encode(pri, gts).numpy(), g["enc"])
    assert np.array_equal(coder.decode(pri, pred).numpy(), g["dec"])
    assert np.array_equal(coder.decode(pri.cuda(), pred.cuda(), max_shape=(480, 640, 3)).cpu().numpy(), g["dec_clip"])
    g2 = golden("ops2")
    pri, gts, pred, nm = T(g2["t_pri"]), T(g2["t_gts"]), T(g2["t_pred"]), [float(v) for v in g2["t_nm"]]
    assert np.array_equal(bboxes2tblr(pri, gts, normalizer=nm).numpy(), g2["t_enc4"])
    assert np.array_equal(bboxes2tblr(pri, gts, normalizer=4.0, normalize_by_wh=False).numpy(), g2["t_enc_nowh"])
    assert np.array_equal(tblr2bboxes(pri, pred, normalizer=nm, max_shape=(300, 350, 3)).numpy(), g2["t_dec4"])
    assert np.array_equal(tblr2bboxes(pri, pred * 20, normalizer=4.0, normalize_by_wh=False).numpy(), g2["t_dec_nowh"])
    assert np.array_equal(tblr2bboxes(pri, pred, normalizer=1 / 8, max_shape=(300, 350, 3), clip_border=False).numpy(),
                          g2["t_dec_noclip"])
    # round trip at size: decode(encode(gt)) == gt up to fp32 rounding, for 200k boxes
    gen = torch.Generator().manual_seed(3)
    xy = torch.rand(200000, 2, generator=gen) * 500
    pri = torch.cat([xy, xy + torch.rand(200000, 2, generator=gen) * 100 + 8], 1).cuda()
    gt = torch.cat([xy - 20, xy + 90], 1).cuda()
    back = coder.decode(pri, coder.encode(pri, gt))
    assert float((back - gt).abs().max()) < 2e-3
    assert tuple(coder.encode(torch.empty(0, 4), torch.empty(0, 4)).shape) == (0, 4)


# ------------------------------------------------------------------------------------------------ losses
def close(a, b, rtol=TOL, atol=1e-7):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.allclose(a, b, rtol=rtol, atol=atol)


def run_loss(mod, x, *args, **kw):
    x = x.clone().requires_grad_(True)
    loss = mod(x, *args, **kw)
    (loss.sum() if loss.dim() else loss).backward()
    return loss.detach(), x.grad


def test_focal_loss_module_vs_reference(golden):
    from radet_amd.models import build_loss
    g = golden("ops")
    fl1 = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0))
    elem = fl1(T(g["logits"]), T(g["labels"]), reduction_override="none")
    assert close(elem, g["focal"])                                      # py_sigmoid_focal_loss of the reference
    g2 = golden("ops2")
    x, lab, wr, we = T(g2["f_logits"]), T(g2["f_labels"]), T(g2["f_w_row"]), T(g2["f_w_el"])
    fl = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.5))
    for tag, kw in (("f_mean", {}), ("f_sum_wrow", dict(weight=wr, reduction_override="sum")),
                    ("f_avg_wel", dict(weight=we.reshape(-1), avg_factor=37.5)),
                    ("f_none_wrow", dict(weight=wr, reduction_override="none"))):
        loss, grad = run_loss(fl, x, lab, **kw)
        assert close(loss, g2[tag]), tag
        assert close(grad, g2[tag + "_g"], atol=1e-8), tag
    fl3 = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=1.5, alpha=0.4, loss_weight=1.0))
    loss, grad = run_loss(fl3, x.cuda(), lab.cuda(), weight=wr.cuda(), avg_factor=torch.tensor(12.0))
    assert loss.is_cuda and close(loss, g2["f_g15"]) and close(grad, g2["f_g15_g"], atol=1e-8)
    with pytest.raises(ValueError):
        fl(x, lab, avg_factor=3.0, reduction_override="sum")


def test_giou_loss_module_vs_reference(golden):
    from radet_amd.models import build_loss
    g2 = golden("ops2")
    gl = build_loss(dict(type="GIoULoss", loss_weight=2.0))
    p, t, w = T(g2["g_pred"]), T(g2["g_tgt"]), T(g2["g_w"])
    for tag, kw in (("g_mean", {}), ("g_avg_w", dict(weight=w, avg_factor=w.sum())),
                    ("g_none_w", dict(weight=w, reduction_override="none")), ("g_sum", dict(reduction_override="sum"))):
        loss, grad = run_loss(gl, p, t, **kw)
        assert close(loss, g2[tag]), tag
        assert close(grad, g2[tag + "_g"], rtol=2e-4, atol=1e-7), tag    # incl. identical / disjoint boxes (tie branches)
    z = gl(p, t, weight=torch.zeros(90))                                # iou_loss.py:335-336: all-zero weights -> 0
    assert float(z) == 0.0


def test_cross_entropy_sigmoid_module_vs_reference(golden):
    from radet_amd.models import build_loss
    g2 = golden("ops2")
    ce = build_loss(dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))
    x, t, w = T(g2["c_x"]), T(g2["c_t"]), T(g2["g_w"])
    for tag, kw in (("c_avg_w", dict(weight=w, avg_factor=w.sum())), ("c_mean", {}),
                    ("c_none", dict(weight=w, reduction_override="none"))):
        loss, grad = run_loss(ce, x, t, **kw)
        assert close(loss, g2[tag]), tag
        assert close(grad, g2[tag + "_g"], atol=1e-8), tag
    loss, grad = run_loss(ce, T(g2["c_x2"]), T(g2["f_labels"]), weight=T(g2["f_w_row"]), avg_factor=20.0)
    assert close(loss, g2["c_onehot"]) and close(grad, g2["c_onehot_g"], atol=1e-8)
    with pytest.raises(NotImplementedError):
        build_loss(dict(type="CrossEntropyLoss", use_sigmoid=False))(T(g2["c_x2"]), T(g2["f_labels"]))


def test_loss_reductions_deterministic_at_size():
    """1.3 M-element focal loss (R = 64000 rows x 21 classes, > 1024 workgroup partials): two runs are bit-identical
    and the sum equals the sum of the unreduced elements."""
    from radet_amd.models import build_loss
    fl = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0))
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(64000, 21, generator=gen) * 2).cuda()
    lab = torch.randint(0, 22, (64000,), generator=gen).cuda()
    a = fl(x, lab, reduction_override="sum")
    b = fl(x, lab, reduction_override="sum")
    assert torch.equal(a, b)
    elem = fl(x, lab, reduction_override="none")
    assert abs(float(a) - float(elem.double().sum())) <= 1e-5 * float(a)


# ------------------------------------------------------------------------------------------------ multiclass_nms
def test_multiclass_nms_vs_oracle():
    from oracle import nms as onms
    from radet_amd.core import multiclass_nms
    gen = torch.Generator().manual_seed(5)
    n, C = 400, 6
    xy = torch.rand(n, 2, generator=gen) * 300
    boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=gen) * 80 + 10], 1)
    scores = torch.rand(n, C + 1, generator=gen) * 0.5
    fac = torch.rand(n, generator=gen)
    dets, labels, inds = multiclass_nms(boxes, scores, 0.1, dict(type="nms", iou_threshold=0.5), max_num=50,
                                        score_factors=fac, return_inds=True)
    # restatement of bbox_nms.py:38-79 on the host with the oracle's batched NMS
    s = (scores[:, :-1] * fac[:, None]).reshape(-1)
    bb = boxes[:, None].expand(n, C, 4).reshape(-1, 4)
    lab = torch.arange(C).view(1, -1).expand(n, C).reshape(-1)
    sel = (s > 0.1).nonzero().squeeze(1)
    od, ok = onms.batched_nms(bb[sel].numpy(), s[sel].numpy(), lab[sel].numpy(), 0.5)
    assert np.array_equal(inds.numpy(), ok[:50]) and np.array_equal(dets.numpy(), od[:50])
    assert np.array_equal(labels.numpy(), lab[sel].numpy()[ok[:50]])
    # per-class boxes [n, C*4] and nothing above the threshold
    d2, l2 = multiclass_nms(boxes.repeat(1, C), scores, 0.1, dict(type="nms", iou_threshold=0.5), score_factors=fac)
    assert np.array_equal(d2.numpy()[:50], od[:50])
    d3, l3 = multiclass_nms(boxes, scores, 0.99, dict(type="nms", iou_threshold=0.5))
    assert tuple(d3.shape) == (0, 4) and tuple(l3.shape) == (0,)


# ------------------------------------------------------------------------------------------------ RADetHead methods
def synth_head_outputs(seed, B, cls_mean=-2.0):
    g = torch.Generator().manual_seed(seed)
    cls, reg, iou = [], [], []
    for (h, w) in LEVEL_HW:
        cls.append(torch.randn(B, 21, h, w, generator=g) * 1.5 + cls_mean)
        reg.append(torch.relu(torch.randn(B, 4, h, w, generator=g) * 2.0 + 2.5))
        iou.append(torch.randn(B, 1, h, w, generator=g))
    return cls, reg, iou


def flat(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


@pytest.fixture(scope="module")
def det():
    from oracle import synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    d = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(d.state_dict(), seed=0)
    return d.cuda().train()


def targets(golden, tags=("g8", "g3")):
    a = golden("assigner")
    return ([T(a[t + "_boxes"]) for t in tags], [T(a[t + "_labels"]) for t in tags],
            [T(a[t + "_p2g"].astype(np.int64)) for t in tags], [T(a[t + "_w"]) for t in tags])


def test_head_loss_method_vs_reference_golden(det, golden):
    """det.bbox_head.loss(...) with the reference's argument lists == the reference's RADetHead.loss: the loss triple
    and the gradients w.r.t. the NCHW inputs (incl. bbox_preds that are exactly 0 after the ReLU)."""
    from oracle import synth
    g = golden("head_loss")
    cls, reg, iou = synth_head_outputs(7, 2)
    for t in cls + reg + iou:
        t.requires_grad_(True)
    gt_b, gt_l, p2g, pw = targets(golden)
    losses = det.bbox_head.loss(cls, reg, iou, gt_b, gt_l, p2g, pw, synth.img_metas(2))
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(float(losses[k]) - float(g[k])) <= TOL * max(1.0, abs(float(g[k]))), (k, float(losses[k]), float(g[k]))
    sum(losses.values()).backward()
    pos = T(g["pos"])
    g_cls, g_reg, g_iou = flat([t.grad for t in cls]), flat([t.grad for t in reg]), flat([t.grad for t in iou])
    assert np.allclose(g_cls[::7].numpy(), g["g_cls_rows"], rtol=TOL, atol=1e-9)
    assert np.allclose(g_cls[pos].numpy(), g["g_cls_pos"], rtol=TOL, atol=1e-9)
    assert np.allclose(g_reg[pos].numpy(), g["g_reg_pos"], rtol=TOL, atol=1e-8)          # every entry, zeros included
    assert (flat(reg)[pos] == 0).any()
    assert np.isclose(g_reg.double().abs().sum().item(), float(g["g_reg_abs"]), rtol=TOL)
    assert np.allclose(g_iou[pos].numpy(), g["g_iou_pos"], rtol=TOL, atol=1e-9)
    assert np.isclose(g_cls.double().abs().sum().item(), float(g["g_cls_abs"]), rtol=TOL)
    # empty-gt batch: the num_pos == 0 branch
    cls0, reg0, iou0 = synth_head_outputs(8, 2)
    l0 = det.bbox_head.loss(cls0, reg0, iou0, [torch.zeros(0, 4)] * 2, [torch.zeros(0, dtype=torch.long)] * 2,
                            [torch.full((6400,), -1, dtype=torch.long)] * 2, [torch.ones(6400)] * 2, synth.img_metas(2))
    assert abs(float(l0["loss_cls"]) - float(g["e_loss_cls"])) <= TOL * float(g["e_loss_cls"])
    assert float(l0["loss_bbox"]) == 0.0 and float(l0["loss_iou"]) == 0.0


def test_head_get_bboxes_method_vs_reference_golden(det, golden):
    from oracle import synth
    g = golden("get_bboxes")
    cls, reg, iou = synth_head_outputs(9, 2, cls_mean=-4.0)
    metas = synth.img_metas(2)
    for name in ("vote", "global_vote"):
        cfg = dict(det.test_cfg)
        cfg["nms"] = dict(cfg["nms"], type=name)
        res = det.bbox_head.get_bboxes(cls, reg, iou, metas, cfg=cfg, rescale=True)
        assert len(res) == 2
        for i, (db, dl) in enumerate(res):
            ref_b, ref_l = g[f"{name}_{i}_b"], g[f"{name}_{i}_l"]
            assert tuple(db.shape) == ref_b.shape and np.array_equal(dl.cpu().numpy(), ref_l)
            assert np.allclose(db[:, :4].cpu().numpy(), ref_b[:, :4], rtol=TOL, atol=1e-3)
            assert np.allclose(db[:, 4].cpu().numpy(), ref_b[:, 4], rtol=TOL, atol=1e-7)
    res = det.bbox_head.get_bboxes([c.cuda() for c in cls], [r.cuda() for r in reg], [i.cuda() for i in iou], metas)   # cfg=None -> test_cfg
    assert len(res) == 2 and res[0][0].shape[1] == 5
    with pytest.raises(NotImplementedError):
        det.bbox_head.get_bboxes(cls, reg, iou, metas, with_nms=False)


def test_head_forward_train_method(det, golden):
    """bbox_head.forward_train(extract_feat(img), ...) == the detector's forward_train: same losses, and after
    backward the same head-parameter gradients (golden gradient norms of the reference) plus feature gradients."""
    from oracle import synth
    gm = golden("model")
    img = synth.synth_images(0, 2).cuda()
    gt_b, gt_l, p2g, pw = targets(golden)
    metas = synth.img_metas(2)
    feats = [f.requires_grad_(True) for f in det.extract_feat(img)]
    det.zero_grad()
    losses = det.bbox_head.forward_train(feats, metas, gt_b, gt_l, p2g, pw)
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(float(losses[k]) - float(gm[k])) <= TOL * max(1.0, abs(float(gm[k]))), k
    sum(losses.values()).backward()
    ref = dict(zip([str(n) for n in gm["grad_names"]], gm["grad_norms"]))
    n_checked = 0
    for n, p in det.named_parameters():
        if n.startswith("bbox_head.") and p.requires_grad:
            a, b = p.grad.double().norm().item(), ref[n]
            assert abs(a - b) <= 5e-4 * max(b, 1e-3), (n, a, b)
            n_checked += 1
    assert n_checked >= 30
    assert all(f.grad is not None and torch.isfinite(f.grad).all() and float(f.grad.abs().sum()) > 0 for f in feats)
    # the feature gradients are the ones the full backward feeds into the neck: compare with the engine's dP
    e = det.runtime().engine
    dP = e.buf["dP"].float()
    r0, r1 = e.plv.level_rows(0)
    assert torch.allclose(feats[0].grad.permute(0, 2, 3, 1).reshape(-1, 256), dP[r0:r1], rtol=0, atol=0)
    losses2, props = det.bbox_head.forward_train(feats, metas, gt_b, gt_l, p2g, pw, proposal_cfg=det.test_cfg)
    assert len(props) == 2 and props[0][0].shape[1] == 5


def test_losses_inside_head_are_callable(det):
    """the head's loss modules are the registered, callable classes (reference: self.loss_cls(...) inside loss())"""
    x = torch.randn(50, 21)
    lab = torch.randint(0, 22, (50,))
    v = det.bbox_head.loss_cls(x, lab, weight=torch.ones(50), avg_factor=7.0)
    assert v.dim() == 0 and float(v) > 0
    b = torch.tensor([[0., 0., 10., 10.]])
    assert abs(float(det.bbox_head.loss_bbox(b, b)) - 0.0) < 1e-6
    assert float(det.bbox_head.loss_iou(torch.zeros(4), torch.full((4,), 0.5))) == pytest.approx(np.log(2.0), rel=1e-6)
