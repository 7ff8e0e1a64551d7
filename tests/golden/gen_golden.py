"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) in this container.

    python tests/golden/gen_golden.py            # writes the .npz files next to this script

Only data (inputs + the reference's outputs) is written; no reference source travels.
Inputs come from oracle/synth.py (our own seeded generators) so that the tests can
rebuild the large ones (full-model weights, images) bit-identically on any box.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_import  # noqa: E402
from oracle import synth  # noqa: E402

torch.set_num_threads(8)
ref_import.install()

from radet.models import build_detector  # noqa: E402
from radet.core.anchor import build_anchor_generator  # noqa: E402
from radet.core.bbox import build_bbox_coder  # noqa: E402
from radet.core import bbox_overlaps  # noqa: E402
from radet.core.mask.structures import BitmapMasks  # noqa: E402
from radet.datasets.pipelines.label_assignment import LabelAssignment  # noqa: E402
from radet.models.losses.focal_loss import py_sigmoid_focal_loss  # noqa: E402
from radet.ops import vote_nms, global_vote_nms, cluster_nms  # noqa: E402

ANCHOR_CFG = dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                  strides=[8, 16, 32, 64, 128])
STRIDES = (8, 16, 32, 64, 128)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------- anchors
def gen_anchors():
    ag = build_anchor_generator(ANCHOR_CFG)
    out = {}
    for tag, (h, w) in dict(a480x640=(480, 640), a800x800=(800, 800)).items():
        sizes = [(int(np.ceil(h / s)), int(np.ceil(w / s))) for s in STRIDES]
        anchors = ag.grid_anchors(sizes, device="cpu")
        out[tag] = torch.cat(anchors).numpy()
        out[tag + "_sizes"] = np.asarray(sizes, np.int64)
    save("anchors", **out)


# ----------------------------------------------------------------------------- assigner
ASSIGN_CASES = [
    # (tag, synth seed, G, tiny_visible, np seed)
    ("g0", 1, 0, False, 5),
    ("g1", 2, 1, False, 6),
    ("g8", 0, 8, False, 123),
    ("g8b", 7, 8, True, 124),
    ("g20", 3, 20, False, 7),
    ("g3", 11, 3, False, 125),
]


class _Maps:
    """what LabelAssignment.__call__ needs of `distance_maps` (label_assignment.py:152): float maps instead of BitmapMasks"""

    def __init__(self, a):
        self.a = a

    def to_ndarray(self):
        return self.a


def run_ref_assigner(boxes, labels, masks, np_seed, graded=False, words=False, **opts):
    kw = dict(adapt_positive_num=False, balance_sample=True)
    kw.update(opts)
    la = LabelAssignment(anchor_generator_cfg=ANCHOR_CFG, neg_threshold=0.2, positive_num=10, **kw)
    np.random.seed(np_seed)
    res = dict(img_shape=(480, 640, 3), gt_bboxes=boxes, gt_labels=labels,
               distance_maps=_Maps(synth.graded_maps(masks)) if graded else BitmapMasks([m for m in masks], 480, 640))
    st0 = np.random.get_state()
    res = la(res)
    # how much of the stream was consumed: replay it until the state matches -- in uniforms (two raw 32-bit outputs each; the
    # weighted draws) or, for the integer draws of random_sample_by_distance=False, in raw outputs (`words=True`)
    probe = np.random.RandomState()
    probe.set_state(st0)
    st1 = np.random.get_state()
    used = 0
    while not (probe.get_state()[2] == st1[2] and np.array_equal(probe.get_state()[1], st1[1])):
        if words:
            probe._bit_generator.random_raw(1)
        else:
            probe.random_sample()
        used += 1
        assert used < 2000000
    return res["points_to_gt_index"], res["points_weight"], used


def gen_assigner():
    out = {}
    for tag, sseed, G, tiny, npseed in ASSIGN_CASES:
        boxes, labels, masks = synth.synth_objects(sseed, G, tiny_visible=tiny)
        if G == 0:
            p2g = np.full(6400, -1, np.int64)
            w = np.ones(6400, np.float32)
            used = 0
            # the reference crashes on an empty gt list inside BitmapMasks/np.stack; the
            # documented behaviour (no gts => all negative, weight 1) is what train uses.
        else:
            p2g, w, used = run_ref_assigner(boxes, labels, masks, npseed)
        out[tag + "_boxes"] = boxes
        out[tag + "_labels"] = labels
        out[tag + "_masks"] = np.packbits(masks.reshape(G, -1), axis=1) if G else np.zeros((0, 38400), np.uint8)
        out[tag + "_npseed"] = np.int64(npseed)
        out[tag + "_p2g"] = p2g.astype(np.int16)
        out[tag + "_w"] = w.astype(np.float32)
        out[tag + "_used"] = np.int64(used)
        print(tag, "pos", int((p2g > 0).sum()), "ign", int((p2g == 0).sum()), "sumw", float(w[p2g > 0].sum()),
              "uniforms", used)
    save("assigner", **out)
    return out


# the constructor options no BOP config sets (label_assignment.py:30-46): outputs of the reference for each of them and all
ASSIGN_OPT_CASES = [
    # (tag, synth seed, G, tiny_visible, np seed, options)
    ("nobal", 7, 8, True, 31, dict(balance_sample=False)),
    ("mulpro", 0, 8, False, 32, dict(multiply_samplepro_for_weight=True)),
    ("mulpro_tiny", 7, 8, True, 33, dict(multiply_samplepro_for_weight=True)),
    ("adapt", 0, 8, False, 34, dict(adapt_positive_num=True)),
    ("adapt20", 3, 20, False, 35, dict(adapt_positive_num=True)),
    ("adapt_nobal", 7, 8, True, 36, dict(adapt_positive_num=True, balance_sample=False)),
    ("all3", 11, 3, False, 37, dict(adapt_positive_num=True, balance_sample=False, multiply_samplepro_for_weight=True)),
    # float maps with graded values (oracle/synth.py::graded_maps): the weights really carry the map value
    ("g_mulpro", 0, 8, False, 38, dict(graded=True, multiply_samplepro_for_weight=True)),
    ("g_all3", 7, 8, True, 39, dict(graded=True, adapt_positive_num=True, balance_sample=False, multiply_samplepro_for_weight=True)),
    ("g_plain", 3, 20, False, 40, dict(graded=True)),
    # random_sample_by_distance=False: np.random.choice without p (randint / permutation inside numpy)
    ("unif", 0, 8, False, 41, dict(random_sample_by_distance=False)),
    ("unif_tiny", 7, 8, True, 42, dict(random_sample_by_distance=False)),
    ("unif_nobal", 7, 8, True, 43, dict(random_sample_by_distance=False, balance_sample=False)),
    ("unif20_all", 3, 20, False, 44, dict(graded=True, random_sample_by_distance=False, adapt_positive_num=True,
                                          multiply_samplepro_for_weight=True)),
]


def gen_assigner_opts():
    out = {}
    for tag, sseed, G, tiny, npseed, opts in ASSIGN_OPT_CASES:
        boxes, labels, masks = synth.synth_objects(sseed, G, tiny_visible=tiny)
        p2g, w, used = run_ref_assigner(boxes, labels, masks, npseed, words=True, **opts)
        out[tag + "_synth"] = np.asarray([sseed, G, int(tiny), npseed, int(bool(opts.get("graded")))], np.int64)   # (inputs: oracle/synth.py)
        out[tag + "_flags"] = np.int64((1 if opts.get("balance_sample", True) else 0) | (2 if opts.get("multiply_samplepro_for_weight") else 0)
                                       | (4 if opts.get("adapt_positive_num") else 0) | (0 if opts.get("random_sample_by_distance", True) else 8))
        out[tag + "_p2g"] = p2g.astype(np.int16)
        out[tag + "_w"] = w.astype(np.float32)
        out[tag + "_used_words"] = np.int64(used)        # raw 32-bit outputs of the RandomState consumed
        print(tag, opts, "pos", int((p2g > 0).sum()), "ign", int((p2g == 0).sum()), "sumw", float(w[p2g > 0].sum()), "uniforms", used,
              "max w", float(w.max()))
    save("assigner_opts", **out)
    return out


# ----------------------------------------------------------------------------- small ops
def gen_ops():
    g = torch.Generator().manual_seed(42)
    coder = build_bbox_coder(dict(type="TBLRBBoxCoder", normalizer=1 / 8))
    ag = build_anchor_generator(ANCHOR_CFG)
    anchors = torch.cat(ag.grid_anchors([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], device="cpu"))
    sel = torch.randint(0, 6400, (64,), generator=g)
    pri = anchors[sel]
    xy = torch.rand(64, 2, generator=g) * torch.tensor([400.0, 300.0])
    wh = torch.rand(64, 2, generator=g) * 200 + 5
    gts = torch.cat([xy, xy + wh], 1)
    enc = coder.encode(pri, gts)
    pred = torch.rand(64, 4, generator=g) * 6
    dec = coder.decode(pri, pred)
    dec_clip = coder.decode(pri, pred, max_shape=(480, 640, 3))
    b1 = torch.cat([xy[:32], xy[:32] + wh[:32]], 1)
    b2 = torch.cat([xy[32:] * 0.8, xy[32:] * 0.8 + wh[32:]], 1)
    logits = torch.randn(96, 21, generator=g) * 2
    lab = torch.randint(0, 22, (96,), generator=g)
    t = torch.zeros(96, 21)
    fg = lab < 21
    t[fg.nonzero().reshape(-1), lab[fg]] = 1
    focal = py_sigmoid_focal_loss(logits, t, reduction="none")
    save("ops", priors=pri.numpy(), gts=gts.numpy(), enc=enc.numpy(), pred=pred.numpy(), dec=dec.numpy(),
         dec_clip=dec_clip.numpy(), b1=b1.numpy(), b2=b2.numpy(),
         iou_aligned=bbox_overlaps(b1, b2, is_aligned=True).numpy(),
         giou_aligned=bbox_overlaps(b1, b2, mode="giou", is_aligned=True, eps=1e-6).numpy(),
         iou_matrix=bbox_overlaps(b1, b2).numpy(),
         giou_matrix=bbox_overlaps(b1, b2, mode="giou").numpy(),
         logits=logits.numpy(), labels=lab.numpy(), focal=focal.numpy())



# ----------------------------------------------------------------------------- stand-alone ops, second set
def gen_ops2():
    """bbox_overlaps corner cases / iof / batch dims / a bigger M x N matrix, TBLR coder with a 4-vector normalizer,
    and the three loss MODULES (forward values under every reduction + input gradients)."""
    from radet.models.builder import build_loss
    g = torch.Generator().manual_seed(77)
    out = {}

    def rnd_boxes(n, scale=400.0):
        xy = torch.rand(n, 2, generator=g) * scale
        wh = torch.rand(n, 2, generator=g) * 150 + 1
        return torch.cat([xy, xy + wh], 1)

    # overlaps: iof, batch dims, degenerate boxes (zero area, identical, disjoint, touching), big matrix samples
    a, b = rnd_boxes(37), rnd_boxes(53)
    out.update(ov_a=a.numpy(), ov_b=b.numpy(), iof_matrix=bbox_overlaps(a, b, mode="iof").numpy(),
               iof_aligned=bbox_overlaps(a, b[:37], mode="iof", is_aligned=True).numpy())
    ab, bb = rnd_boxes(2 * 3 * 9).reshape(2, 3, 9, 4), rnd_boxes(2 * 3 * 5).reshape(2, 3, 5, 4)
    out.update(ovb_a=ab.numpy(), ovb_b=bb.numpy(), ovb_giou=bbox_overlaps(ab, bb, mode="giou").numpy(),
               ovb_iou_aligned=bbox_overlaps(ab, ab.flip(2), is_aligned=True).numpy())
    deg = torch.tensor([[10., 10., 10., 10.], [0., 0., 20., 20.], [0., 0., 20., 20.], [20., 0., 40., 20.],
                        [100., 100., 120., 130.], [5., 5., 5., 30.], [1e-4, 1e-4, 2e-4, 2e-4]])
    for mode in ("iou", "iof", "giou"):
        out["deg_" + mode] = bbox_overlaps(deg, deg, mode=mode).numpy()
        out["deg_al_" + mode] = bbox_overlaps(deg, deg.roll(1, 0), mode=mode, is_aligned=True).numpy()
    out["deg"] = deg.numpy()
    A, B = rnd_boxes(700, 600.0), rnd_boxes(1300, 600.0)
    big = bbox_overlaps(A, B, mode="giou")
    idx = torch.randint(0, big.numel(), (4096,), generator=g)
    out.update(big_a=A.numpy(), big_b=B.numpy(), big_idx=idx.numpy(), big_val=big.reshape(-1)[idx].numpy(),
               big_sum=big.double().sum().numpy(), big_iou_sum=bbox_overlaps(A, B).double().sum().numpy())

    # TBLR with per-side normalizers, normalize_by_wh on / off, clipping
    from radet.core.bbox.coder.tblr_bbox_coder import bboxes2tblr, tblr2bboxes
    pri, gts = rnd_boxes(40), rnd_boxes(40)
    nm = [0.5, 0.25, 2.0, 4.0]
    enc4 = bboxes2tblr(pri, gts, normalizer=nm)
    enc_nowh = bboxes2tblr(pri, gts, normalizer=4.0, normalize_by_wh=False)
    pred = torch.rand(40, 4, generator=g) * 3
    out.update(t_pri=pri.numpy(), t_gts=gts.numpy(), t_nm=np.asarray(nm, np.float32), t_enc4=enc4.numpy(),
               t_enc_nowh=enc_nowh.numpy(), t_pred=pred.numpy(),
               t_dec4=tblr2bboxes(pri, pred, normalizer=nm, max_shape=(300, 350, 3)).numpy(),
               t_dec_nowh=tblr2bboxes(pri, pred * 20, normalizer=4.0, normalize_by_wh=False).numpy(),
               t_dec_noclip=tblr2bboxes(pri, pred, normalizer=1 / 8, max_shape=(300, 350, 3), clip_border=False).numpy())

    # loss modules
    N, C = 150, 21
    logits = (torch.randn(N, C, generator=g) * 2).requires_grad_(True)
    lab = torch.randint(0, C + 1, (N,), generator=g)
    w_row = torch.rand(N, generator=g)
    w_el = torch.rand(N, C, generator=g)
    fl = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.5))
    out.update(f_logits=logits.detach().numpy(), f_labels=lab.numpy(), f_w_row=w_row.numpy(), f_w_el=w_el.numpy())

    def rec(tag, loss, x):
        x.grad = None
        (loss.sum() if loss.dim() else loss).backward()
        out[tag] = loss.detach().numpy()
        out[tag + "_g"] = x.grad.numpy().copy()

    rec("f_mean", fl(logits, lab), logits)
    rec("f_sum_wrow", fl(logits, lab, weight=w_row, reduction_override="sum"), logits)
    rec("f_avg_wel", fl(logits, lab, weight=w_el.reshape(-1), avg_factor=37.5), logits)
    rec("f_none_wrow", fl(logits, lab, weight=w_row, reduction_override="none"), logits)
    fl3 = build_loss(dict(type="FocalLoss", use_sigmoid=True, gamma=1.5, alpha=0.4, loss_weight=1.0))
    rec("f_g15", fl3(logits, lab, weight=w_row, avg_factor=torch.tensor(12.0)), logits)

    M = 90
    pb = rnd_boxes(M).requires_grad_(True)
    tb = (pb.detach() + torch.randn(M, 4, generator=g) * 15)
    tb = torch.cat([torch.min(tb[:, :2], tb[:, 2:] - 1), tb[:, 2:]], 1)
    pb.data[:5] = tb[:5]                       # identical boxes: the max / min tie branches of the gradient
    pb.data[5:8, :2] = tb[5:8, 2:] + 30        # disjoint
    pb.data[5:8, 2:] = tb[5:8, 2:] + 60
    gw = torch.rand(M, generator=g)
    gl = build_loss(dict(type="GIoULoss", loss_weight=2.0))
    out.update(g_pred=pb.detach().numpy(), g_tgt=tb.numpy(), g_w=gw.numpy())
    rec("g_mean", gl(pb, tb), pb)
    rec("g_avg_w", gl(pb, tb, weight=gw, avg_factor=gw.sum()), pb)
    rec("g_none_w", gl(pb, tb, weight=gw, reduction_override="none"), pb)
    rec("g_sum", gl(pb, tb, reduction_override="sum"), pb)

    ce = build_loss(dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))
    x1 = (torch.randn(M, generator=g) * 2).requires_grad_(True)
    t1 = torch.rand(M, generator=g)
    out.update(c_x=x1.detach().numpy(), c_t=t1.numpy())
    rec("c_avg_w", ce(x1, t1, weight=gw, avg_factor=gw.sum()), x1)
    rec("c_mean", ce(x1, t1), x1)
    rec("c_none", ce(x1, t1, weight=gw, reduction_override="none"), x1)
    x2 = (torch.randn(N, C, generator=g)).requires_grad_(True)      # class-index labels -> one-hot expansion
    out.update(c_x2=x2.detach().numpy())
    rec("c_onehot", ce(x2, lab, weight=w_row, avg_factor=20.0), x2)
    save("ops2", **out)


# ----------------------------------------------------------------------------- raw NMS ops
def synth_nms_boxes(seed, n_base, per, n_labels=21):
    g = torch.Generator().manual_seed(seed)
    base_xy = torch.rand(n_base, 2, generator=g) * torch.tensor([500.0, 350.0])
    base_wh = torch.rand(n_base, 2, generator=g) * 120 + 20
    base = torch.cat([base_xy, base_xy + base_wh], 1)
    base_lab = torch.randint(0, n_labels, (n_base,), generator=g)
    boxes = base.repeat_interleave(per, 0) + torch.randn(n_base * per, 4, generator=g) * 4.0
    labels = base_lab.repeat_interleave(per, 0)
    cls = torch.rand(n_base * per, generator=g) * 0.9 + 0.05
    ctr = torch.rand(n_base * per, generator=g) * 0.9 + 0.05
    perm = torch.randperm(n_base * per, generator=g)
    return boxes[perm].contiguous(), cls[perm].contiguous(), ctr[perm].contiguous(), labels[perm].contiguous()


def nms_margin(boxes, labels, thr):
    iou = bbox_overlaps(boxes, boxes)
    same = labels[:, None] == labels[None, :]
    d = (iou - thr).abs()[same]
    return float(d.min())


def gen_nms(test_cfg):
    out = {}
    for tag, seed, nb, per in [("a", 100, 60, 25), ("b", 101, 300, 5), ("c", 102, 8, 3)]:
        while True:   # tie-free scores and no IoU within 2e-6 of the threshold (knife-edge guard)
            boxes, cls, ctr, labels = synth_nms_boxes(seed, nb, per)
            score = cls * ctr
            if score.unique().numel() == score.numel() and nms_margin(boxes, labels, 0.65) > 2e-6:
                break
            seed += 1000
        out[f"{tag}_seed"] = np.int64(seed)
        vb, vl = vote_nms(boxes, cls, labels, test_cfg.nms, score_factor=ctr, max_num=0)
        gb, gl = global_vote_nms(boxes, cls, labels, test_cfg.nms, score_factor=ctr, max_num=0)
        ids, num = cluster_nms(boxes, score, labels, 0.65)
        out.update({f"{tag}_boxes": boxes.numpy(), f"{tag}_cls": cls.numpy(), f"{tag}_ctr": ctr.numpy(),
                    f"{tag}_labels": labels.numpy(), f"{tag}_vote_b": vb.numpy(), f"{tag}_vote_l": vl.numpy(),
                    f"{tag}_gvote_b": gb.numpy(), f"{tag}_gvote_l": gl.numpy(),
                    f"{tag}_cl_ids": ids.numpy(), f"{tag}_cl_num": num.numpy()})
        print("nms", tag, boxes.shape[0], "->", vb.shape[0], gb.shape[0])
    save("nms", **out)


# ----------------------------------------------------------------------------- head loss / decode
LEVEL_HW = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]


def synth_head_outputs(seed, B, cls_mean=-2.0):
    """Flat, level-major head outputs -> lists of NCHW tensors (what the reference consumes)."""
    g = torch.Generator().manual_seed(seed)
    cls, reg, iou = [], [], []
    for (h, w) in LEVEL_HW:
        cls.append(torch.randn(B, 21, h, w, generator=g) * 1.5 + cls_mean)
        reg.append(torch.relu(torch.randn(B, 4, h, w, generator=g) * 2.0 + 2.5))
        iou.append(torch.randn(B, 1, h, w, generator=g))
    return cls, reg, iou


def flat(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


def gen_head(det, assign):
    head = det.bbox_head
    B = 2
    tags = ["g8", "g3"]
    gt_b = [torch.from_numpy(assign[t + "_boxes"]) for t in tags]
    gt_l = [torch.from_numpy(assign[t + "_labels"]) for t in tags]
    p2g = [torch.from_numpy(assign[t + "_p2g"].astype(np.int64)) for t in tags]
    pw = [torch.from_numpy(assign[t + "_w"]) for t in tags]
    cls, reg, iou = synth_head_outputs(7, B)
    for t in cls + reg + iou:
        t.requires_grad_(True)
    metas = synth.img_metas(B)
    losses = head.loss(cls, reg, iou, gt_b, gt_l, p2g, pw, metas)
    total = sum(losses.values())
    total.backward()
    # targets, flattened exactly like loss() does
    anchor_list, _ = head.get_anchors([t.shape[-2:] for t in cls], metas, device="cpu")
    labels, bbox_t, weights, _ = head.get_targets(anchors_list=anchor_list, bbox_preds=reg, cls_scores=cls,
                                                 gt_bboxes_list=gt_b, gt_labels_list=gt_l,
                                                 points_to_gt_index_list=p2g, points_weight_list=pw,
                                                 image_metas=metas)
    labels, bbox_t, weights = torch.cat(labels), torch.cat(bbox_t), torch.cat(weights)
    pos = ((labels >= 0) & (labels < 21)).nonzero().reshape(-1)
    g_cls, g_reg, g_iou = flat([t.grad for t in cls]), flat([t.grad for t in reg]), flat([t.grad for t in iou])
    out = dict(loss_cls=losses["loss_cls"].detach().numpy(), loss_bbox=losses["loss_bbox"].detach().numpy(),
               loss_iou=losses["loss_iou"].detach().numpy(),
               labels=labels.numpy().astype(np.int16), bbox_targets_pos=bbox_t[pos].numpy(),
               weights=weights.numpy(), pos=pos.numpy(),
               g_cls_rows=g_cls[::7].numpy(), g_cls_sum=g_cls.double().sum().numpy(),
               g_cls_abs=g_cls.double().abs().sum().numpy(),
               g_cls_pos=g_cls[pos].numpy(),
               g_reg_pos=g_reg[pos].numpy(), g_reg_abs=g_reg.double().abs().sum().numpy(),
               g_iou_pos=g_iou[pos].numpy(), g_iou_abs=g_iou.double().abs().sum().numpy())
    # empty-gt image pair: num_pos == 0 branch
    cls0, reg0, iou0 = synth_head_outputs(8, B)
    for t in cls0 + reg0 + iou0:
        t.requires_grad_(True)
    e_b = [torch.zeros(0, 4), torch.zeros(0, 4)]
    e_l = [torch.zeros(0, dtype=torch.long)] * 2
    e_p = [torch.full((6400,), -1, dtype=torch.long)] * 2
    e_w = [torch.ones(6400)] * 2
    l0 = head.loss(cls0, reg0, iou0, e_b, e_l, e_p, e_w, metas)
    sum(l0.values()).backward()
    out.update(e_loss_cls=l0["loss_cls"].detach().numpy(), e_loss_bbox=l0["loss_bbox"].detach().numpy(),
               e_loss_iou=l0["loss_iou"].detach().numpy(),
               e_g_cls_abs=flat([t.grad for t in cls0]).double().abs().sum().numpy(),
               e_g_reg_abs=flat([t.grad for t in reg0]).double().abs().sum().numpy())
    save("head_loss", **out)

    # decode + NMS (inference) on synthetic head outputs with a healthy number of candidates
    res = {}
    for nms_type in ["vote", "global_vote"]:
        cfg = ref_import.attrify(dict(head.test_cfg))
        cfg["nms"] = ref_import.attrify(dict(cfg["nms"], type=nms_type))
        with torch.no_grad():
            cls1, reg1, iou1 = synth_head_outputs(9, B, cls_mean=-4.0)
            dets = head.get_bboxes(cls1, reg1, iou1, metas, cfg=cfg, rescale=True)
        for i, (db, dl) in enumerate(dets):
            res[f"{nms_type}_{i}_b"] = db.numpy()
            res[f"{nms_type}_{i}_l"] = dl.numpy()
            print("get_bboxes", nms_type, i, db.shape)
    # candidate statistics so the tests know the case is non-trivial
    sc = flat(cls1).sigmoid()
    res["n_candidates"] = np.int64((sc > 0.05).sum().item())
    save("get_bboxes", **res)


# ----------------------------------------------------------------------------- full model
def gen_model(det, assign):
    synth.fill_state_dict(det.state_dict(), seed=0)
    det.train()
    B = 2
    img = synth.synth_images(0, B)
    tags = ["g8", "g3"]
    gt_b = [torch.from_numpy(assign[t + "_boxes"]) for t in tags]
    gt_l = [torch.from_numpy(assign[t + "_labels"]) for t in tags]
    p2g = [torch.from_numpy(assign[t + "_p2g"].astype(np.int64)) for t in tags]
    pw = [torch.from_numpy(assign[t + "_w"]) for t in tags]
    metas = synth.img_metas(B)
    feats = det.extract_feat(img)
    outs = det.bbox_head(feats)
    out = {}
    g = torch.Generator().manual_seed(5)
    c_feats = det.backbone(img)
    for i, f in enumerate(c_feats):
        idx = torch.randint(0, f.numel(), (256,), generator=g)
        out[f"c{i + 2}_idx"] = idx.numpy()
        out[f"c{i + 2}_val"] = f.detach().reshape(-1)[idx].numpy()
        out[f"c{i + 2}_absmean"] = f.detach().double().abs().mean().numpy()
    for i, f in enumerate(feats):
        idx = torch.randint(0, f.numel(), (256,), generator=g)
        out[f"p{i + 3}_idx"] = idx.numpy()
        out[f"p{i + 3}_val"] = f.detach().reshape(-1)[idx].numpy()
        out[f"p{i + 3}_absmean"] = f.detach().double().abs().mean().numpy()
    for nm, ts in zip(["cls", "reg", "iou"], outs):
        fl = flat(ts).detach()
        idx = torch.randint(0, fl.numel(), (512,), generator=g)
        out[f"{nm}_idx"] = idx.numpy()
        out[f"{nm}_val"] = fl.reshape(-1)[idx].numpy()
        out[f"{nm}_absmean"] = fl.double().abs().mean().numpy()
    det.zero_grad()
    losses = det(img=img, img_metas=metas, return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    names, norms = [], []
    for n, p in det.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(p.grad.double().norm().item())
    out.update(loss_cls=losses["loss_cls"].detach().numpy(), loss_bbox=losses["loss_bbox"].detach().numpy(),
               loss_iou=losses["loss_iou"].detach().numpy(), loss=loss.detach().numpy(),
               grad_names=np.asarray(names), grad_norms=np.asarray(norms, np.float64),
               total_grad_norm=np.float64(np.sqrt((np.asarray(norms) ** 2).sum())),
               n_params=np.int64(sum(p.numel() for p in det.parameters())),
               n_trainable=np.int64(sum(p.numel() for p in det.parameters() if p.requires_grad)))
    print("model loss", {k: float(v) for k, v in losses.items()}, "gradnorm", out["total_grad_norm"])
    # inference
    det.eval()
    with torch.no_grad():
        results = det(img=[img], img_metas=[metas], return_loss=False, rescale=True)
    for i, per_cls in enumerate(results):
        dets = np.concatenate([np.concatenate([d, np.full((d.shape[0], 1), c, np.float32)], 1)
                               for c, d in enumerate(per_cls)], 0)
        out[f"det_{i}"] = dets
        print("simple_test", i, dets.shape)
    save("model", **out)


def gen_recall():
    """eval_recalls (radet/core/evaluation/recall.py) on seeded boxes: ragged images (no ground truth, no proposals, more
    ground truths than proposals), tied scores and exact-threshold IoUs."""
    import types
    from radet.core.evaluation import recall as ref_recall
    from radet.core.evaluation.recall import eval_recalls

    # the reference builds a ragged array with np.array(list of matrices), which NumPy >= 1.24 only does with dtype=object
    # (older NumPy did it implicitly): run it with that one call made explicit
    class _Np(types.ModuleType):
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def array(x, *a, **kw):
            try:
                return np.array(x, *a, **kw)
            except ValueError:
                out = np.empty(len(x), dtype=object)
                for i, v in enumerate(x):
                    out[i] = v
                return out
    ref_recall.np = _Np("np_compat")
    ref_recall.print_recall_summary = lambda *a, **k: None        # (the table printer needs terminaltables; not data)
    rng = np.random.RandomState(7)
    gts, props = [], []
    for i in range(12):
        g = rng.randint(0, 7) if i not in (3, 9) else (0 if i == 3 else 5)
        k = rng.randint(0, 40) if i not in (5, 9) else (0 if i == 5 else 2)
        xy = rng.uniform(0, 200, (g, 2))
        gt = np.concatenate([xy, xy + rng.uniform(10, 120, (g, 2))], 1).astype(np.float32)
        pxy = rng.uniform(0, 200, (k, 2))
        pr = np.concatenate([pxy, pxy + rng.uniform(10, 120, (k, 2)), rng.uniform(0, 1, (k, 1))], 1).astype(np.float32)
        for j in range(min(g, k) // 2):                     # some proposals are jittered copies of ground truths
            pr[j, :4] = gt[j] + rng.uniform(-6, 6, 4).astype(np.float32)
        if k > 3:
            pr[1, 4] = pr[2, 4]                             # tied scores
        if g and k:
            pr[-1, :4] = gt[-1] * np.float32(1.0)           # IoU exactly 1
        gts.append(gt)
        props.append(pr)
    gts[0] = np.array([[0, 0, 10, 10]], np.float32)
    props[0] = np.array([[0, 0, 10, 5, 0.9], [0, 0, 10, 7.5, 0.8]], np.float32)      # IoU exactly 0.5 and 0.75
    nums = np.array([1, 5, 20, 100])
    thrs = np.linspace(0.5, 0.95, 10)
    rec = eval_recalls(gts, props, nums, thrs, logger="silent")
    out = dict(nums=nums, thrs=thrs, recalls=rec, n=np.int64(len(gts)),
               recalls_single=eval_recalls(gts, props, 10, 0.5, logger="silent"))
    for i, (g, p) in enumerate(zip(gts, props)):
        out[f"gt{i}"], out[f"pr{i}"] = g, p
    save("recall", **out)


def gen_model_grads(det, assign, out_name="model_grads"):
    """Sampled gradient ELEMENTS of every trainable parameter (model.npz holds only their norms, which a tap transposition
    or a sign error inside a tensor would preserve): same weights / batch as gen_model."""
    B = 2
    img = synth.synth_images(0, B)
    tags = ["g8", "g3"]
    gt_b = [torch.from_numpy(assign[t + "_boxes"]) for t in tags]
    gt_l = [torch.from_numpy(assign[t + "_labels"]) for t in tags]
    p2g = [torch.from_numpy(assign[t + "_p2g"].astype(np.int64)) for t in tags]
    pw = [torch.from_numpy(assign[t + "_w"]) for t in tags]
    det.train()
    det.zero_grad()
    losses = det(img=img, img_metas=synth.img_metas(B), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    loss, _ = det._parse_losses(losses)
    loss.backward()
    g = torch.Generator().manual_seed(11)
    names, off, idx, val, rms = [], [0], [], [], []
    for n, p in det.named_parameters():
        if p.grad is None:
            continue
        k = min(48, p.numel())
        i = torch.randperm(p.numel(), generator=g)[:k]
        names.append(n)
        idx.append(i.numpy().astype(np.int64))
        val.append(p.grad.reshape(-1)[i].numpy())
        rms.append(float(p.grad.double().pow(2).mean().sqrt()))
        off.append(off[-1] + k)
    total = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in det.parameters() if p.grad is not None)))
    save(out_name, names=np.asarray(names), offsets=np.asarray(off, np.int64), idx=np.concatenate(idx),
         val=np.concatenate(val), rms=np.asarray(rms, np.float64), total_grad_norm=np.float64(total),
         losses=np.asarray([float(losses[k]) for k in ("loss_cls", "loss_bbox", "loss_iou")], np.float64))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ops2":       # only the second op set (the other fixtures stay as committed)
        gen_ops2()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "assigner_opts":   # only assigner_opts.npz
        gen_assigner_opts()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "recall":     # only recall.npz
        gen_recall()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stem":       # only model_grads_stem.npz: ResNet(frozen_stages=-1), trainable stem
        torch.manual_seed(0)
        model_cfg, train_cfg, test_cfg = ref_import.load_cfg()
        model_cfg["backbone"] = dict(model_cfg["backbone"], frozen_stages=-1)
        det = build_detector(model_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
        synth.fill_state_dict(det.state_dict(), seed=0)
        gen_model_grads(det, dict(np.load(os.path.join(HERE, "assigner.npz"))), out_name="model_grads_stem")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "grads":      # only model_grads.npz (inputs from the committed assigner.npz)
        torch.manual_seed(0)
        model_cfg, train_cfg, test_cfg = ref_import.load_cfg()
        det = build_detector(model_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
        synth.fill_state_dict(det.state_dict(), seed=0)
        gen_model_grads(det, dict(np.load(os.path.join(HERE, "assigner.npz"))))
        return
    torch.manual_seed(0)
    model_cfg, train_cfg, test_cfg = ref_import.load_cfg()
    gen_anchors()
    assign = gen_assigner()
    gen_assigner_opts()
    gen_ops()
    gen_ops2()
    gen_nms(test_cfg)
    det = build_detector(model_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=0)
    gen_head(det, assign)
    gen_model(det, assign)
    gen_model_grads(det, assign)


if __name__ == "__main__":
    main()
