"""ORACLE (test infrastructure, never imported by radet_amd/): PyTorch-CPU fp32 restatement of the
reference's detector forward / loss / decode path, written functionally over a state dict that
uses the reference's parameter names (SURVEY.md §8b "State-dict compatibility").

Follows:
  ResNet / Bottleneck (style='pytorch', BN eval)  radet/models/backbones/resnet.py:260-299, 622-648
  FPN (start_level=1, add_extra_convs='on_output') radet/models/necks/fpn.py:170-221
  ATSS/RADet head forward                          radet/models/dense_heads/atss_head.py:118-145,
                                                   radet_head.py:27-30
  targets                                          radet_head.py:290-392, tblr_bbox_coder.py:71-114
  loss                                             radet_head.py:173-288, focal_loss.py:10-41,
                                                   iou_loss.py:82-98, iou2d_calculator.py:43-159,
                                                   cross_entropy_loss.py:58-91, losses/utils.py
  decode + NMS                                     radet_head.py:55-169, atss_head.py:325-387,
                                                   tblr_bbox_coder.py:117-172, vote_wrapper.py:7-43
  _parse_losses                                    radet/models/detectors/base.py:185-218
Pinned by tests/golden/{ops,head_loss,get_bboxes,model}.npz (outputs of the reference run here).
This is also the "port" CPU baseline timed by bench.py.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import nms as onms

STRIDES = (8, 16, 32, 64, 128)
NUM_CLASSES = 21
ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


# ----------------------------------------------------------------------------- parameters
def make_state_dict(depth=50, num_classes=NUM_CLASSES):
    """Zero-filled tensors with the reference's names and shapes (OIHW conv weights)."""
    sd = OrderedDict()

    def conv(name, co, ci, k, bias=False):
        sd[name + ".weight"] = torch.zeros(co, ci, k, k)
        if bias:
            sd[name + ".bias"] = torch.zeros(co)

    def bn(name, c):
        sd[name + ".weight"] = torch.zeros(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.zeros(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("backbone.conv1", 64, 3, 7)
    bn("backbone.bn1", 64)
    inplanes = 64
    for li, nblocks in enumerate(ARCH[depth]):
        planes = 64 * 2 ** li
        for b in range(nblocks):
            p = f"backbone.layer{li + 1}.{b}"
            conv(p + ".conv1", planes, inplanes, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1)
            bn(p + ".bn3", planes * 4)
            if b == 0:
                conv(p + ".downsample.0", planes * 4, inplanes, 1)
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i, ci in enumerate((512, 1024, 2048)):
        conv(f"neck.lateral_convs.{i}.conv", 256, ci, 1, bias=True)
    for i in range(5):
        conv(f"neck.fpn_convs.{i}.conv", 256, 256, 3, bias=True)
    for tower in ("cls_convs", "reg_convs"):
        for i in range(4):
            conv(f"bbox_head.{tower}.{i}.conv", 256, 256, 3)
            sd[f"bbox_head.{tower}.{i}.gn.weight"] = torch.zeros(256)
            sd[f"bbox_head.{tower}.{i}.gn.bias"] = torch.zeros(256)
    conv("bbox_head.atss_cls", num_classes, 256, 3, bias=True)
    conv("bbox_head.atss_reg", 4, 256, 3, bias=True)
    conv("bbox_head.atss_centerness", 1, 256, 3, bias=True)
    for i in range(5):
        sd[f"bbox_head.scales.{i}.scale"] = torch.zeros(())
    return sd


def is_trainable(name, frozen_stages=1):
    """requires_grad rule of the reference config (stem + layer1 frozen; buffers never)."""
    if name.endswith(("running_mean", "running_var", "num_batches_tracked")):
        return False
    if name.startswith(("backbone.conv1", "backbone.bn1")):
        return frozen_stages < 0
    for s in range(1, frozen_stages + 1):
        if name.startswith(f"backbone.layer{s}."):
            return False
    return True


# ----------------------------------------------------------------------------- forward
# Mixed-precision arithmetic of BASELINE config 3 (the reference's fp16 wrapper, apis/train.py:113-117, in bf16 and
# without loss scaling): every matrix-core convolution rounds BOTH operands to bf16 (RNE) and accumulates in fp32,
# in forward, dgrad and wgrad; eval-mode BN is folded into the weight BEFORE the rounding (as the HIP engine does);
# the 7x7 stem (frozen, VALU kernel), GroupNorm, the loss and the optimizer stay fp32.
MATH = "fp32"


class conv_math:
    def __init__(self, mode):
        assert mode in ("fp32", "bf16")
        self.mode = mode

    def __enter__(self):
        global MATH
        self.prev, MATH = MATH, self.mode

    def __exit__(self, *a):
        global MATH
        MATH = self.prev


def _r(t):
    return t.bfloat16().float()


class _ConvBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, padding)
        return F.conv2d(_r(x), _r(w), None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding = ctx.sp
        gyr = _r(gy)
        gx = torch.nn.grad.conv2d_input(x.shape, _r(w), gyr, stride=stride, padding=padding) if ctx.needs_input_grad[0] else None
        gw = torch.nn.grad.conv2d_weight(_r(x), w.shape, gyr, stride=stride, padding=padding) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


def _conv(x, w, b=None, stride=1, padding=0):
    if MATH == "fp32":
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    y = _ConvBF16.apply(x, w, stride, padding)
    return y if b is None else y + b.view(1, -1, 1, 1)


# Test hook: RELU_HOOK(name, pre_activation) -> a 0 / 1 tensor (or None) that REPLACES the ReLU's own decision y > 0.  Two
# correct fp32 implementations decide a pre-activation within an ulp of zero differently ("knife edge"), and one flipped mask
# moves a whole row of a weight gradient; with the other implementation's masks handed in, everything else can be compared
# tightly (tests/test_gpu_model.py::test_gradients_vs_fp64_oracle).  Names: stem, l<stage>.<block>.o1 / .o2 / .out,
# cls.y<i>.L<level>, reg.y<i>.L<level>, bbox.L<level>.
RELU_HOOK = None


def _relu(x, name):
    if RELU_HOOK is not None:
        m = RELU_HOOK(name, x)
        if m is not None:
            return x * m.to(x.dtype)
    return F.relu(x)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=1e-5)


def _conv_bn(x, sd, wname, p, stride=1, padding=0):
    """conv (no bias) followed by eval-mode BN `p`"""
    if MATH == "fp32":
        return _bn(F.conv2d(x, sd[wname], stride=stride, padding=padding), sd, p)
    s = sd[p + ".weight"] * torch.rsqrt(sd[p + ".running_var"] + 1e-5)
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * s
    return _conv(x, sd[wname] * s.view(-1, 1, 1, 1), None, stride, padding) + shift.view(1, -1, 1, 1)


def backbone(sd, img, depth=50):
    x = F.conv2d(img, sd["backbone.conv1.weight"], stride=2, padding=3)
    x = _relu(_bn(x, sd, "backbone.bn1"), "stem")
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    outs = []
    for li, nblocks in enumerate(ARCH[depth]):
        for b in range(nblocks):
            p = f"backbone.layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            idt = x
            o = _relu(_conv_bn(x, sd, p + ".conv1.weight", p + ".bn1"), f"l{li + 1}.{b}.o1")
            o = _relu(_conv_bn(o, sd, p + ".conv2.weight", p + ".bn2", stride=stride, padding=1), f"l{li + 1}.{b}.o2")
            o = _conv_bn(o, sd, p + ".conv3.weight", p + ".bn3")
            if b == 0:
                idt = _conv_bn(x, sd, p + ".downsample.0.weight", p + ".downsample.1", stride=stride)
            x = _relu(o + idt, f"l{li + 1}.{b}.out")
        outs.append(x)
    return outs  # C2..C5


def neck(sd, feats):
    lat = [_conv(feats[i + 1], sd[f"neck.lateral_convs.{i}.conv.weight"], sd[f"neck.lateral_convs.{i}.conv.bias"])
           for i in range(3)]
    for i in (2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    outs = [_conv(lat[i], sd[f"neck.fpn_convs.{i}.conv.weight"], sd[f"neck.fpn_convs.{i}.conv.bias"], padding=1)
            for i in range(3)]
    for i in (3, 4):
        outs.append(_conv(outs[-1], sd[f"neck.fpn_convs.{i}.conv.weight"], sd[f"neck.fpn_convs.{i}.conv.bias"],
                          stride=2, padding=1))
    return outs  # P3..P7


def head(sd, feats):
    cls_scores, bbox_preds, iou_preds = [], [], []
    for l, x in enumerate(feats):
        c, r = x, x
        for i in range(4):
            c = _relu(F.group_norm(_conv(c, sd[f"bbox_head.cls_convs.{i}.conv.weight"], padding=1), 32,
                                   sd[f"bbox_head.cls_convs.{i}.gn.weight"], sd[f"bbox_head.cls_convs.{i}.gn.bias"], 1e-5),
                      f"cls.y{i}.L{l}")
            r = _relu(F.group_norm(_conv(r, sd[f"bbox_head.reg_convs.{i}.conv.weight"], padding=1), 32,
                                   sd[f"bbox_head.reg_convs.{i}.gn.weight"], sd[f"bbox_head.reg_convs.{i}.gn.bias"], 1e-5),
                      f"reg.y{i}.L{l}")
        cls_scores.append(_conv(c, sd["bbox_head.atss_cls.weight"], sd["bbox_head.atss_cls.bias"], padding=1))
        reg = _conv(r, sd["bbox_head.atss_reg.weight"], sd["bbox_head.atss_reg.bias"], padding=1)
        bbox_preds.append(_relu(reg * sd[f"bbox_head.scales.{l}.scale"], f"bbox.L{l}"))
        iou_preds.append(_conv(r, sd["bbox_head.atss_centerness.weight"], sd["bbox_head.atss_centerness.bias"], padding=1))
    return cls_scores, bbox_preds, iou_preds


def flatten_levels(ts):
    """list of [B,C,h,w] -> [sum_l B*h*w, C], level-major then image then row-major (radet_head.py:222-243)."""
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


# ----------------------------------------------------------------------------- geometry
def grid_anchors(level_hw, strides=STRIDES):
    out = []
    for (h, w), s in zip(level_hw, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) * s, torch.arange(w, dtype=torch.float32) * s,
                                indexing="ij")
        c = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
        half = 4.0 * s
        out.append(torch.cat([c - half, c + half], 1))
    return out


def tblr_encode(priors, gts, normalizer=0.125):
    cx = (priors[:, 0] + priors[:, 2]) / 2
    cy = (priors[:, 1] + priors[:, 3]) / 2
    w = priors[:, 2] - priors[:, 0]
    h = priors[:, 3] - priors[:, 1]
    loc = torch.stack([(cy - gts[:, 1]) / h, (gts[:, 3] - cy) / h, (cx - gts[:, 0]) / w, (gts[:, 2] - cx) / w], 1)
    return loc / normalizer


def tblr_decode(priors, tblr, normalizer=0.125, max_shape=None):
    cx = (priors[:, 0] + priors[:, 2]) / 2
    cy = (priors[:, 1] + priors[:, 3]) / 2
    w = priors[:, 2] - priors[:, 0]
    h = priors[:, 3] - priors[:, 1]
    d = tblr * normalizer
    top, bottom, left, right = d[:, 0] * h, d[:, 1] * h, d[:, 2] * w, d[:, 3] * w
    x1, y1, x2, y2 = cx - left, cy - top, cx + right, cy + bottom
    if max_shape is not None:
        x1, x2 = x1.clamp(0, max_shape[1]), x2.clamp(0, max_shape[1])
        y1, y2 = y1.clamp(0, max_shape[0]), y2.clamp(0, max_shape[0])
    return torch.stack([x1, y1, x2, y2], 1)


def overlaps_aligned(a, b, mode="iou", eps=1e-6):
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = torch.max(area1 + area2 - inter, inter.new_tensor(eps))
    iou = inter / union
    if mode == "iou":
        return iou
    elt = torch.min(a[:, :2], b[:, :2])
    erb = torch.max(a[:, 2:], b[:, 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[:, 0] * ewh[:, 1], inter.new_tensor(eps))
    return iou - (earea - union) / earea


def overlaps_matrix(a, b, mode="iou", eps=1e-6):
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = torch.max(area1[:, None] + area2[None, :] - inter, inter.new_tensor(eps))
    iou = inter / union
    if mode == "iou":
        return iou
    elt = torch.min(a[:, None, :2], b[None, :, :2])
    erb = torch.max(a[:, None, 2:], b[None, :, 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], inter.new_tensor(eps))
    return iou - (earea - union) / earea


# ----------------------------------------------------------------------------- losses
def focal_elementwise(logits, labels, gamma=2.0, alpha=0.25):
    """Sigmoid focal loss per (row, class); label == C is background (focal_loss.py:10-41 on one-hot)."""
    t = F.one_hot(labels.clamp(0, logits.shape[1]), logits.shape[1] + 1)[:, :logits.shape[1]].to(logits.dtype)
    p = logits.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(logits, t, reduction="none") * fw


def build_targets(gt_bboxes, gt_labels, p2g, pw, level_hw, num_classes=NUM_CLASSES):
    """Per-image targets regrouped level-major (radet_head.py:290-392). Returns flat
    labels i64[B*N], bbox_targets f32[B*N,4], weights f32[B*N], anchors f32[B*N,4]."""
    anchors_lv = grid_anchors(level_hw)
    anchors = torch.cat(anchors_lv)
    nl = [a.shape[0] for a in anchors_lv]
    B = len(gt_bboxes)
    labs, tgts, wts = [], [], []
    for b in range(B):
        N = anchors.shape[0]
        lab = torch.full((N,), num_classes, dtype=torch.long)
        tgt = torch.zeros(N, 4)
        if gt_labels[b].shape[0] > 0:
            nonneg = p2g[b] > -1
            pos = p2g[b] > 0
            lab[nonneg] = gt_labels[b][p2g[b][nonneg] - 1]        # ignore (0) -> gt_labels[-1] quirk
            tgt[pos] = tblr_encode(anchors[pos], gt_bboxes[b][p2g[b][pos] - 1])
        labs.append(lab.split(nl))
        tgts.append(tgt.split(nl))
        wts.append(pw[b].split(nl))
    cat = lambda xs: torch.cat([torch.cat([x[l] for x in xs]) for l in range(len(nl))])  # noqa: E731
    anc = torch.cat([a.repeat(B, 1) for a in anchors_lv])
    return cat(labs), cat(tgts), cat(wts), anc


def head_loss(cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, p2g, pw, num_classes=NUM_CLASSES):
    level_hw = [t.shape[-2:] for t in cls_scores]
    B = cls_scores[0].shape[0]
    fc, fb, fi = flatten_levels(cls_scores), flatten_levels(bbox_preds), flatten_levels(iou_preds).reshape(-1)
    labels, tgts, weights, anchors = build_targets(gt_bboxes, gt_labels, p2g, pw, level_hw, num_classes)
    pos = ((labels >= 0) & (labels < num_classes)).nonzero().reshape(-1)
    pos_w = weights[pos]
    num_pos = pos_w.sum()
    loss_cls = (focal_elementwise(fc, labels) * weights[:, None]).sum() / (num_pos + B)
    if num_pos > 0:
        pa = anchors[pos]
        dp = tblr_decode(pa, fb[pos])
        dt = tblr_decode(pa, tgts[pos])
        iou_t = overlaps_aligned(dp, dt).detach()
        w = iou_t.clamp(min=1e-12) * pos_w
        loss_bbox = 2.0 * ((1 - overlaps_aligned(dp, dt, "giou", eps=1e-6)) * w).sum() / w.sum()
        loss_iou = (F.binary_cross_entropy_with_logits(fi[pos], iou_t, reduction="none") * pos_w).sum() / pos_w.sum()
    else:
        loss_bbox = fb[pos].sum()
        loss_iou = fi[pos].sum()
    return dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_iou=loss_iou), (labels, tgts, weights, pos)


def parse_losses(losses):
    loss = sum(v for k, v in losses.items() if "loss" in k)
    return loss


# ----------------------------------------------------------------------------- decode + NMS
def get_bboxes_single(cls_scores, bbox_preds, iou_preds, img_shape, scale_factor, test_cfg, rescale=True):
    """One image: per-level threshold / top-k / decode, then NMS (radet_head.py:55-169)."""
    level_hw = [t.shape[-2:] for t in cls_scores]
    anchors_lv = grid_anchors(level_hw)
    bs, ss, cs, ls = [], [], [], []
    for c, r, q, anc in zip(cls_scores, bbox_preds, iou_preds, anchors_lv):
        scores = c.permute(1, 2, 0).reshape(-1, c.shape[0]).sigmoid()
        reg = r.permute(1, 2, 0).reshape(-1, 4)
        ctr = q.permute(1, 2, 0).reshape(-1).sigmoid()
        cand = scores > test_cfg["score_thr"]
        k = min(int(test_cfg.get("nms_pre", -1)), int(cand.sum()))
        if k == 0:
            continue
        vals, top = scores[cand].topk(k, sorted=False)
        nz = cand.nonzero()[top]
        pi, ci = nz[:, 0], nz[:, 1]
        bs.append(tblr_decode(anc[pi], reg[pi], max_shape=img_shape))
        ss.append(vals)
        cs.append(ctr[pi])
        ls.append(ci)
    if not bs:
        return np.zeros((0, 5), np.float32), np.zeros((0,), np.int64)
    boxes = torch.cat(bs)
    if rescale:
        boxes = boxes / torch.as_tensor(scale_factor, dtype=boxes.dtype)
    scores, ctr, labels = torch.cat(ss), torch.cat(cs), torch.cat(ls)
    ncfg = dict(test_cfg["nms"])
    typ = ncfg.get("type")
    if typ == "vote":
        return onms.vote_nms(boxes.numpy(), scores.numpy(), labels.numpy(), ncfg, score_factor=ctr.numpy(),
                             max_num=test_cfg["max_per_img"])
    if typ == "global_vote":
        return onms.global_vote_nms(boxes.numpy(), scores.numpy(), labels.numpy(), ncfg, score_factor=ctr.numpy(),
                                    max_num=test_cfg["max_per_img"])
    dets, keep = onms.batched_nms(boxes.numpy(), (scores * ctr).numpy(), labels.numpy(), ncfg["iou_threshold"])
    if test_cfg["max_per_img"] > 0:
        dets, keep = dets[:test_cfg["max_per_img"]], keep[:test_cfg["max_per_img"]]
    return dets, labels.numpy()[keep]


def get_bboxes(cls_scores, bbox_preds, iou_preds, img_metas, test_cfg, rescale=True):
    out = []
    for b in range(cls_scores[0].shape[0]):
        out.append(get_bboxes_single([t[b].detach() for t in cls_scores], [t[b].detach() for t in bbox_preds],
                                     [t[b].detach() for t in iou_preds], img_metas[b]["img_shape"],
                                     img_metas[b]["scale_factor"], test_cfg, rescale))
    return out


# ----------------------------------------------------------------------------- whole model
class OracleDetector:
    """State-dict-driven detector. `sd` tensors that are trainable get requires_grad=True."""

    def __init__(self, depth=50, seed=None, test_cfg=None, math="fp32", num_classes=NUM_CLASSES, frozen_stages=1):
        from . import synth
        self.depth, self.math, self.num_classes = depth, math, num_classes
        self.sd = make_state_dict(depth, num_classes)
        if seed is not None:
            synth.fill_state_dict(self.sd, seed)
        for n, t in self.sd.items():
            if t.is_floating_point() and is_trainable(n, frozen_stages):
                t.requires_grad_(True)
        self.test_cfg = test_cfg or dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, max_per_img=100,
                                         nms=dict(type="vote", iou_threshold=0.65, cluster_score=["cls", "iou"],
                                                  vote_score=["iou", "cls"], iou_enable=False, sima=0.025))

    def extract_feat(self, img):
        return neck(self.sd, backbone(self.sd, img, self.depth))

    def forward_train(self, img, gt_bboxes, gt_labels, p2g, pw):
        with conv_math(self.math):
            outs = head(self.sd, self.extract_feat(img))
        losses, _ = head_loss(*outs, gt_bboxes, gt_labels, p2g, pw, num_classes=self.num_classes)
        return losses

    def simple_test(self, img, img_metas, rescale=True):
        with torch.no_grad(), conv_math(self.math):
            outs = head(self.sd, self.extract_feat(img))
        return get_bboxes(*outs, img_metas, self.test_cfg, rescale)

    def zero_grad(self):
        for t in self.sd.values():
            t.grad = None

    def named_grads(self):
        return {n: t.grad for n, t in self.sd.items() if t.requires_grad and t.grad is not None}
