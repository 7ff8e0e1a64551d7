"""Box coder / assigner / sampler / IoU-calculator registries (radet/core/bbox/builder.py:1-20,
iou_calculators/builder.py) and the pieces the RADet hot path touches.  Inside the detector the TBLR
arithmetic and the aligned IoU run fused (radet_head_loss, radet_decode_candidates); `TBLRBBoxCoder.encode /
decode` and `bbox_overlaps` / `BboxOverlaps2D` are the same arithmetic as stand-alone HIP kernels
(csrc/boxops.hip) for callers that use them directly.  MaxIoUAssigner / PseudoSampler are constructed by
the head from train_cfg but never called on this path."""
import numpy as np
import torch

from .. import kernels as K
from ..utils import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry("bbox_assigner")
BBOX_SAMPLERS = Registry("bbox_sampler")
BBOX_CODERS = Registry("bbox_coder")
IOU_CALCULATORS = Registry("IoU calculator")


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)


def _dev():
    if not torch.cuda.is_available():
        from .._lib import RadetHipError
        raise RadetHipError("radet_amd box operators need an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _boxes(t, dev):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.detach().to(dev, torch.float32).contiguous()


def bbox_overlaps(bboxes1, bboxes2, mode="iou", is_aligned=False, eps=1e-6):
    """radet/core/bbox/iou_calculators/iou2d_calculator.py:43-159 on the GPU: boxes [..., m, 4] x [..., n, 4] ->
    [..., m, n] (or [..., m] when is_aligned); mode 'iou' | 'iof' | 'giou'.  Same fp32 operation order as the
    reference expressions, so results equal the PyTorch-CPU ones bit for bit.  Returned on the inputs' device;
    not differentiable (GIoULoss carries its own backward)."""
    assert mode in ["iou", "iof", "giou"], f"Unsupported mode {mode}"
    assert (bboxes1.shape[-1] == 4 or bboxes1.shape[0] == 0)
    assert (bboxes2.shape[-1] == 4 or bboxes2.shape[0] == 0)
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    src = bboxes1.device if isinstance(bboxes1, torch.Tensor) else torch.device("cpu")
    batch_shape = tuple(bboxes1.shape[:-2])
    rows, cols = int(bboxes1.shape[-2]), int(bboxes2.shape[-2])
    if is_aligned:
        assert rows == cols
    oshape = batch_shape + ((rows,) if is_aligned else (rows, cols))
    if rows * cols == 0:
        return torch.empty(oshape, device=src)
    dev = _dev()
    b1, b2 = _boxes(bboxes1, dev), _boxes(bboxes2, dev)
    nb = 1
    for d in batch_shape:
        nb *= int(d)
    out = torch.empty(oshape, device=dev)
    K.bbox_overlaps(b1, b2, out, nb, rows, cols, mode, is_aligned, float(eps))
    return out.to(src)


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    """2D overlaps calculator (iou2d_calculator.py:6-40): accepts [m, 4] or [m, 5] (score column dropped)."""

    def __call__(self, bboxes1, bboxes2, mode="iou", is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5]
        assert bboxes2.size(-1) in [0, 4, 5]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + "()"


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


@BBOX_CODERS.register_module()
class TBLRBBoxCoder:
    """tblr_bbox_coder.py:8-69: (x1, y1, x2, y2) <-> (top, bottom, left, right) relative to the prior's centre,
    normalised by the prior's side lengths and by `normalizer` (a float or 4 factors)."""

    def __init__(self, normalizer=4.0, clip_border=True):
        self.normalizer, self.clip_border = normalizer, clip_border

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bboxes2tblr(bboxes, gt_bboxes, normalizer=self.normalizer)

    def decode(self, bboxes, pred_bboxes, max_shape=None):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return tblr2bboxes(bboxes, pred_bboxes, normalizer=self.normalizer, max_shape=max_shape,
                           clip_border=self.clip_border)


def bboxes2tblr(priors, gts, normalizer=4.0, normalize_by_wh=True):
    """tblr_bbox_coder.py:74-118 (radet_tblr_encode); bit-exact with the PyTorch-CPU expression"""
    assert priors.size(0) == gts.size(0)
    src = priors.device
    dev = _dev()
    p, g = _boxes(priors, dev), _boxes(gts, dev)
    out = torch.empty_like(p)
    K.tblr_encode(p, g, out, normalizer, normalize_by_wh)
    return out.to(src)


def tblr2bboxes(priors, tblr, normalizer=4.0, normalize_by_wh=True, max_shape=None, clip_border=True):
    """tblr_bbox_coder.py:121-172 (radet_tblr_decode)"""
    assert priors.size(0) == tblr.size(0)
    src = priors.device
    dev = _dev()
    p, t = _boxes(priors, dev), _boxes(tblr, dev)
    out = torch.empty_like(p)
    K.tblr_decode(p, t, out, normalizer, normalize_by_wh, max_shape, clip_border)
    return out.to(src)


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1, iou_calculator=None):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr, self.ignore_wrt_candidates = gt_max_assign_all, ignore_iof_thr, ignore_wrt_candidates
        self.match_low_quality, self.gpu_assign_thr = match_low_quality, gpu_assign_thr

    def assign(self, *a, **k):
        raise NotImplementedError("MaxIoUAssigner.assign is dead code on the RADet path (labels come from the "
                                  "visibility-guided LabelAssignment); only construction from train_cfg is supported")


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, *a, **k):
        raise NotImplementedError("PseudoSampler.sample is dead code on the RADet path")


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:101-116: [K,5] + labels -> list[num_classes] of ndarray[k,5]."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
