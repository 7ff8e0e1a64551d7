"""Box coder / assigner / sampler registries (radet/core/bbox/builder.py:1-20) and the pieces the
RADet hot path touches.  TBLR encode/decode arithmetic lives in the fused HIP kernels
(radet_head_loss, radet_decode_candidates); MaxIoUAssigner / PseudoSampler are constructed by the head
from train_cfg but never called on this path."""
import numpy as np
import torch

from ..utils import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry("bbox_assigner")
BBOX_SAMPLERS = Registry("bbox_sampler")
BBOX_CODERS = Registry("bbox_coder")


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


@BBOX_CODERS.register_module()
class TBLRBBoxCoder:
    def __init__(self, normalizer=4.0, clip_border=True):
        self.normalizer, self.clip_border = normalizer, clip_border


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
