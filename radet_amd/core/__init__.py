from .anchor import ANCHOR_GENERATORS, AnchorGenerator, build_anchor_generator
from .bbox import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS, BboxOverlaps2D, MaxIoUAssigner,
                   PseudoSampler, TBLRBBoxCoder, bbox2result, bbox_overlaps, bboxes2tblr, build_assigner, build_bbox_coder,
                   build_iou_calculator, build_sampler, tblr2bboxes)
from .post_processing import multiclass_nms
from .mask import BitmapMasks, rescale_size
from .misc import multi_apply

__all__ = ["ANCHOR_GENERATORS", "AnchorGenerator", "build_anchor_generator", "BBOX_ASSIGNERS", "BBOX_CODERS",
           "BBOX_SAMPLERS", "MaxIoUAssigner", "PseudoSampler", "TBLRBBoxCoder", "bbox2result", "build_assigner",
           "build_bbox_coder", "build_sampler", "multi_apply", "BitmapMasks", "rescale_size", "IOU_CALCULATORS", "BboxOverlaps2D", "bbox_overlaps",
           "build_iou_calculator", "bboxes2tblr", "tblr2bboxes", "multiclass_nms"]
