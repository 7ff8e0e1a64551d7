from .anchor import ANCHOR_GENERATORS, AnchorGenerator, build_anchor_generator
from .bbox import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, MaxIoUAssigner, PseudoSampler, TBLRBBoxCoder,
                   bbox2result, build_assigner, build_bbox_coder, build_sampler)
from .mask import BitmapMasks, rescale_size
from .misc import multi_apply

__all__ = ["ANCHOR_GENERATORS", "AnchorGenerator", "build_anchor_generator", "BBOX_ASSIGNERS", "BBOX_CODERS",
           "BBOX_SAMPLERS", "MaxIoUAssigner", "PseudoSampler", "TBLRBBoxCoder", "bbox2result", "build_assigner",
           "build_bbox_coder", "build_sampler", "multi_apply", "BitmapMasks", "rescale_size"]
