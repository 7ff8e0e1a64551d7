"""`multiclass_nms` (radet/core/post_processing/bbox_nms.py:8-79) on the GPU: score filter = ordered stream
compaction (radet_threshold_compact), suppression = the hard-NMS mode of the HIP NMS pipeline
(radet_amd.ops.batched_nms).  Same arguments, return values and ordering as the reference."""
import torch

from .. import kernels as K


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    """multi_bboxes [n, #class*4] or [n, 4]; multi_scores [n, #class + 1] (last column = background, ignored).
    Returns (dets [k, 5], labels [k][, inds [k] into the flattened (n * #class) candidates])."""
    from ..ops import _dev, batched_nms
    src = multi_bboxes.device
    dev = _dev()
    num_classes = multi_scores.size(1) - 1
    n = multi_scores.size(0)
    boxes = multi_bboxes.detach().to(dev, torch.float32)
    if boxes.shape[1] > 4:
        boxes = boxes.reshape(n, -1, 4)
    else:
        boxes = boxes[:, None].expand(n, num_classes, 4)
    scores = multi_scores.detach().to(dev, torch.float32)[:, :-1]
    if score_factors is not None:
        scores = scores * score_factors.detach().to(dev, torch.float32)[:, None]   # bbox_nms.py:44-45 (elementwise product)
    scores = scores.reshape(-1).contiguous()
    idx = torch.empty(max(scores.numel(), 1), dtype=torch.long, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    K.threshold_compact(scores, float(score_thr), idx, cnt)
    inds = idx[:int(cnt.item())]
    labels = inds % num_classes if num_classes > 0 else inds           # labels = arange(C) tiled over the rows
    bboxes = boxes.reshape(-1, 4)[inds]
    sel_scores = scores[inds]
    if inds.numel() == 0:
        out = (bboxes.to(src), labels.to(src))
        return out + (inds.to(src),) if return_inds else out
    dets, keep = batched_nms(bboxes, sel_scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    if return_inds:
        return dets.to(src), labels[keep].to(src), keep.to(src)
    return dets.to(src), labels[keep].to(src)
