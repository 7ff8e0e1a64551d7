"""radet.ops-compatible entry points (radet/ops/__init__.py:1-11 of the reference), executed by the
HIP NMS kernel (radet_amd/csrc/decode_nms.hip).  Inputs may be CPU / GPU tensors or ndarrays; results come
back on the input's device, same shapes / dtypes / ordering as the reference's C++ ops."""
import numpy as np
import torch

from .. import kernels as K

__all__ = ["vote_nms", "global_vote_nms", "cluster_nms", "batched_nms", "MBD_box2distance", "GDT_box2distance"]

_MAX = 8192


def _dev():
    if not torch.cuda.is_available():
        from .._lib import RadetHipError
        raise RadetHipError("radet_amd.ops need an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _t(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(_dev(), dtype).contiguous()


def _run(mode, boxes, cscore, vscore, labels, thr, iou_enable=False, sigma=0.025, max_out=0):
    n = int(boxes.shape[0])
    if n > _MAX:
        raise ValueError(f"NMS input of {n} boxes exceeds the on-chip sort capacity ({_MAX})")
    dev = _dev()
    cap = max(n, 1)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    k = max_out if max_out > 0 else cap
    ob = torch.zeros(1, k, 4, device=dev)
    osc = torch.zeros(1, k, device=dev)
    ol = torch.zeros(1, k, dtype=torch.long, device=dev)
    oc = torch.zeros(1, dtype=torch.int32, device=dev)
    a0 = torch.zeros(1, cap, dtype=torch.long, device=dev)
    a1 = torch.zeros(1, cap, dtype=torch.long, device=dev)
    ws = torch.empty(K.nms_ws_bytes(1, cap), dtype=torch.uint8, device=dev)
    if n == 0:
        return ob[0, :0], osc[0, :0], ol[0, :0], a0[0, :0], a1[0, :0]
    K.nms(boxes, cscore, vscore, labels, cnt, 1, cap, mode, float(thr), bool(iou_enable), float(sigma), int(max_out), ob,
          osc, ol, oc, a0, a1, ws)
    kk = int(oc.item())
    return ob[0, :kk], osc[0, :kk], ol[0, :kk], a0[0], a1[0]


def _vote(mode, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num):
    """Wrapper semantics of radet/ops/vote/vote_wrapper.py:7-83: list-valued score types select
    cls*score_factor; the op pops 'sigma' (the BOP configs spell it 'sima', so the default 0.025 applies)."""
    src = bboxes.device if isinstance(bboxes, torch.Tensor) else torch.device("cpu")
    cfg = dict(nms_cfg)
    thr = cfg.pop("iou_threshold", 0.6)
    ctype = cfg.pop("cluster_score", "cls")
    vtype = cfg.pop("vote_score", "iou")
    iou_enable = cfg.pop("iou_enable", False)
    sigma = cfg.pop("sigma", 0.025)
    b = _t(bboxes, torch.float32).reshape(-1, 4)
    c = _t(cls_scores, torch.float32).reshape(-1)
    f = None if score_factor is None else _t(score_factor, torch.float32).reshape(-1)
    lab = _t(labels, torch.long).reshape(-1)

    def pick(t):
        if isinstance(t, (list, tuple)):
            return (c * f).contiguous()
        if t == "cls":
            return c
        if t == "iou":
            return f
        raise RuntimeError(f"Unexpected score type:{t}")

    ob, osc, ol, _, _ = _run(mode, b, pick(ctype), pick(vtype), lab, thr, iou_enable, sigma, max_out=max_num)
    dets = torch.cat([ob, osc[:, None]], dim=-1)
    return dets.to(src), ol.to(src)


def vote_nms(bboxes, cls_scores, labels, nms_cfg, score_factor=None, max_num=0):
    return _vote(0, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num)


def global_vote_nms(bboxes, cls_scores, labels, nms_cfg, score_factor=None, max_num=0):
    return _vote(1, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num)


def cluster_nms(bboxes, scores, categories, iou_threshold=0.65):
    """radet/ops/cluster/cluster_wrapper.py:7-22 -> (instance_ids i64[N], clusters_num i64[N])."""
    src = bboxes.device if isinstance(bboxes, torch.Tensor) else torch.device("cpu")
    b = _t(bboxes, torch.float32).reshape(-1, 4)
    s = _t(scores, torch.float32).reshape(-1)
    lab = _t(categories, torch.long).reshape(-1)
    _, _, _, ids, num = _run(2, b, s, s, lab, iou_threshold)
    n = b.shape[0]
    return ids[:n].to(src), num[:n].to(src)


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms semantics as used by radet_head.py:160: returns (cat[boxes, score], keep)."""
    src = boxes.device if isinstance(boxes, torch.Tensor) else torch.device("cpu")
    cfg = dict(nms_cfg)
    thr = cfg.get("iou_threshold", cfg.get("iou_thr", 0.5))
    b = _t(boxes, torch.float32).reshape(-1, 4)
    s = _t(scores, torch.float32).reshape(-1)
    lab = _t(idxs, torch.long).reshape(-1)
    if class_agnostic:
        lab = torch.zeros_like(lab)
    n = b.shape[0]
    ob, osc, _, keep, _ = _run(3, b, s, s, lab, thr, max_out=max(n, 1))
    k = ob.shape[0]
    return torch.cat([ob, osc[:, None]], -1).to(src), keep[:k].to(src)


class _OutOfScope:
    def __init__(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__}: the MBD/GDT box-to-distance transforms (radet/ops/bbox2distance) never run in the "
            "BOP configs (GenerateDistanceMap(with_gt_mask=True)); they are listed as 'next' in SURVEY.md §8f")


class MBD_box2distance(_OutOfScope):
    pass


class GDT_box2distance(_OutOfScope):
    pass
