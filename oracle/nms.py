"""ORACLE ctypes wrapper over oracle/liboracle.so (nms.c). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "nms.c")
    src2 = os.path.join(_HERE, "dist.c")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_vote_nms.restype = ctypes.c_int64
        _LIB.oracle_global_vote_nms.restype = ctypes.c_int64
        _LIB.oracle_batched_nms.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _vote(fn, boxes, cluster_scores, vote_scores, labels, thr, iou_enable, sigma):
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    cs = np.ascontiguousarray(cluster_scores, np.float32)
    vs = np.ascontiguousarray(vote_scores, np.float32)
    lb = np.ascontiguousarray(labels, np.int64)
    n = boxes.shape[0]
    ob = np.zeros((max(n, 1), 4), np.float32)
    ol = np.zeros(max(n, 1), np.int64)
    os_ = np.zeros(max(n, 1), np.float32)
    k = fn(_p(boxes), _p(cs), _p(vs), _p(lb), ctypes.c_int64(n), ctypes.c_float(thr),
           ctypes.c_int(int(iou_enable)), ctypes.c_float(sigma), _p(ob), _p(ol), _p(os_))
    return ob[:k], ol[:k], os_[:k]


def vote_nms_raw(boxes, cluster_scores, vote_scores, labels, thr=0.65, iou_enable=False, sigma=0.025):
    return _vote(lib().oracle_vote_nms, boxes, cluster_scores, vote_scores, labels, thr, iou_enable, sigma)


def global_vote_nms_raw(boxes, cluster_scores, vote_scores, labels, thr=0.65, iou_enable=False, sigma=0.025):
    return _vote(lib().oracle_global_vote_nms, boxes, cluster_scores, vote_scores, labels, thr, iou_enable, sigma)


def _wrap(raw, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num):
    """Wrapper semantics of radet/ops/vote/vote_wrapper.py:7-43 (incl. the list-valued score types
    and the 'sigma' key lookup, which the config misspells as 'sima')."""
    cfg = dict(nms_cfg)
    thr = cfg.pop("iou_threshold", 0.6)
    ctype = cfg.pop("cluster_score", "cls")
    vtype = cfg.pop("vote_score", "iou")
    iou_enable = cfg.pop("iou_enable", False)
    sigma = cfg.pop("sigma", 0.025)
    cls_scores = np.asarray(cls_scores, np.float32)
    sf = None if score_factor is None else np.asarray(score_factor, np.float32)

    def pick(t):
        if isinstance(t, (list, tuple)):
            return cls_scores * sf
        if t == "cls":
            return cls_scores
        if t == "iou":
            return sf
        raise RuntimeError(f"Unexpected score type:{t}")

    b, l, s = raw(bboxes, pick(ctype), pick(vtype), labels, thr, iou_enable, sigma)
    out = np.concatenate([b, s[:, None]], axis=1)
    if max_num > 0:
        out, l = out[:max_num], l[:max_num]
    return out, l


def vote_nms(bboxes, cls_scores, labels, nms_cfg, score_factor=None, max_num=0):
    return _wrap(vote_nms_raw, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num)


def global_vote_nms(bboxes, cls_scores, labels, nms_cfg, score_factor=None, max_num=0):
    return _wrap(global_vote_nms_raw, bboxes, cls_scores, labels, nms_cfg, score_factor, max_num)


def cluster_nms(bboxes, scores, labels, iou_threshold=0.65):
    boxes = np.ascontiguousarray(bboxes, np.float32).reshape(-1, 4)
    sc = np.ascontiguousarray(scores, np.float32)
    lb = np.ascontiguousarray(labels, np.int64)
    n = boxes.shape[0]
    ids = np.zeros(n, np.int64)
    num = np.zeros(n, np.int64)
    lib().oracle_cluster_nms(_p(boxes), _p(sc), _p(lb), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                             _p(ids), _p(num))
    return ids, num


def batched_nms(bboxes, scores, labels, iou_threshold, class_agnostic=False):
    boxes = np.ascontiguousarray(bboxes, np.float32).reshape(-1, 4)
    sc = np.ascontiguousarray(scores, np.float32)
    lb = np.ascontiguousarray(labels, np.int64)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    k = lib().oracle_batched_nms(_p(boxes), _p(sc), _p(lb), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                                 ctypes.c_int(int(class_agnostic)), _p(keep))
    keep = keep[:k]
    return np.concatenate([boxes[keep], sc[keep, None]], 1), keep
