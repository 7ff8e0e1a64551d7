"""radet.ops-compatible entry points (radet/ops/__init__.py:1-11 of the reference), executed by the
HIP NMS kernel (radet_amd/csrc/decode_nms.hip).  Inputs may be CPU / GPU tensors or ndarrays; results come
back on the input's device, same shapes / dtypes / ordering as the reference's C++ ops."""
import numpy as np
import torch

from .. import kernels as K

__all__ = ["vote_nms", "global_vote_nms", "cluster_nms", "batched_nms", "MBD_box2distance", "GDT_box2distance", "MBD", "GDT",
           "mbd_batch", "gdt_batch", "border_seeds"]

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


def border_seeds(h, w, interval=3):
    """Seeds on the crop border every `interval` pixels -- top, bottom, left, right -- as the reference wrapper lays
    them out (bbox2distance_wrapper.py:22-36).  Returns (seeds_x, seeds_y) int64 tensors."""
    hx = list(range(0, w, interval))
    if hx[-1] != w - 1:
        hx.append(w - 1)
    hx = torch.tensor(hx, dtype=torch.int64)
    vy = torch.arange(1, h - 1, interval, dtype=torch.int64)
    sx = torch.cat((hx, hx, torch.zeros_like(vy), torch.full_like(vy, w - 1)))
    sy = torch.cat((torch.zeros_like(hx), torch.full_like(hx, h - 1), vy, vy))
    return sx, sy


def _pack_crops(items, seeds, dtype, chans):
    """items: list of [h, w(, 3)] arrays/tensors; seeds: list of (sx, sy). -> packed device buffers + descriptor"""
    dev = _dev()
    desc, px, so = [], 0, 0
    flat, sxs, sys_ = [], [], []
    for it, (sx, sy) in zip(items, seeds):
        t = torch.as_tensor(it)
        h, w = int(t.shape[0]), int(t.shape[1])
        assert h >= 2 and w >= 2 and (t.dim() == 3 and t.shape[2] == 3 if chans == 3 else t.dim() == 2)
        sx, sy = torch.as_tensor(sx).reshape(-1).to(torch.int32), torch.as_tensor(sy).reshape(-1).to(torch.int32)
        desc += [px, h, w, so, int(sx.numel())]
        flat.append(t.to(dtype).reshape(-1))
        sxs.append(sx); sys_.append(sy)
        px += h * w
        so += int(sx.numel())
    return (torch.cat(flat).to(dev), torch.tensor(desc, dtype=torch.int32).to(dev), torch.cat(sxs).to(dev),
            torch.cat(sys_).to(dev), px, [(d[1], d[2]) for d in zip(*[iter(desc)] * 5)])


def mbd_batch(images, seeds, alpha=0.1, niter=4, base_size=300):
    """Minimum-barrier distance maps of a list of u8 HWC crops (one workgroup per crop); seeds: list of (sx, sy).
    Returns a list of f64 [h, w] device tensors == bbox2distance_ext.MBD per crop (bit-identical)."""
    if not len(images):
        return []
    img, desc, sx, sy, px, hw = _pack_crops(images, seeds, torch.uint8, 3)
    dmap = torch.empty(px, dtype=torch.float64, device=img.device)
    ws = torch.empty(K.mbd_ws_bytes(px), dtype=torch.uint8, device=img.device)
    K.mbd(img, desc, len(images), sx, sy, float(alpha), int(niter), int(base_size), dmap, px, ws)
    out, o = [], 0
    for h, w in hw:
        out.append(dmap[o:o + h * w].view(h, w))
        o += h * w
    return out


def gdt_batch(costs, seeds):
    """Geodesic distance transforms of a list of f32 [h, w] cost maps; == bbox2distance_ext.GDT per map."""
    if not len(costs):
        return []
    cost, desc, sx, sy, px, hw = _pack_crops(costs, seeds, torch.float32, 1)
    dist = torch.empty(px, dtype=torch.float32, device=cost.device)
    ws = torch.empty(px, dtype=torch.int32, device=cost.device)
    K.gdt(cost, desc, len(costs), sx, sy, dist, ws)
    out, o = [], 0
    for h, w in hw:
        out.append(dist[o:o + h * w].view(h, w))
        o += h * w
    return out


def MBD(image, seeds_x, seeds_y, alpha, niter, base_size):
    """Signature of the reference's pybind op `bbox2distance_ext.MBD` (bbox2distance_ext.cpp:127-133)."""
    return mbd_batch([image], [(seeds_x, seeds_y)], alpha, niter, base_size)[0]


def GDT(costmap, seeds_x, seeds_y):
    """Signature of `bbox2distance_ext.GDT` (bbox2distance_ext.cpp:225-236)."""
    return gdt_batch([costmap], [(seeds_x, seeds_y)])[0]


_NEEDS_CV2 = ("needs cv2.resize / cv2.GaussianBlur / Sobel, which this build does not restate (cv2 is absent from the "
              "image, so they could not be pinned); pass pre-processed crops / cost maps to mbd_batch / gdt_batch")


class MBD_box2distance:
    """`radet.ops.MBD_box2distance` (bbox2distance_wrapper.py:9-95) on the GPU, all enabled crops of a call in one
    launch.  mode='mean' with multi_scale=False is complete; mode='center' and multi_scale additionally resize / blur
    with cv2 in the reference and raise here."""

    def __init__(self, mode="center", multi_scale=False, alpha=0.1, niter=4, base_size=300, interval=3):
        assert mode in ["center", "mean"]
        self.mode, self.multi_scale, self.alpha, self.niter = mode, multi_scale, alpha, niter
        self.base_size, self.interval, self.size = base_size, interval, 150

    def cal_dmap_single_scale(self, image):
        h, w = image.shape[:2]
        return MBD(image, *border_seeds(h, w, self.interval), self.alpha, self.niter, self.base_size)

    def __call__(self, box_images, mask_enable, bbox_images_xy):
        if self.mode == "center" or isinstance(self.multi_scale, dict):
            raise NotImplementedError("MBD_box2distance(mode='center' / multi_scale): " + _NEEDS_CV2)
        idx = [i for i, e in enumerate(mask_enable) if e]
        maps = mbd_batch([box_images[i] for i in idx], [border_seeds(*box_images[i].shape[:2], self.interval) for i in idx],
                         self.alpha, self.niter, self.base_size)
        maps = dict(zip(idx, maps))
        out = []
        for i, (img, xy) in enumerate(zip(box_images, bbox_images_xy)):
            d = maps[i] if i in maps else torch.ones(img.shape[:2], dtype=torch.uint8, device=_dev())
            out.append(d[xy[1]:xy[-1], xy[0]:xy[2]])
        return out


class GDT_box2distance:
    """`radet.ops.GDT_box2distance` (bbox2distance_wrapper.py:98-185): the transform runs on the GPU; the edge map that
    feeds it comes from `extract_edge_func(image) -> f32[h, w]` (the reference's Sobel / structured-edge extractors are
    cv2 code and are not restated), mode='mean' only."""

    def __init__(self, edge_mode="sobel", interval=3, mode="center", extract_edge_func=None):
        self.interval, self.mode, self.size = interval, mode, 150
        self.extract_edge_func = extract_edge_func

    def cal_dmap_single_scale(self, box_image):
        if self.extract_edge_func is None:
            raise NotImplementedError("GDT_box2distance edge extraction: " + _NEEDS_CV2)
        h, w = box_image.shape[:2]
        return GDT(self.extract_edge_func(box_image), *border_seeds(h, w, self.interval))

    def __call__(self, box_images, mask_enable, bbox_images_xy):
        if self.mode == "center":
            raise NotImplementedError("GDT_box2distance(mode='center'): " + _NEEDS_CV2)
        if self.extract_edge_func is None:
            raise NotImplementedError("GDT_box2distance edge extraction: " + _NEEDS_CV2)
        idx = [i for i, e in enumerate(mask_enable) if e]
        maps = gdt_batch([self.extract_edge_func(box_images[i]) for i in idx],
                         [border_seeds(*box_images[i].shape[:2], self.interval) for i in idx])
        maps = dict(zip(idx, maps))
        out = []
        for i, (img, xy) in enumerate(zip(box_images, bbox_images_xy)):
            d = maps[i] if i in maps else torch.ones(img.shape[:2], dtype=torch.float32, device=_dev())
            out.append(d[xy[1]:xy[-1], xy[0]:xy[2]])
        return out
