"""radet.ops-compatible entry points (radet/ops/__init__.py:1-11 of the reference), executed by the
HIP NMS kernel (radet_amd/csrc/decode_nms.hip).  Inputs may be CPU / GPU tensors or ndarrays; results come
back on the input's device, same shapes / dtypes / ordering as the reference's C++ ops."""
import numpy as np
import torch

from .. import kernels as K

__all__ = ["vote_nms", "global_vote_nms", "cluster_nms", "batched_nms", "MBD_box2distance", "GDT_box2distance", "MBD", "GDT",
           "mbd_batch", "gdt_batch", "border_seeds", "resize_batch", "gaussian_blur9_batch", "sobel_edge_batch"]

_MAX = 65536      # 16-bit positions in the kernel's sort keys (above 8192 the sorts run in global memory)


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
        raise ValueError(f"NMS input of {n} boxes exceeds the kernel's capacity ({_MAX})")
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


# ---------------------------------------------------------------------------------------------- image processing (packed crops)
class _Packed:
    """box crops of one call packed back to back on the device: data [sum h*w (* c)], desc i32 [n, 3] = (pixel offset, h, w)"""

    def __init__(self, data, hw, channels):
        self.data, self.hw, self.c = data, [(int(h), int(w)) for h, w in hw], channels
        offs, o = [], 0
        for h, w in self.hw:
            offs.append(o)
            o += h * w
        self.offs, self.px = offs, o
        self.desc = torch.tensor([[o_, h, w] for o_, (h, w) in zip(offs, self.hw)], dtype=torch.int32,
                                 device=data.device).reshape(-1, 3).contiguous()
        self.max_px = max([h * w for h, w in self.hw], default=0)

    @staticmethod
    def pack(items, dtype, channels):
        dev = _dev()
        ts = [torch.as_tensor(it).to(dev, dtype) for it in items]
        for t in ts:
            assert (t.dim() == 3 and t.shape[2] == channels) if channels > 1 else t.dim() == 2, tuple(t.shape)
        flat = torch.cat([t.reshape(-1) for t in ts]) if ts else torch.empty(0, dtype=dtype, device=dev)
        return _Packed(flat.contiguous(), [t.shape[:2] for t in ts], channels)

    def empty_like(self, hw=None, dtype=None, channels=None):
        hw = self.hw if hw is None else hw
        c = self.c if channels is None else channels
        n = sum(int(h) * int(w) for h, w in hw) * c
        return _Packed(torch.empty(n, dtype=dtype or self.data.dtype, device=self.data.device), hw, c)

    def unpack(self):
        out = []
        for o, (h, w) in zip(self.offs, self.hw):
            t = self.data[o * self.c:(o + h * w) * self.c]
            out.append(t.view(h, w, self.c) if self.c > 1 else t.view(h, w))
        return out


def _resize(p, dsizes):
    """cv2.resize(crop, (w, h)) per crop (INTER_LINEAR): uint8 HWC, float32 or float64 HW"""
    out = p.empty_like(hw=[(int(h), int(w)) for (w, h) in dsizes])
    if len(p.hw):
        if p.data.dtype == torch.uint8:
            K.resize_linear_u8(p.data, p.desc, out.data, out.desc, len(p.hw), out.max_px, p.c)
        else:
            K.resize_linear_f(p.data, p.desc, out.data, out.desc, len(p.hw), out.max_px)
    return out


def _blur9(p):
    out = p.empty_like()
    if len(p.hw):
        tmp = torch.empty(p.px * 3, dtype=torch.float32, device=p.data.device)
        K.gaussian_blur9_u8(p.data, p.desc, out.data, tmp, len(p.hw), p.max_px)
    return out


def _sobel(p):
    out = p.empty_like(dtype=torch.float32, channels=1)
    if len(p.hw):
        gray = torch.empty(p.px, dtype=torch.uint8, device=p.data.device)
        mx = torch.empty(len(p.hw), dtype=torch.int32, device=p.data.device)
        K.sobel_edge(p.data, p.desc, out.data, gray, mx, len(p.hw), p.max_px)
    return out


def resize_batch(images, dsizes):
    """list of crops (uint8 [h, w, 3] or float [h, w]) resized to dsizes = [(w, h), ...] like cv2.resize(img, (w, h))"""
    t0 = torch.as_tensor(images[0])
    if t0.dtype == torch.uint8:
        p = _Packed.pack(images, torch.uint8, 3 if t0.dim() == 3 else 1)
    else:
        p = _Packed.pack(images, torch.float64 if t0.dtype == torch.float64 else torch.float32, 1)
    return _resize(p, dsizes).unpack()


def gaussian_blur9_batch(images):
    """cv2.GaussianBlur(img, (9, 9), sigmaX=0, borderType=BORDER_DEFAULT) per uint8 [h, w, 3] crop"""
    return _blur9(_Packed.pack(images, torch.uint8, 3)).unpack()


def sobel_edge_batch(images):
    """GDT_box2distance.sobel_extract_edge per uint8 [h, w, 3] crop -> float32 [h, w]"""
    return _sobel(_Packed.pack(images, torch.uint8, 3)).unpack()


def _center_prepare(box_images, idx, size):
    """mode='center': short edge -> `size` pixels (cv2.resize), then the 9x9 Gaussian (wrapper.py:80-88 / 170-177)"""
    orig, dsz = [], []
    for i in idx:
        h, w = box_images[i].shape[:2]
        ratio = size / min(w, h)
        dsz.append((int(w * ratio), int(h * ratio)))
        orig.append((w, h))
    p = _Packed.pack([box_images[i] for i in idx], torch.uint8, 3)
    return _blur9(_resize(p, dsz)), orig


class MBD_box2distance:
    """`radet.ops.MBD_box2distance` (bbox2distance_wrapper.py:9-95) on the GPU, all enabled crops of a call batched:
    mode='center' (the reference default) = resize to a 150-pixel short edge -> 9x9 Gaussian -> MBD -> resize back;
    mode='mean' = MBD on the crop itself.  `multi_scale` is stored and, like in the reference's __call__, not used."""

    def __init__(self, mode="center", multi_scale=False, alpha=0.1, niter=4, base_size=300, interval=3):
        assert mode in ["center", "mean"]
        self.mode, self.multi_scale, self.alpha, self.niter = mode, multi_scale, alpha, niter
        self.base_size, self.interval, self.size = base_size, interval, 150

    def cal_dmap_single_scale(self, image):
        h, w = image.shape[:2]
        return MBD(image, *border_seeds(h, w, self.interval), self.alpha, self.niter, self.base_size)

    def __call__(self, box_images, mask_enable, bbox_images_xy):
        idx = [i for i, e in enumerate(mask_enable) if e]
        if self.mode == "center" and idx:
            blurred, orig = _center_prepare(box_images, idx, self.size)
            crops = blurred.unpack()
            maps = mbd_batch(crops, [border_seeds(h, w, self.interval) for h, w in blurred.hw], self.alpha, self.niter,
                             self.base_size)
            maps = _resize(_Packed.pack(maps, torch.float64, 1), orig).unpack()
        else:
            maps = mbd_batch([box_images[i] for i in idx],
                             [border_seeds(*box_images[i].shape[:2], self.interval) for i in idx], self.alpha, self.niter,
                             self.base_size)
        maps = dict(zip(idx, maps))
        out = []
        for i, (img, xy) in enumerate(zip(box_images, bbox_images_xy)):
            d = maps[i] if i in maps else torch.ones(img.shape[:2], dtype=torch.uint8, device=_dev())
            out.append(d[xy[1]:xy[-1], xy[0]:xy[2]])
        return out


class GDT_box2distance:
    """`radet.ops.GDT_box2distance` (bbox2distance_wrapper.py:98-185) on the GPU.  edge_mode='sobel' runs the Sobel edge
    extractor as HIP passes; the reference's default edge_mode='sed' needs OpenCV's structured-edge model file
    (cv2.ximgproc, 'model.yml'), which is data this repository does not have: pass `extract_edge_func(image) -> f32[h, w]`
    for it."""

    def __init__(self, edge_mode="sed", interval=3, mode="center", extract_edge_func=None):
        self.interval, self.mode, self.size, self.edge_mode = interval, mode, 150, edge_mode
        self.extract_edge_func = extract_edge_func
        if extract_edge_func is None and edge_mode == "sed":
            raise NotImplementedError("GDT_box2distance(edge_mode='sed') needs cv2.ximgproc's structured-edge model "
                                      "(model.yml); use edge_mode='sobel' or pass extract_edge_func")

    def sobel_extract_edge(self, image):
        return sobel_edge_batch([image])[0]

    def _edges(self, crops):
        if self.extract_edge_func is not None:
            return [torch.as_tensor(self.extract_edge_func(c.cpu().numpy() if isinstance(c, torch.Tensor) else c)) for c in crops]
        return sobel_edge_batch(crops)

    def cal_dmap_single_scale(self, box_image):
        h, w = box_image.shape[:2]
        return GDT(self._edges([box_image])[0], *border_seeds(h, w, self.interval))

    def __call__(self, box_images, mask_enable, bbox_images_xy):
        idx = [i for i, e in enumerate(mask_enable) if e]
        if self.mode == "center" and idx:
            blurred, orig = _center_prepare(box_images, idx, self.size)
            maps = gdt_batch(self._edges(blurred.unpack()), [border_seeds(h, w, self.interval) for h, w in blurred.hw])
            maps = _resize(_Packed.pack(maps, torch.float32, 1), orig).unpack()
        else:
            maps = gdt_batch(self._edges([box_images[i] for i in idx]),
                             [border_seeds(*box_images[i].shape[:2], self.interval) for i in idx])
        maps = dict(zip(idx, maps))
        out = []
        for i, (img, xy) in enumerate(zip(box_images, bbox_images_xy)):
            d = maps[i] if i in maps else torch.ones(img.shape[:2], dtype=torch.float32, device=_dev())
            out.append(d[xy[1]:xy[-1], xy[0]:xy[2]])
        return out
