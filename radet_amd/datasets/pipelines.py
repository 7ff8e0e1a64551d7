"""Data-pipeline stages that belong to the detector hot path: the visibility-guided label assigner
(radet/datasets/pipelines/label_assignment.py:15-201) on the GPU, batched over images, and GenerateDistanceMap
(loading.py:543-650): the pass-through of the visible masks and the mask-free variant built on the GPU box-to-distance
transforms."""
import ctypes as C
import math

import numpy as np
import torch

from .. import kernels as K
from ..utils import Registry, build_from_cfg

PIPELINES = Registry("pipeline")
INF = 1e8


def build_pipeline(cfg):
    return build_from_cfg(cfg, PIPELINES)


@PIPELINES.register_module()
class GenerateDistanceMap:
    """radet/datasets/pipelines/loading.py:543-650.  with_gt_mask=True (every BOP config): the visible masks are the
    sampling maps.  with_gt_mask=False (mask-free sampler): every gt box is cropped with `pad_ratio` padding (random fill
    colour outside the image, Python's `random` like the reference), the crops go through the GDT / MBD box-to-distance
    transform on the GPU (radet_amd.ops.GDT_box2distance / MBD_box2distance, all crops of the image batched) and the
    resulting maps are pasted into zero images: `distance_maps` = f32 [G, img_h, img_w] on the device."""

    def __init__(self, with_gt_mask=True, small_object_size=32 ** 2, pad_ratio=0.05, distance_transform="gdt", **kwargs):
        self.with_gt_mask, self.small_object_size = with_gt_mask, small_object_size
        if not with_gt_mask:
            from ..ops import GDT_box2distance, MBD_box2distance
            self.pad_ratio = pad_ratio
            if distance_transform == "gdt":
                self.distance_transform = GDT_box2distance(**kwargs)
            elif distance_transform == "mbd":
                self.distance_transform = MBD_box2distance(**kwargs)
            else:
                raise RuntimeError(f"Unexpected distance transform type, expect 'mbd' or 'gdt', got{distance_transform}")

    def forward_with_gt_mask(self, results):
        return results["gt_masks"]

    def crop_boxes(self, img, img_shape, gt_bboxes):
        """What loading.py:596-634 hands to the distance transform, computed for all G boxes at once: per box a crop canvas
        of the box grown by ceil(pad_ratio * side) on every side, filled with one random colour (three `random.randint(0, 255)`
        per box, box after box: the only sequencing visible from outside) and overlaid with the part of the image the grown
        window covers; the box's own rectangle inside its canvas; and which boxes exceed `small_object_size`.
        Returns (list of u8 [h, w, 3] canvases, bool [G], int [G, 4] regions)."""
        import random
        bounds = np.array([img_shape[1] - 1, img_shape[0] - 1], dtype=np.int_)        # last valid (x, y)
        side = gt_bboxes[:, 2:4] - gt_bboxes[:, 0:2]
        large = (side + 1).prod(axis=1) > self.small_object_size
        corners = gt_bboxes.astype(np.int_)                                             # truncation
        lo, hi = corners[:, 0:2], corners[:, 2:4]
        grow = np.ceil((hi - lo) * self.pad_ratio).astype(np.int_)
        canvas_wh = hi - lo + 2 * grow
        win_lo, win_hi = lo - grow, hi + grow                                           # grown window, image coordinates
        src_lo, src_hi = np.clip(win_lo, 0, bounds), np.clip(win_hi, 0, bounds)         # its part inside the image
        dst_lo, dst_hi = src_lo - win_lo, canvas_wh - (win_hi - src_hi)                 # where that part sits on the canvas
        regions = np.concatenate([grow, canvas_wh - grow], axis=1)
        fill = np.array([random.randint(0, 255) for _ in range(3 * len(corners))], dtype=np.uint8).reshape(-1, 3)
        canvases = []
        for g, (w, h) in enumerate(canvas_wh):
            canvas = np.empty((h, w, 3), dtype=np.uint8)
            canvas[...] = fill[g]
            canvas[dst_lo[g, 1]:dst_hi[g, 1], dst_lo[g, 0]:dst_hi[g, 0]] = img[src_lo[g, 1]:src_hi[g, 1], src_lo[g, 0]:src_hi[g, 0]]
            canvases.append(canvas)
        return canvases, large, regions

    def forward_wo_gt_mask(self, results):
        """loading.py:586-645: distance maps of the padded box crops, pasted back at the (truncated) box positions of
        otherwise-zero images; here one f32 [G, img_h, img_w] device tensor instead of G host tensors."""
        img, (img_h, img_w) = results["img"], results["img_shape"][:2]
        for ok, why in ((isinstance(img, np.ndarray), f"image should be numpy.ndarray, got {type(img)}"),
                        (getattr(img, "dtype", None) == np.uint8, f"image dtype should be np.uint8, got{getattr(img, 'dtype', None)}"),
                        (getattr(img, "ndim", 0) == 3, f"image should have three channel and BGR format, got{getattr(img, 'ndim', 0)} channels")):
            if not ok:
                raise AssertionError(why)          # the reference's three asserts, same messages
        gt_bboxes = results["gt_bboxes"]
        maps = self.distance_transform(*self.crop_boxes(img, (img_h, img_w), gt_bboxes))
        dev = maps[0].device if maps else torch.device("cuda", torch.cuda.current_device())
        pasted = torch.zeros(len(maps), img_h, img_w, dtype=torch.float32, device=dev)
        for g, (x0, y0, x1, y1) in enumerate(gt_bboxes.astype(np.int_)):
            pasted[g, y0:y1, x0:x1] = maps[g]
        return pasted

    def __call__(self, results):
        results["distance_maps"] = self.forward_with_gt_mask(results) if self.with_gt_mask else self.forward_wo_gt_mask(results)
        return results


@PIPELINES.register_module()
class LabelAssignment:
    MAX_GTS = 256
    """Same constructor and result keys as the reference. `__call__(results)` handles one image like the
    reference; `assign_batch` is the MI355X-native entry point (one workgroup per image).

    RNG: the reference consumes NumPy's global legacy RandomState. Here the next raw 32-bit outputs of `rng` (default: the
    RandomState behind the global np.random functions) are read on the host and handed to the kernel, which performs numpy's
    own draws on them -- the weighted `choice(p=...)` (two outputs per uniform) or, with random_sample_by_distance=False, the
    integer draws of `choice()` without p -- and reports how many it consumed; the RNG is then advanced by exactly that
    number: results AND the generator's state are identical to the reference's for equal seeds.  Images that share one RNG
    object are assigned one after the other so that each continues the stream where the previous one stopped (the reference's
    sequential consumption); one RandomState per image runs as a single launch."""

    def __init__(self, strides=(8, 16, 32, 64, 128),
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF)), anchor_generator_cfg=None,
                 positive_num=10, neg_threshold=0.2, adapt_positive_num=False, balance_sample=False,
                 multiply_samplepro_for_weight=False, ambiguous_sample="min_area", random_sample_by_distance=True):
        assert len(strides) == len(regress_ranges)
        if adapt_positive_num and anchor_generator_cfg is not None:
            # the adapted positive_num reads the anchor side of a level (label_assignment.py:88-110: concat_anchor_boxes[:, 2] -
            # [:, 0]); kernel and oracle compute it as 8 * stride -- the one anchor per location of every RADet config.  Any
            # other generator would silently change K per gt: refuse it.
            ag = dict(anchor_generator_cfg)
            ok = (ag.get("octave_base_scale", 8) == 8 and ag.get("scales_per_octave", 1) == 1
                  and [float(r) for r in ag.get("ratios", [1.0])] == [1.0]
                  and tuple(ag.get("strides", strides)) == tuple(strides) and float(ag.get("center_offset", 0.0)) == 0.0
                  and ag.get("scales") is None and ag.get("base_sizes") is None)
            if not ok:
                raise NotImplementedError("LabelAssignment(adapt_positive_num=True): the anchor side is 8 * stride here (ratios [1.0], "
                                          "octave_base_scale 8, scales_per_octave 1, center_offset 0, strides as given); got "
                                          f"anchor_generator_cfg={anchor_generator_cfg}")
        if ambiguous_sample != "min_area":
            # ('max_dis' does not run in the reference either: label_assignment.py:158-161 reads an undefined `is_candidate`)
            raise NotImplementedError("LabelAssignment: ambiguous_sample='min_area' is the only rule the reference can execute")
        self.strides, self.regress_ranges = tuple(strides), tuple(tuple(r) for r in regress_ranges)
        self.positive_num, self.neg_threshold = positive_num, neg_threshold
        self.adapt_positive_num, self.balance_sample = bool(adapt_positive_num), bool(balance_sample)
        self.multiply_sample_pro_for_weight = bool(multiply_samplepro_for_weight)
        self.random_sample_by_distance = bool(random_sample_by_distance)
        self.flags = (1 if balance_sample else 0) | (2 if multiply_samplepro_for_weight else 0) | (4 if adapt_positive_num else 0) \
            | (0 if random_sample_by_distance else 8)
        # raw 32-bit outputs handed to the kernel per image: two per uniform of the weighted draw (positive_num + rejection
        # redraws per gt: 256 gts x 10 fit); the uniform integer draw shuffles ALL non-negative candidates of a gt (one or two
        # words each): its budget grows on demand
        self.word_budget = 8192

    def _levels(self, H, W):
        return K.Levels([(math.ceil(H / s), math.ceil(W / s)) for s in self.strides], 1)

    def assign_batch(self, gt_bboxes, masks, img_shape, rngs=None, device=None):
        """gt_bboxes: list of f32[G_b,4]; masks: list of u8[G_b,H,W] (ndarray or tensor); rngs: list of
        np.random.RandomState (or None -> global). Returns (p2g i64[B,N], pw f32[B,N]) device tensors."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        B = len(gt_bboxes)
        rngs = list(rngs) if rngs is not None else [None] * B
        # (None, the np.random module and np.random.mtrand._rand are ONE generator: compare what they resolve to)
        if B > 1 and len({id(np.random.mtrand._rand if (r is None or r is np.random) else r) for r in rngs}) < B:
            # several images share one RNG object (e.g. the global np.random, as in the reference's loader): image i
            # must start where image i-1 stopped in that stream, which is only known after i-1 has been assigned ->
            # one launch per image, in order.  Distinct RandomStates per image (the fast path below) need no ordering.
            outs = [self.assign_batch([gt_bboxes[i]], [masks[i]], img_shape, rngs=[rngs[i]], device=dev) for i in range(B)]
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        H, W = int(img_shape[0]), int(img_shape[1])
        lv = self._levels(H, W)
        N = lv.rows
        counts = [int(np.asarray(b).reshape(-1, 4).shape[0]) for b in gt_bboxes]
        if max(counts, default=0) > self.MAX_GTS:       # kernel limit (ASG_MAXG): refuse before anything is launched
            raise ValueError(f"LabelAssignment: {max(counts)} gt boxes in one image exceed the kernel's limit of "
                             f"{self.MAX_GTS} (BOP scenes hold at most a few dozen objects)")
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        tot = int(off[-1])
        boxes = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in gt_bboxes]) if tot else np.zeros((1, 4), np.float32)
        if tot:
            ts = [torch.as_tensor(np.asarray(m) if not isinstance(m, torch.Tensor) else m).reshape(-1, H, W)
                  for m, c in zip(masks, counts) if c > 0]
            # visible masks are bytes; the mask-free sampler hands float distance maps (read as float32 like the reference)
            mdt = torch.float32 if any(t.is_floating_point() for t in ts) else torch.uint8
            mk = torch.cat([t.to(dev, mdt) for t in ts]).contiguous()
        else:
            mk = torch.zeros(1, H, W, dtype=torch.uint8, device=dev)
        ldesc, nlvl = K.level_desc(lv, self.strides)
        rr = (C.c_float * (2 * nlvl))(*[float(v) for r in self.regress_ranges for v in r])
        p2g = torch.empty(B, N, dtype=torch.long, device=dev)
        pw = torch.empty(B, N, device=dev)
        used = torch.zeros(B, dtype=torch.int32, device=dev)
        ws = torch.empty(K.assign_ws_bytes(B, N), dtype=torch.uint8, device=dev)
        boxes_d, off_d = torch.from_numpy(boxes).to(dev), torch.from_numpy(off).to(dev)
        # the RandomStates' next raw outputs (legacy RandomState = MT19937: random_sample() is two of them, the integer draws
        # of choice() without p one per trial); the global np.random is the RandomState behind the module functions
        gens = [np.random.mtrand._rand if (r is None or r is np.random) else r for r in rngs]
        states = [g.get_state() for g in gens]
        U = self.word_budget
        # what a stream can need: two words per weighted draw incl. redraws, one or two per candidate of a shuffle -- a few
        # words per (point, gt) at the very most; beyond that the stream cannot be satisfied and more words only cost memory
        U_max = max(self.word_budget, min(1 << 22, 4 * N * max(max(counts, default=1), 1) + 64 * self.positive_num * max(tot, 1)))
        while True:
            words = np.empty((B, U), np.uint32)
            for i, g in enumerate(gens):
                g.set_state(states[i])
                words[i] = g._bit_generator.random_raw(U)
            K.assign_points(boxes_d, off_d, mk, H, W, torch.from_numpy(words.view(np.int32)).to(dev), U, ldesc, rr, nlvl, B,
                            self.positive_num, float(self.neg_threshold), p2g, pw, used, ws, flags=self.flags)
            used_h = used.cpu().numpy()
            if (used_h == -1).any() and U < U_max:         # stream exhausted (a shuffle of thousands of candidates): more words,
                U = min(4 * U, U_max)                      # same results -- the kernel is a function of the stream's prefix
                continue
            if (used_h == -1).any() and U < (1 << 24):     # the rejection redraws of a weighted choice are bounded heuristically
                U = U_max = 1 << 24                        # only: one last pass at the former hard cap before giving up
                continue
            break
        for g, st in zip(gens, states):
            g.set_state(st)
        if (used_h < 0).any():
            raise RuntimeError("LabelAssignment: random stream exhausted (-1) / more than 256 gts (-2) / an adapted positive_num "
                               f"above 64 (-3): codes {used_h.tolist()}")
        for g, k in zip(gens, used_h):              # leave each RNG exactly where the reference would
            if k:
                g._bit_generator.random_raw(int(k))
        return p2g, pw

    def __call__(self, results):
        h, w, _ = results["img_shape"]
        dm = results["distance_maps"]
        if hasattr(dm, "masks") and isinstance(dm.masks, torch.Tensor):
            dm = dm.masks                                     # device-resident BitmapMasks
        elif hasattr(dm, "to_ndarray"):
            dm = dm.to_ndarray()
        elif isinstance(dm, (list, tuple)) and len(dm) and isinstance(dm[0], torch.Tensor):
            dm = torch.stack(list(dm))                        # per-box distance maps of the mask-free sampler
        elif not isinstance(dm, torch.Tensor):
            dm = np.asarray(dm)
        p2g, pw = self.assign_batch([results["gt_bboxes"]], [dm], (h, w))
        results["points_to_gt_index"] = p2g[0].cpu().numpy()
        results["points_weight"] = pw[0].cpu().numpy()
        return results
