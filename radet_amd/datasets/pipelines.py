"""Data-pipeline stages that belong to the detector hot path: the visibility-guided label assigner
(radet/datasets/pipelines/label_assignment.py:15-201) on the GPU, batched over images, and the
GenerateDistanceMap(with_gt_mask=True) pass-through (loading.py:579-581)."""
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
    def __init__(self, with_gt_mask=True, **kwargs):
        if not with_gt_mask:
            raise NotImplementedError("GenerateDistanceMap(with_gt_mask=False): the MBD / GDT transforms exist "
                                      "(radet_amd.ops.mbd_batch / gdt_batch), the cv2 crop resize / blur / edge extraction "
                                      "around them is not restated; every BOP config uses with_gt_mask=True")

    def __call__(self, results):
        results["distance_maps"] = results["gt_masks"]
        return results


@PIPELINES.register_module()
class LabelAssignment:
    """Same constructor and result keys as the reference. `__call__(results)` handles one image like the
    reference; `assign_batch` is the MI355X-native entry point (one workgroup per image).

    RNG: the reference consumes NumPy's global legacy RandomState. Here the stream is drawn on the host
    from `rng` (default: the global np.random, whose state is advanced by exactly the number of uniforms
    the kernel consumed) and the draw itself runs on the GPU -- results are identical for equal seeds.  Images that
    share one RNG object are assigned one after the other so that each continues the stream where the previous
    one stopped (the reference's sequential consumption); one RandomState per image runs as a single launch."""

    def __init__(self, strides=(8, 16, 32, 64, 128),
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF)), anchor_generator_cfg=None,
                 positive_num=10, neg_threshold=0.2, adapt_positive_num=False, balance_sample=False,
                 multiply_samplepro_for_weight=False, ambiguous_sample="min_area", random_sample_by_distance=True):
        assert len(strides) == len(regress_ranges)
        if adapt_positive_num or not balance_sample or multiply_samplepro_for_weight or ambiguous_sample != "min_area" \
                or not random_sample_by_distance:
            raise NotImplementedError("LabelAssignment on MI355X implements the configuration of "
                                      "configs/base/datasets/bop_detection.py (balance_sample=True, min_area, "
                                      "probability-weighted draw)")
        self.strides, self.regress_ranges = tuple(strides), tuple(tuple(r) for r in regress_ranges)
        self.positive_num, self.neg_threshold = positive_num, neg_threshold
        self.uniform_budget = 4096

    def _levels(self, H, W):
        return K.Levels([(math.ceil(H / s), math.ceil(W / s)) for s in self.strides], 1)

    def assign_batch(self, gt_bboxes, masks, img_shape, rngs=None, device=None):
        """gt_bboxes: list of f32[G_b,4]; masks: list of u8[G_b,H,W] (ndarray or tensor); rngs: list of
        np.random.RandomState (or None -> global). Returns (p2g i64[B,N], pw f32[B,N]) device tensors."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        B = len(gt_bboxes)
        rngs = list(rngs) if rngs is not None else [None] * B
        if B > 1 and len({id(r) for r in rngs}) < B:
            # several images share one RNG object (e.g. the global np.random, as in the reference's loader): image i
            # must start where image i-1 stopped in that stream, which is only known after i-1 has been assigned ->
            # one launch per image, in order.  Distinct RandomStates per image (the fast path below) need no ordering.
            outs = [self.assign_batch([gt_bboxes[i]], [masks[i]], img_shape, rngs=[rngs[i]], device=dev) for i in range(B)]
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        H, W = int(img_shape[0]), int(img_shape[1])
        lv = self._levels(H, W)
        N = lv.rows
        counts = [int(np.asarray(b).reshape(-1, 4).shape[0]) for b in gt_bboxes]
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        tot = int(off[-1])
        boxes = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in gt_bboxes]) if tot else np.zeros((1, 4), np.float32)
        if tot:
            mk = torch.cat([torch.as_tensor(np.asarray(m) if not isinstance(m, torch.Tensor) else m).reshape(-1, H, W).to(torch.uint8)
                            for m, c in zip(masks, counts) if c > 0]).to(dev).contiguous()
        else:
            mk = torch.zeros(1, H, W, dtype=torch.uint8, device=dev)
        U = self.uniform_budget
        states, u = [], np.empty((B, U), np.float64)
        for i, r in enumerate(rngs):
            r = np.random if r is None else r
            states.append((r, r.get_state()))
            u[i] = r.random_sample(U)
        ldesc, nlvl = K.level_desc(lv, self.strides)
        rr = (C.c_float * (2 * nlvl))(*[float(v) for r in self.regress_ranges for v in r])
        p2g = torch.empty(B, N, dtype=torch.long, device=dev)
        pw = torch.empty(B, N, device=dev)
        used = torch.zeros(B, dtype=torch.int32, device=dev)
        ws = torch.empty(K.assign_ws_bytes(B, N), dtype=torch.uint8, device=dev)
        K.assign_points(torch.from_numpy(boxes).to(dev), torch.from_numpy(off).to(dev), mk, H, W,
                        torch.from_numpy(u).to(dev), U, ldesc, rr, nlvl, B, self.positive_num, float(self.neg_threshold),
                        p2g, pw, used, ws)
        used_h = used.cpu().numpy()
        if (used_h < 0).any():
            raise RuntimeError(f"LabelAssignment: uniform stream exhausted / too many gts (codes {used_h.tolist()})")
        for (r, st), k in zip(states, used_h):      # leave each RNG exactly where the reference would
            r.set_state(st)
            if k:
                r.random_sample(int(k))
        return p2g, pw

    def __call__(self, results):
        h, w, _ = results["img_shape"]
        dm = results["distance_maps"]
        dm = dm.to_ndarray() if hasattr(dm, "to_ndarray") else np.asarray(dm)
        p2g, pw = self.assign_batch([results["gt_bboxes"]], [dm], (h, w))
        results["points_to_gt_index"] = p2g[0].cpu().numpy()
        results["points_weight"] = pw[0].cpu().numpy()
        return results
