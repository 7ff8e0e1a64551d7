"""RADet single-stage detector: same constructor, call signatures and outputs as
radet/models/detectors/{base.py:65-253, single_stage.py:17-124, radet.py:8-32}."""
import operator
import weakref
from collections import OrderedDict
from functools import reduce

import torch
import torch.distributed as dist
from torch import nn

from ..core import bbox2result
from ..utils import to_config_dict
from .builder import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module()
class RADet(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        train_cfg, test_cfg = to_config_dict(train_cfg), to_config_dict(test_cfg)
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fp16_enabled = False
        self._runtime = None
        for m in (self.backbone, self.neck, self.bbox_head):     # module-level API reaches the runtime through the owner
            if m is not None:
                object.__setattr__(m, "_owner_ref", weakref.ref(self))
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return self.neck is not None

    @property
    def with_bbox(self):
        return self.bbox_head is not None

    def init_weights(self, pretrained=None):
        """single_stage.py:36-52 -> `backbone.init_weights(pretrained)` (resnet.py:590-599): the checkpoint is loaded into
        the BACKBONE (non-strict, like mmcv's load_checkpoint).  `torchvision://<name>` resolves offline to
        $RADET_PRETRAINED_DIR/<name>.pth or torch-hub's cache (~/.cache/torch/hub/checkpoints/<name>-*.pth); the
        reference would download it, here a missing file is an error instead of a silent random init."""
        if pretrained is None:
            return
        import glob
        import os
        path = str(pretrained)
        if path.startswith("torchvision://"):
            name = path[len("torchvision://"):]
            cands = []
            if os.environ.get("RADET_PRETRAINED_DIR"):
                cands += [os.path.join(os.environ["RADET_PRETRAINED_DIR"], name + ext) for ext in (".pth", ".pt")]
            hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")
            cands += sorted(glob.glob(os.path.join(hub, name + "-*.pth")))
            found = [c for c in cands if os.path.exists(c)]
            if not found:
                raise FileNotFoundError(
                    f"pretrained={pretrained!r}: no local copy (looked for $RADET_PRETRAINED_DIR/{name}.pth and "
                    f"{hub}/{name}-*.pth; there is no network to download it).  Pass pretrained=None for a random init.")
            path = found[0]
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        if any(k.startswith("backbone.") for k in sd):           # a detector checkpoint: take its backbone
            sd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        own = self.backbone.state_dict()
        hit = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
        if not hit:
            raise RuntimeError(f"pretrained={pretrained!r}: none of its {len(sd)} entries matches a backbone parameter "
                               "(expected torchvision-style keys such as 'conv1.weight', 'layer1.0.conv1.weight')")
        self.backbone.load_state_dict(hit, strict=False)
        missing = [k for k in own if k not in hit and not k.endswith("num_batches_tracked")]
        if missing:
            import warnings
            warnings.warn(f"pretrained={pretrained!r}: {len(missing)} backbone entries not in the checkpoint "
                          f"(e.g. {missing[:3]})")

    # ------------------------------------------------------------------ runtime
    def runtime(self, math=None):
        """The HIP engine behind this module.  math: "fp32" | "bf16"; default: "bf16" when the model was wrapped for
        mixed precision (`fp16_enabled`, what mmcv's wrap_fp16_model sets -- apis/train.py:113-117), else RADET_MATH."""
        from ..runtime import DetectorRuntime
        if math is None and getattr(self, "fp16_enabled", False):
            math = getattr(self, "fp16_mode", "bf16-storage")
        rt = self._runtime
        if rt is not None and math is not None and rt.engine.math_name != math:
            rt = None
        if rt is None or not rt.flat.still_bound():
            rt = DetectorRuntime(self, depth=self.backbone.depth, num_classes=self.bbox_head.num_classes,
                                 frozen_stages=self.backbone.frozen_stages, strides=self.bbox_head.strides,
                                 stacked_convs=self.bbox_head.stacked_convs, math=math)
            object.__setattr__(self, "_runtime", rt)
            rt.owner = weakref.ref(self)
        return rt

    def invalidate_folded_weights(self):
        """Re-fold the conv weights (BN scale, layouts, plane triples) on the next call.  Needed only after writing
        parameters or BN statistics through `.data` (e.g. `p.data.copy_()`, an EMA hook) or with a raw kernel: writes
        torch tracks -- `load_state_dict`, optimizers, `copy_` on the Parameter -- are seen by themselves."""
        rt = self._runtime
        if rt is not None:
            rt.engine.invalidate_fold()

    def train(self, mode=True):
        # mode switches are rare and typically follow user code (checkpoint surgery, EMA swaps): fold again afterwards
        self.invalidate_folded_weights()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_folded_weights()
        return super()._load_from_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------ reference API
    def extract_feat(self, img):
        return self.runtime().extract_feat_api(img)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, points_to_gt_index, points_weight,
                      gt_bboxes_ignore=None):
        batch_input_shape = tuple(img[0].size()[-2:])
        for m in img_metas:
            m["batch_input_shape"] = batch_input_shape
        return self.runtime().losses_autograd(img, gt_bboxes, gt_labels, points_to_gt_index, points_weight)

    def forward_test(self, imgs, img_metas, **kwargs):
        if not isinstance(imgs, list):
            imgs, img_metas = [imgs], [img_metas]
        if len(imgs) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})")
        if len(imgs) != 1:
            raise NotImplementedError("test-time augmentation is out of scope (flip=False in the BOP configs)")
        for m in img_metas[0]:
            m["batch_input_shape"] = tuple(imgs[0].size()[-2:])
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def simple_test(self, img, img_metas, rescale=False):
        dets = self.runtime().detect(img, img_metas, self.test_cfg, rescale)
        return [bbox2result(b, l, self.bbox_head.num_classes) for b, l in dets]

    def _parse_losses(self, losses):
        """detectors/base.py:185-216, same (loss, log_vars) result.  The logged scalars of the step are stacked into ONE
        tensor: one all-reduce (mean over the ranks) and one device -> host copy for all of them, instead of one
        collective and one `.item()` synchronisation per entry."""
        def scalar(key, entry):
            if torch.is_tensor(entry):
                return entry.mean()
            if isinstance(entry, list):
                return reduce(operator.add, (t.mean() for t in entry))       # left to right, like the reference's sum()
            raise TypeError(f"{key} is not a tensor or list of tensors")

        keys = list(losses)
        terms = [scalar(k, losses[k]) for k in keys]
        picked = [t for k, t in zip(keys, terms) if "loss" in k]
        loss = reduce(operator.add, picked) if picked else 0     # left to right; (0 = the reference's `sum()` of nothing)
        report = torch.stack([t.detach() for t in terms] + [torch.as_tensor(loss).detach().to(terms[0])]) if terms \
            else torch.zeros(1)
        if dist.is_available() and dist.is_initialized():
            report = report / dist.get_world_size()
            dist.all_reduce(report)
        return loss, OrderedDict(zip(keys + ["loss"], report.tolist()))

    def train_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["img_metas"]))

    def val_step(self, data, optimizer):
        return self.train_step(data, optimizer)
