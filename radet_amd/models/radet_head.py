"""RADetHead (ATSS-style shared head), API- and parameter-compatible with
radet/models/dense_heads/{anchor_head.py:33-170, atss_head.py:26-145,325-387, radet_head.py:17-392}.
The class owns parameters / config objects and exposes the reference's methods; forward, loss and
get_bboxes run as HIP kernels through the detector runtime."""
from torch import nn

from ..core import build_anchor_generator, build_assigner, build_bbox_coder, build_sampler
from .builder import HEADS, build_loss
from .shells import ConvModuleShell, ConvShell, Scale, bias_init_with_prob, normal_


@HEADS.register_module()
class RADetHead(nn.Module):
    def __init__(self, num_classes, in_channels, strides=(8, 16, 32, 64, 128), stacked_convs=4, feat_channels=256,
                 conv_cfg=None, quality="centerness", norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                 anchor_generator=None, bbox_coder=None, reg_decoded_bbox=False,
                 loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type="GIoULoss", loss_weight=2.0),
                 loss_centerness=dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        if in_channels != 256 or feat_channels != 256 or conv_cfg is not None or norm_cfg.get("type") != "GN" \
                or norm_cfg.get("num_groups") != 32:
            raise NotImplementedError("RADetHead on MI355X: in_channels = feat_channels = 256 with GN(32) towers")
        self.num_classes, self.in_channels, self.feat_channels = num_classes, in_channels, feat_channels
        self.strides, self.stacked_convs, self.quality = tuple(strides), stacked_convs, quality
        self.use_sigmoid_cls = loss_cls.get("use_sigmoid", False)
        if not self.use_sigmoid_cls or loss_cls["type"] != "FocalLoss":
            raise NotImplementedError("RADetHead: loss_cls must be sigmoid FocalLoss (the fused kernel's formula)")
        self.cls_out_channels = num_classes
        self.sampling = False
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder or dict(type="TBLRBBoxCoder", normalizer=1 / 8))
        if abs(self.bbox_coder.normalizer - 0.125) > 1e-12:
            raise NotImplementedError("TBLRBBoxCoder.normalizer must be 1/8 (distance / stride targets)")
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.loss_centerness = build_loss(loss_centerness)
        self.loss_iou = self.loss_centerness
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(dict(type="PseudoSampler"), context=self)
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator or dict(
            type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1, strides=list(strides)))
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()
        self.init_weights()

    def _init_layers(self):
        f = self.feat_channels
        self.cls_convs = nn.ModuleList(ConvModuleShell(f, f, 3, padding=1, gn_groups=32) for _ in range(self.stacked_convs))
        self.reg_convs = nn.ModuleList(ConvModuleShell(f, f, 3, padding=1, gn_groups=32) for _ in range(self.stacked_convs))
        self.atss_cls = ConvShell(f, self.num_anchors * self.cls_out_channels, 3, padding=1, bias=True)
        self.atss_reg = ConvShell(f, self.num_anchors * 4, 3, padding=1, bias=True)
        self.atss_centerness = ConvShell(f, self.num_anchors * 1, 3, padding=1, bias=True)
        self.scales = nn.ModuleList(Scale(1.0) for _ in self.anchor_generator.strides)

    def init_weights(self):
        for m in list(self.cls_convs) + list(self.reg_convs):
            normal_(m.conv, std=0.01)
        normal_(self.atss_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_(self.atss_reg, std=0.01)
        normal_(self.atss_centerness, std=0.01)

    # ---- module API (routed through the owning detector's runtime)
    def _rt(self):
        from ..runtime import owner_runtime
        return owner_runtime(self)

    def forward(self, feats):
        return self._rt().head_forward_api(feats)

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, points_to_gt_index=None, points_weight=None,
                      gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        return self._rt().head_forward_train_api(x, img_metas, gt_bboxes, gt_labels, points_to_gt_index, points_weight,
                                                 proposal_cfg=proposal_cfg)

    def loss(self, cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, points_to_gt_index, points_weight,
             img_metas, gt_bboxes_ignore=None):
        return self._rt().head_loss_api(cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, points_to_gt_index,
                                        points_weight)

    def get_bboxes(self, cls_scores, bbox_preds, centernesses, img_metas, cfg=None, rescale=False, with_nms=True):
        return self._rt().get_bboxes_api(cls_scores, bbox_preds, centernesses, img_metas, cfg or self.test_cfg, rescale,
                                         with_nms)

    def get_anchors(self, featmap_sizes, img_metas, device="cuda"):
        anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        flags = [self.anchor_generator.valid_flags(featmap_sizes, m["pad_shape"], device) for m in img_metas]
        return [anchors for _ in img_metas], flags
