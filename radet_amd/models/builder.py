"""Model registries and builders -- same names and semantics as radet/models/builder.py:4-57."""
from torch import nn

from ..utils import Registry, build_from_cfg

BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
