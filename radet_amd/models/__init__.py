from .builder import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone, build_detector, build_head,
                      build_loss, build_neck)
from .fpn import FPN
from .losses import CrossEntropyLoss, FocalLoss, GIoULoss
from .radet import RADet
from .radet_head import RADetHead
from .resnet import ResNet

__all__ = ["BACKBONES", "NECKS", "HEADS", "LOSSES", "DETECTORS", "build_backbone", "build_neck", "build_head",
           "build_loss", "build_detector", "ResNet", "FPN", "RADetHead", "RADet", "FocalLoss", "GIoULoss",
           "CrossEntropyLoss"]
