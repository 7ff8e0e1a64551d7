"""Loss configuration holders (FocalLoss / GIoULoss / CrossEntropyLoss of radet/models/losses).
Inside the detector the three losses are evaluated by the fused HIP kernel radet_head_loss; these
classes carry the hyper-parameters from the config and validate them."""
from torch import nn

from .builder import LOSSES


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, "Only sigmoid focal loss supported now."
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight


@LOSSES.register_module()
class GIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not (use_sigmoid and use_mask)
        self.use_sigmoid, self.use_mask, self.reduction, self.class_weight, self.loss_weight = \
            use_sigmoid, use_mask, reduction, class_weight, loss_weight
