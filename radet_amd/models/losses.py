"""FocalLoss / GIoULoss / CrossEntropyLoss with the reference's constructors and `forward` signatures
(radet/models/losses/{focal_loss.py:90-157, iou_loss.py:320-354, cross_entropy_loss.py:128-201}, reduction rules of
losses/utils.py:24-51), evaluated by the stand-alone HIP loss kernels (csrc/boxops.hip) with hand-written
backward passes (torch.autograd.Function).  Inside the detector's train step the three losses run fused in
`radet_head_loss`; these modules serve callers that use a loss on its own (e.g. `RADetHead.loss_cls(...)`).

Inputs may live on the host or the GPU; the arithmetic always runs on the GPU and the result comes back on the input's
device.  There is no CPU fallback."""
import torch
from torch import nn

from .. import kernels as K
from .builder import LOSSES


def _dev():
    if not torch.cuda.is_available():
        from .._lib import RadetHipError
        raise RadetHipError("radet_amd losses need an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _f32(t, dev):
    return t.detach().to(dev, torch.float32).contiguous()


def _avg_tensor(avg_factor, dev):
    if avg_factor is None:
        return None
    if isinstance(avg_factor, torch.Tensor):
        return avg_factor.detach().to(dev, torch.float32).reshape(1).contiguous()
    return torch.tensor([float(avg_factor)], device=dev)


def _resolve(reduction, avg_factor, n_elem, loss_weight):
    """-> (reduce?, scale); scale multiplies sum / avg_factor (reduced) or each element (none)"""
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(f"{reduction} is not a valid value for reduction")
    if avg_factor is not None and reduction == "sum":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    if reduction == "none":
        return False, float(loss_weight)
    if reduction == "mean" and avg_factor is None:          # loss.mean(): nan for an empty tensor, like torch
        return True, (float(loss_weight) / n_elem) if n_elem else float("nan")
    return True, float(loss_weight)


class _ElemLoss(torch.autograd.Function):
    """loss = reduce(kind(pred, target) * weight) through the HIP kernels; backward w.r.t. pred only."""

    @staticmethod
    def forward(ctx, pred, kind, target, weight, wcols, reduction, avg_factor, loss_weight, hp):
        src = pred.device
        dev = _dev()
        x = _f32(pred, dev)
        shape = tuple(pred.shape)
        if kind == "giou":
            N, C = x.shape[0], 1
            x = x.reshape(N, 4)
            tgt = _f32(target, dev).reshape(N, 4)
            n_elem = N
        else:
            C = shape[-1] if len(shape) > 1 else 1
            N = x.numel() // max(C, 1)
            x = x.reshape(N, C)
            tgt = target.detach().to(dev, torch.long).contiguous() if kind == "focal" else _f32(target, dev).reshape(N, C)
            n_elem = N * C
        w = None if weight is None else _f32(weight, dev)
        avg = _avg_tensor(avg_factor if reduction == "mean" else None, dev)
        reduce_, scale = _resolve(reduction, avg_factor, n_elem, loss_weight)
        elem = None if reduce_ else torch.empty(N if kind == "giou" else (N, C), device=dev)
        partials = torch.empty(K.loss_partials(n_elem), device=dev) if reduce_ else None
        if kind == "focal":
            K.sigmoid_focal_loss(x, tgt, w, wcols, N, C, hp["gamma"], hp["alpha"], elem, partials, 1.0 if reduce_ else scale)
        elif kind == "bce":
            K.bce_logits_loss(x, tgt, w, wcols, N, C, elem, partials, 1.0 if reduce_ else scale)
        else:
            K.giou_loss(x, tgt, w, N, hp["eps"], elem, partials, 1.0 if reduce_ else scale)
        if reduce_:
            out = torch.empty(1, device=dev)
            K.loss_finalize(partials, avg, scale, out)
            out = out.reshape(())
        else:
            out = elem.reshape(shape[:-1] if kind == "giou" else shape)
        ctx.save_for_backward(x, tgt, w if w is not None else x.new_empty(0), avg if avg is not None else x.new_empty(0))
        ctx.meta = (kind, wcols, N, C, reduce_, scale, hp, shape, src, w is not None, avg is not None)
        return out.to(src)

    @staticmethod
    def backward(ctx, g):
        kind, wcols, N, C, reduce_, scale, hp, shape, src, has_w, has_avg = ctx.meta
        x, tgt, w, avg = ctx.saved_tensors
        w = w if has_w else None
        avg = avg if has_avg else None
        dev = x.device
        g = _f32(g, dev)
        ge, gs = (None, g.reshape(1)) if reduce_ else (g.reshape(-1), None)
        dx = torch.empty_like(x)
        if kind == "focal":
            K.sigmoid_focal_loss_bwd(x, tgt, w, wcols, N, C, hp["gamma"], hp["alpha"], ge, gs, avg, scale, dx)
        elif kind == "bce":
            K.bce_logits_loss_bwd(x, tgt, w, wcols, N, C, ge, gs, avg, scale, dx)
        else:
            K.giou_loss_bwd(x, tgt, w, N, hp["eps"], ge, gs, avg, scale, dx)
        return (dx.reshape(shape).to(src),) + (None,) * 8


def _weight_cols(weight, n_rows, n_cols):
    """focal_loss.py:67-81: a weight per row, or per element (possibly flattened)"""
    if weight is None:
        return None, 0
    if weight.numel() == n_rows * n_cols and n_cols > 1 and tuple(weight.shape) != (n_rows,):
        return weight.reshape(n_rows, n_cols), n_cols
    if weight.numel() == n_rows:
        return weight.reshape(n_rows), 1
    raise AssertionError("weight must have one entry per row or per element")


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, "Only sigmoid focal loss supported now."
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """pred [N, C] logits, target i64 [N] (C = background), weight [N] | [N, C] | [N*C]"""
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        w, wcols = _weight_cols(weight, pred.shape[0], pred.shape[1])
        return _ElemLoss.apply(pred, "focal", target, w, wcols, reduction, avg_factor, self.loss_weight,
                               dict(gamma=float(self.gamma), alpha=float(self.alpha)))


@LOSSES.register_module()
class GIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        """pred / target [n, 4] boxes (x1, y1, x2, y2), weight [n] (or [n, 4], averaged like the reference)"""
        if weight is not None and not torch.any(weight > 0):          # iou_loss.py:335-336 (host sync there too)
            return pred.sum() * 0.0
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return _ElemLoss.apply(pred, "giou", target, weight, 1 if weight is not None else 0, reduction, avg_factor,
                               self.loss_weight, dict(eps=float(self.eps)))


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """use_sigmoid=True (binary cross entropy with logits: the head's `loss_centerness` / `loss_iou`) runs on the GPU;
    the softmax and mask variants are not on the RADet path and raise."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not (use_sigmoid and use_mask)
        self.use_sigmoid, self.use_mask, self.reduction, self.class_weight, self.loss_weight = \
            use_sigmoid, use_mask, reduction, class_weight, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if not self.use_sigmoid or self.class_weight is not None:
            raise NotImplementedError("CrossEntropyLoss on MI355X: only use_sigmoid=True without class_weight (the RADet "
                                      "head's IoU / centerness loss) is implemented")
        if cls_score.dim() != label.dim():
            # cross_entropy_loss.py:41-54 `_expand_onehot_labels`: class indices -> one-hot rows, weight per row
            C = cls_score.size(-1)
            onehot = torch.zeros(label.size(0), C, device=label.device)
            inds = ((label >= 0) & (label < C)).nonzero(as_tuple=False).reshape(-1)
            if inds.numel() > 0:
                onehot[inds, label[inds]] = 1
            label = onehot
        n_cols = cls_score.shape[-1] if cls_score.dim() > 1 else 1
        n_rows = cls_score.numel() // max(n_cols, 1)
        w, wcols = _weight_cols(weight, n_rows, n_cols) if weight is not None else (None, 0)
        if cls_score.dim() == 1 and weight is not None:
            w, wcols = weight.reshape(-1), 1
        return _ElemLoss.apply(cls_score, "bce", label, w, wcols, reduction, avg_factor, self.loss_weight, {})
