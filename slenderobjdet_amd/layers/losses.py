"""Loss operators with the reference's signatures, running on the HIP kernels (autograd-enabled).

  sigmoid_focal_loss(_jit)  — fvcore.nn.sigmoid_focal_loss_jit as called at slender_det/modeling/meta_arch/fcos/fcosv2.py:124
  iou_loss                  — slender_det/layers/iou_loss.py:4-37
"""
import torch
from torch.autograd.function import once_differentiable

from . import functional as HF


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma, reduction):
        x = inputs.contiguous().float()
        shape = x.shape
        x2 = x.view(-1, shape[-1])
        if targets.dtype in (torch.int32, torch.int64) and targets.dim() == inputs.dim() - 1:
            labels, dense = targets.reshape(-1).to(torch.int32).contiguous(), None
        else:
            labels, dense = None, targets.contiguous().float().view(-1, shape[-1])
        s, elem = HF.focal_loss_fwd(x2, labels, dense, alpha, gamma, want_elem=(reduction == "none"))
        ctx.save_for_backward(x2, labels if labels is not None else dense)
        ctx.cfg = (alpha, gamma, reduction, labels is not None, shape)
        if reduction == "none":
            return elem.view(shape)
        return s[0] / x2.numel() if reduction == "mean" else s[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x2, t = ctx.saved_tensors
        alpha, gamma, reduction, is_labels, shape = ctx.cfg
        labels, dense = (t, None) if is_labels else (None, t)
        if reduction == "none":
            grad = HF.focal_loss_bwd(x2, labels, dense, alpha, gamma).view(shape) * g
        else:
            scale = g.reshape(1).float().contiguous()
            if reduction == "mean":
                scale = scale / x2.numel()
            grad = HF.focal_loss_bwd(x2, labels, dense, alpha, gamma, scale_num=scale).view(shape)
        return grad, None, None, None, None


def sigmoid_focal_loss(inputs, targets, alpha: float = -1, gamma: float = 2, reduction: str = "none"):
    """``targets``: float one-hot/soft tensor like ``inputs`` (the fvcore contract) or integer class indices of shape
    ``inputs.shape[:-1]`` (value outside [0, K) = background; avoids materialising the one-hot)."""
    return _FocalFn.apply(inputs, targets, float(alpha), float(gamma), reduction)


sigmoid_focal_loss_jit = sigmoid_focal_loss


class _IouLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, loss_type):
        p, t = pred.contiguous().float(), target.contiguous().float()
        w = weight.contiguous().float() if weight is not None else None
        s, _ = HF.iou_loss_fwd(p, t, w, loss_type)
        ctx.save_for_backward(p, t, w)
        ctx.loss_type = loss_type
        return s[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        p, t, w = ctx.saved_tensors
        return HF.iou_loss_bwd(p, t, w, ctx.loss_type, grad_scale=g.reshape(1).float().contiguous()), None, None, None


def iou_loss(pred, target, weight=None, loss_type="iou"):
    """Weighted SUM of the per-row IoU / linear-IoU / GIoU loss on (l, t, r, b) distances."""
    if loss_type not in HF.IOU_TYPES:
        raise NotImplementedError(loss_type)
    if weight is None:
        assert pred.shape[0] != 0
    return _IouLossFn.apply(pred, target, weight, loss_type)


class _BceSoftFn(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x[fg], t[fg], reduction="sum") with fg = rows whose label is a foreground class; the rows
    are selected by the label inside the kernel (no boolean gather)."""

    @staticmethod
    def forward(ctx, logits, targets, labels, bg_label):
        x, t, l = logits.contiguous().float(), targets.contiguous().float(), labels.contiguous().to(torch.int32)
        ctx.save_for_backward(x, t, l)
        ctx.bg = int(bg_label)
        return HF.bce_logits_soft_fwd(x, t, l, ctx.bg)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, t, l = ctx.saved_tensors
        return HF.bce_logits_soft_bwd(x, t, l, ctx.bg, g.reshape(1).float().contiguous()), None, None, None


def bce_with_logits_fg_sum(logits, targets, labels, bg_label):
    return _BceSoftFn.apply(logits, targets, labels, bg_label)
