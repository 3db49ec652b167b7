"""BorderAlign with the reference's Python surface (slender_det/layers/border_align.py:9-43) on the HIP kernel."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._C import call, ptr, stream_ptr


class _BorderAlign(Function):
    @staticmethod
    def forward(ctx, input, boxes, wh, pool_size):
        input, boxes = input.contiguous().float(), boxes.contiguous().float()
        B, C4, H, W = input.shape
        K = boxes.shape[1]
        out = torch.empty((B, C4 // 4, K, 4), dtype=torch.float32, device=input.device)
        call("sod_border_align_fwd", ptr(input), ptr(boxes), ptr(out), B, C4 // 4, K, H, W, int(pool_size), stream_ptr())
        ctx.pool_size = pool_size
        ctx.save_for_backward(input, boxes)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, boxes = ctx.saved_tensors
        B, C4, H, W = input.shape
        grad = torch.zeros_like(input)
        call("sod_border_align_bwd", ptr(grad_output.contiguous().float()), ptr(input), ptr(boxes), ptr(grad), B, C4 // 4, boxes.shape[1], H, W,
             int(ctx.pool_size), stream_ptr())
        return grad, None, None, None


border_align = _BorderAlign.apply


class BorderAlign(nn.Module):
    def __init__(self, pool_size):
        super().__init__()
        self.pool_size = pool_size

    def forward(self, feature, boxes):
        feature, boxes = feature.contiguous(), boxes.contiguous()
        wh = (boxes[:, :, 2:] - boxes[:, :, :2]).contiguous()
        return border_align(feature, boxes, wh, self.pool_size)

    def __repr__(self):
        return self.__class__.__name__
