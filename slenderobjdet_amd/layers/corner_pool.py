"""CornerPool with the reference's surface (slender_det/layers/corner_pool.py:70-116) on the HIP scan kernels."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._C import call, ptr, stream_ptr

_MODES = {"bottom": 0, "top": 1, "right": 2, "left": 3}


class _CornerPoolFn(Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous().float()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        call("sod_corner_pool_fwd", ptr(x), ptr(y), N * C, H, W, mode, stream_ptr())
        ctx.mode = mode
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        dx = torch.zeros_like(x)
        call("sod_corner_pool_bwd", ptr(x), ptr(dy.contiguous().float()), ptr(dx), N * C, H, W, ctx.mode, 1, stream_ptr())
        return dx, None


class CornerPool(nn.Module):
    """mode in {'bottom', 'left', 'right', 'top'}."""

    def __init__(self, mode):
        super().__init__()
        assert mode in _MODES
        self.mode = mode

    def forward(self, x):
        return _CornerPoolFn.apply(x, _MODES[self.mode])
