"""Oracle building blocks (CPU fp32) in the same NHWC / KRSC layouts the HIP library uses.

Restates third-party arithmetic the reference reaches through detectron2 / ATen:
conv2d forward/backward, FrozenBatchNorm2d folding, GroupNorm(32)+ReLU, max-pool, nearest-2x upsample.
"parity unpinned" vs upstream; built only from torch built-ins (F.conv2d, F.group_norm, autograd).
"""
import torch
import torch.nn.functional as F


def rb(t):
    """Round to bf16 and back (the storage precision of activations/weights on the HIP side)."""
    return t.to(torch.bfloat16).to(torch.float32)


def to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def w_to_kcrs(w):   # (K,R,S,C) -> (K,C,R,S)
    return w.permute(0, 3, 1, 2).contiguous()


def w_to_krsc(w):   # (K,C,R,S) -> (K,R,S,C)
    return w.permute(0, 2, 3, 1).contiguous()


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1, res=None, relu=False, res_up2=False):
    """x NHWC, w KRSC -> y NHWC (fp32)."""
    y = F.conv2d(to_nchw(x), w_to_kcrs(w), bias, stride=stride, padding=pad, dilation=dil)
    y = to_nhwc(y)
    if res is not None:
        if res_up2:
            res = to_nhwc(F.interpolate(to_nchw(res), scale_factor=2, mode="nearest"))
        y = y + res
    if relu:
        y = torch.relu(y)
    return y


def conv2d_backward(x, w, dy, stride=1, pad=0, dil=1):
    """Returns (dx NHWC, dw KRSC) of y = conv2d(x, w)."""
    xn = to_nchw(x).requires_grad_(True)
    wk = w_to_kcrs(w).requires_grad_(True)
    y = F.conv2d(xn, wk, None, stride=stride, padding=pad, dilation=dil)
    dx, dw = torch.autograd.grad(y, (xn, wk), to_nchw(dy))
    return to_nhwc(dx), w_to_krsc(dw)


def group_norm(x, gamma, beta, groups, eps=1e-5, relu=False):
    y = to_nhwc(F.group_norm(to_nchw(x), groups, gamma, beta, eps))
    return torch.relu(y) if relu else y


def group_norm_backward(x, gamma, beta, groups, dy, eps=1e-5, relu=False):
    xn = x.clone().requires_grad_(True)
    g = gamma.clone().requires_grad_(True)
    b = beta.clone().requires_grad_(True)
    y = group_norm(xn, g, b, groups, eps, relu)
    return torch.autograd.grad(y, (xn, g, b), dy)


def frozen_bn_fold(weight, bias, mean, var, eps=1e-5):
    """d2 FrozenBatchNorm2d: y = x*scale + shift."""
    scale = weight * torch.rsqrt(var + eps)
    return scale, bias - mean * scale


def max_pool_3x3_s2(x):
    return to_nhwc(F.max_pool2d(to_nchw(x), kernel_size=3, stride=2, padding=1))


def upsample2x_backward(g):
    """Gradient of nearest-2x upsampling: sum over each 2x2 block."""
    N, H, W, C = g.shape
    return g.reshape(N, H // 2, 2, W // 2, 2, C).sum(dim=(2, 4))


def sgd_step(p, g, buf, lr, momentum, wd, nesterov=False, first=False):
    """torch.optim.SGD single-tensor update (what slender_det/solver/build.py:21-25 constructs)."""
    d = g + wd * p
    if momentum != 0:
        buf = d.clone() if first else momentum * buf + d
        d = d + momentum * buf if nesterov else buf
    return p - lr * d, buf


def preprocess(img, mean, std, Hp, Wp):
    """fcosv2.py:268-275 + ImageList.from_tensors: normalise then zero-pad; returns (Hp,Wp,3)."""
    x = (img.float() - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    out = torch.zeros(3, Hp, Wp)
    out[:, : x.shape[1], : x.shape[2]] = x
    return out.permute(1, 2, 0).contiguous()
