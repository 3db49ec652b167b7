"""Oracle building blocks (CPU fp32) in the same NHWC / KRSC layouts the HIP library uses.

Restates third-party arithmetic the reference reaches through detectron2 / ATen:
conv2d forward/backward, FrozenBatchNorm2d folding, GroupNorm(32)+ReLU, max-pool, nearest-2x upsample.
"parity unpinned" vs upstream; built only from torch built-ins (F.conv2d, F.group_norm, autograd).
"""
import torch
import torch.nn.functional as F


class ReluBand:
    """Test-side handling of ReLU's discrete decisions (tests/test_gpu_f32_mode.py).  Two correct fp32 implementations
    of one network disagree on the sign of a pre-activation that cancels to within their rounding of zero; the forward value is ~0
    either way, but the backward mask differs, and where the incoming gradient is sparse (the regression branch: positives only) one
    such unit moves a whole weight-gradient tensor by 1e-3 ... 1e-2 of its norm.  While ``ReluBand.active`` is set, every ``relu`` of
    the whole-model oracles records the units with |x| < tau * rms(x) (the UNDECIDED band); backward then uses, by ``mode``, the
    natural mask (0), the mask with every undecided unit ON (+1) or OFF (-1).  One forward pass and three backward passes give the
    gradient and an envelope of what flipping undecided units can do to each tensor: the tolerance of a comparison is then wide
    exactly where - and only where - an undecided unit sits in a sensitive place."""
    active = None      # None, or {"tau": float, "mode": 0 | +1 | -1, "count": int, "units": int}


class _ReluBandFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        st = ReluBand.active
        xd = x.detach()
        band = xd.abs() < st["tau"] * xd.pow(2).mean().sqrt()
        st["count"] += int(band.sum())
        st["units"] += xd.numel()
        ctx.save_for_backward(xd > 0, band)
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        pos, band = ctx.saved_tensors
        mode = ReluBand.active["mode"] if ReluBand.active is not None else 0
        m = pos if mode == 0 else ((pos | band) if mode > 0 else (pos & ~band))
        return g * m.to(g.dtype)


def relu(x):
    """torch.relu, or (inside a ReluBand context) the same forward with a switchable backward mask for the undecided units."""
    return _ReluBandFn.apply(x) if ReluBand.active is not None else torch.relu(x)


class ForcedMasks:
    """Teacher-forced ReLU decisions: while ``active`` = {"masks": {(key, i): bool tensor shaped like the activation}, ...} is set, the
    i-th ``relu_at(x, key)`` call returns x * mask - the decisions of ANOTHER implementation of the same network (the HIP validation
    mode: layers/functional_f32.RELU_TAP) - so that both differentiate the same piecewise-linear function.  Keys without a mask fall
    back to ``relu`` and are listed in ``missed``."""
    active = None

    @staticmethod
    def begin(masks, record=False, tau=1e-4):
        """``record=True``: natural decisions everywhere, and every position's (x > 0) is stored into ``masks`` (a dict).
        ``tau``: a forced decision may differ from the oracle's OWN (x > 0) only on units inside the undecided band |x| < tau * rms(x)
        of that activation (ReluBand's band) - a pre-activation both implementations hold to rounding of zero.  ``disagree`` counts the
        units decided differently, ``outside`` those of them OUTSIDE the band (a product pre-activation with the wrong sign, which the
        forced mask would otherwise hide by producing 0 on both sides); ``outside_at`` lists the positions.  Callers assert outside == 0."""
        ForcedMasks.active = {"masks": masks, "count": {}, "used": 0, "missed": [], "record": bool(record), "tau": float(tau), "units": 0,
                              "disagree": 0, "outside": 0, "outside_at": []}
        return ForcedMasks.active

    @staticmethod
    def end():
        st, ForcedMasks.active = ForcedMasks.active, None
        return st


def relu_at(x, key):
    """``relu(x)`` at the network position ``key`` (a module name; shared modules are called once per level, in level order)."""
    fm = ForcedMasks.active
    if fm is not None:
        i = fm["count"].get(key, 0)
        fm["count"][key] = i + 1
        if fm["record"]:
            fm["masks"][(key, i)] = x.detach() > 0
            return relu(x)
        m = fm["masks"].get((key, i))
        if m is not None:
            if tuple(m.shape) != tuple(x.shape):
                if m.numel() != x.numel() or [d for d in m.shape if d != 1] != [d for d in x.shape if d != 1]:
                    raise ValueError(f"forced mask for {key}#{i} has shape {tuple(m.shape)}, activation {tuple(x.shape)}")
                m = m.reshape(x.shape)          # (R, C, 1, 1) rows of a fully connected layer run as a 1x1 convolution
            fm["used"] += 1
            m = m.to(device=x.device, dtype=torch.bool)
            xd = x.detach()
            dis = (xd > 0) != m                 # decided differently by the other implementation
            nd = int(dis.sum())
            fm["units"] += xd.numel()
            if nd:
                fm["disagree"] += nd
                rms = xd.double().pow(2).mean().sqrt().clamp_min(1e-300)
                out = dis & (xd.abs().double() >= fm["tau"] * rms)
                no = int(out.sum())
                if no:
                    fm["outside"] += no
                    fm["outside_at"].append((key, i, no, float((xd.abs().double()[out] / rms).max())))
            return x * m.to(x.dtype)
        fm["missed"].append((key, i))
    return relu(x)


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
