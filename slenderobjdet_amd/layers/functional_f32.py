"""fp32-STORAGE validation variants of the wrappers in :mod:`functional` (``SOD_PRECISION=fp32`` / ``functional.set_precision("fp32")``).

Same names and signatures as the bf16 wrappers they shadow; ``functional`` dispatches here when the validation mode is on.  Every
activation, compute copy of a weight and gradient is an fp32 tensor, every kernel is one of the ``sod_*_f32`` entry points of
``csrc/f32_path.hip`` (untuned, fixed-order reductions), everything runs on the current stream.  The mode exists for one purpose:
``tests/test_gpu_parity100.py`` asserts north_star's "total-loss delta < 1e-3 vs the reference's CPU path after 100 iterations"
in it (with bf16 storage any two runs drift by a few 1e-3 over 100 SGD steps).  It is never what ``bench.py`` times.

Arguments that only steer the bf16 kernels (``splits``, ``k_real`` / ``c_real``, ``relu_bits``) are accepted and ignored or refused.
"""
import ctypes

import torch

from .. import _C
from .._C import call, ptr, stream_ptr

F32 = torch.float32
CONV_RELU, CONV_RES_UP2 = 1, 2

# Validation-mode observer (tests only): when set, called as RELU_TAP(kind, key_ptr, y) with every tensor this mode produces THROUGH a
# ReLU - kind "conv" (key = the weight operand's address), "gn" (key = gamma's address), "relu" (key = 0).  tests/test_gpu_f32_mode.py
# hands the recorded decisions (y > 0) to the CPU oracle, so that both implementations differentiate the SAME piecewise-linear
# function and a whole-model gradient comparison is not decided by pre-activations that cancel to within rounding of zero.
RELU_TAP = None


def _chk(t, name="tensor", dtype=F32):
    if t is None:
        return
    if not t.is_cuda:
        raise _C.SlenderHipError(f"{name} must be a CUDA/HIP tensor (the HIP ops have no CPU fallback)")
    if not t.is_contiguous():
        raise _C.SlenderHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _C.SlenderHipError(f"{name} must be {dtype} in the fp32 validation mode, got {t.dtype}")


def conv_out_size(H, W, R, S, stride, pad, dil):
    return (H + 2 * pad - dil * (R - 1) - 1) // stride + 1, (W + 2 * pad - dil * (S - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------------------------- convolutions
def conv2d_fwd(x, w, bias=None, res=None, stride=1, pad=0, dil=1, relu=False, res_up2=False, out_f32=False, out=None, y_img_stride=0,
               x_img_stride=0, x_shape=None, c_real=None, relu_bits=None):
    if relu_bits is not None:
        raise _C.SlenderHipError("conv2d_fwd: 1-bit ReLU masks are a bf16-path device (the fp32 validation mode keeps the tensor)")
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias"); _chk(res, "res")
    N, H, W, C = x_shape if x_shape is not None else x.shape
    K, R, S, Cw = w.shape
    if Cw != C:
        raise _C.SlenderHipError(f"weight channels {Cw} != input channels {C}")
    Ho, Wo = conv_out_size(H, W, R, S, stride, pad, dil)
    if out is None:
        out = torch.empty((N, Ho, Wo, K), dtype=F32, device=x.device)
    flags = (CONV_RELU if relu else 0) | (CONV_RES_UP2 if res_up2 else 0)
    call("sod_conv2d_fwd_f32", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(out), N, H, W, C, K, R, S, stride, pad, dil, x_img_stride, y_img_stride,
         flags, stream_ptr())
    if relu and RELU_TAP is not None:
        RELU_TAP("conv", w.data_ptr(), out)
    return out


def conv2d_dgrad(dy, wt, x_hw, stride=1, pad=0, dil=1, accum=None, relu_mask=None, dy_img_stride=0, dy_shape=None, out=None, relu_bits=None,
                 accum_even=False):
    """``wt`` is the KRSC fp32 compute copy here (the fp32 data-gradient kernel needs no transposed weights)."""
    if relu_bits is not None:
        raise _C.SlenderHipError("conv2d_dgrad: 1-bit ReLU masks are a bf16-path device")
    _chk(dy, "dy"); _chk(wt, "w"); _chk(accum, "accum"); _chk(relu_mask, "relu_mask")
    N = dy_shape[0] if dy_shape is not None else dy.shape[0]
    K, R, S, C = wt.shape
    H, W = x_hw
    if out is None:
        out = torch.empty((N, H, W, C), dtype=F32, device=dy.device)
    if accum_even and (accum is None or tuple(accum.shape) != (N, H // 2, W // 2, C) or H % 2 or W % 2):
        raise _C.SlenderHipError("conv2d_dgrad: accum_even needs accum of shape (N, H/2, W/2, C) with even H, W")
    call("sod_conv2d_dgrad_f32", ptr(dy), ptr(wt), ptr(accum), ptr(relu_mask), ptr(out), N, H, W, C, K, R, S, stride, pad, dil, dy_img_stride,
         1 if accum_even else 0, stream_ptr())
    return out


def conv2d_wgrad(dy, x, dw, R, S, stride=1, pad=0, dil=1, dy_img_stride=0, x_img_stride=0, K=None, x_shape=None, splits=0, qscale=None,
                 k_real=None, c_real=None):
    _chk(dy, "dy"); _chk(x, "x"); _chk(dw, "dw"); _chk(qscale, "qscale")
    N, H, W, C = x_shape if x_shape is not None else x.shape
    if K is None:
        K = dy.shape[-1]
    call("sod_conv2d_wgrad_f32", ptr(dy), ptr(x), ptr(dw), ptr(qscale), N, H, W, C, K, R, S, stride, pad, dil, dy_img_stride, x_img_stride, stream_ptr())
    return dw


def conv2d_fwd_ml(xs, w, bias=None, stride=1, pad=0, dil=1, relu=False, out_f32=False, outs=None, y_img_stride=0, k_real=None):
    if outs is None:
        outs = [None] * len(xs)
    return [conv2d_fwd(x, w, bias, None, stride, pad, dil, relu=relu, out=o, y_img_stride=y_img_stride) for x, o in zip(xs, outs)]


def conv2d_dgrad_ml(dys, wt, x_hws, stride=1, pad=0, dil=1, dy_img_stride=0, N=None, k_real=None, relu_masks=None):
    if N is None:
        N = dys[0].shape[0]
    masks = relu_masks if relu_masks is not None else [None] * len(dys)
    return [conv2d_dgrad(dy, wt, hw, stride, pad, dil, relu_mask=m, dy_img_stride=dy_img_stride, dy_shape=(N,)) for dy, hw, m in zip(dys, x_hws, masks)]


def conv2d_wgrad_ml(dys, xs, dw, R, S, stride=1, pad=0, dil=1, dy_img_stride=0, K=None, splits=0, qscale=None, k_real=None):
    if K is None:
        K = dys[0].shape[-1]
    for dy, x in zip(dys, xs):
        conv2d_wgrad(dy, x, dw, R, S, stride, pad, dil, dy_img_stride=dy_img_stride, K=K, qscale=qscale)
    return dw


def weight_prep(w_master, scale=None, want_krsc=True, want_crsk=True, cpad=None):
    """fp32 (K,R,S,C) -> fp32 (K,R,S,Cpad) [* scale[k]]; returned twice: the data gradient reads the same KRSC copy."""
    _chk(w_master, "w"); _chk(scale, "scale")
    K, R, S, C = w_master.shape
    cpad = C if cpad is None else cpad
    wk = torch.empty((K, R, S, cpad), dtype=F32, device=w_master.device)
    call("sod_weight_prep_f32", ptr(w_master), ptr(scale), ptr(wk), None, K, R * S, C, cpad, stream_ptr())
    return wk, wk


# ----------------------------------------------------------------------------------------------- GroupNorm
def groupnorm_fwd(x, gamma, beta, G, eps=1e-5, relu=False, stats=None):
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    if stats is None:
        stats = torch.empty((N, G, 2), dtype=F32, device=x.device)
    call("sod_groupnorm_fwd_f32", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), N, HW, C, G, eps, 1 if relu else 0, stream_ptr())
    if relu and RELU_TAP is not None:
        RELU_TAP("gn", gamma.data_ptr(), y)
    return y, stats


def groupnorm_bwd(dy, x, gamma, beta, stats, G, dgamma, dbeta, relu=False, dxsum=None):
    _chk(dy, "dy"); _chk(x, "x"); _chk(dgamma, "dgamma"); _chk(dbeta, "dbeta"); _chk(stats, "stats"); _chk(dxsum, "dxsum")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    call("sod_groupnorm_bwd_f32", ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(stats), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dxsum), N, HW, C, G,
         1 if relu else 0, stream_ptr())
    return dx


def groupnorm_fwd_ml(xs, gamma, beta, G, eps=1e-5, relu=False):
    N = xs[0].shape[0]
    stats = torch.empty((len(xs), N, G, 2), dtype=F32, device=xs[0].device)
    ys = []
    for l, x in enumerate(xs):
        ys.append(groupnorm_fwd(x, gamma, beta, G, eps, relu, stats=stats[l])[0])      # the level's (N, G, 2) slice of the stats tensor
    return ys, stats


def groupnorm_bwd_ml(dys, xs, gamma, beta, stats, G, dgamma, dbeta, relu=False, dxsum=None):
    return [groupnorm_bwd(dy, x, gamma, beta, stats[l], G, dgamma, dbeta, relu, dxsum) for l, (dy, x) in enumerate(zip(dys, xs))]


# ----------------------------------------------------------------------------------------------- element-wise
def _elt(op, a, b=None):
    _chk(a, "a"); _chk(b, "b")
    o = torch.empty_like(a)
    call("sod_eltwise_f32", op, ptr(a), ptr(b), ptr(o), a.numel(), stream_ptr())
    return o


def relu_fwd(x):
    y = _elt(0, x)
    if RELU_TAP is not None:
        RELU_TAP("relu", 0, y)
    return y


def relu_bwd(dy, y):
    return _elt(1, dy, y)


def add_bf16(a, b):
    return _elt(2, a, b)


def f32_to_bf16(x):
    """The bf16 path converts fp32 output gradients for its MFMA kernels here; the validation mode keeps them."""
    return x


def add_up2(a, b):
    _chk(a, "a"); _chk(b, "b")
    N, H, W, C = a.shape
    if tuple(b.shape) != (N, H // 2, W // 2, C) or H % 2 or W % 2:
        raise _C.SlenderHipError(f"add_up2: {tuple(a.shape)} is not the 2x upsampling of {tuple(b.shape)}")
    o = torch.empty_like(a)
    call("sod_add_up2_f32", ptr(a), ptr(b), ptr(o), N, H, W, C, stream_ptr())
    return o


def upsample2x_bwd(g):
    _chk(g, "g")
    N, H, W, C = g.shape
    d = torch.empty((N, H // 2, W // 2, C), dtype=F32, device=g.device)
    call("sod_upsample2x_bwd_f32", ptr(g), ptr(d), N, H // 2, W // 2, C, stream_ptr())
    return d


def maxpool3x3s2(x):
    _chk(x, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C), dtype=F32, device=x.device)
    call("sod_maxpool3x3s2_f32", ptr(x), ptr(y), N, H, W, C, stream_ptr())
    return y


def bias_grad(dy, dbias, N, HW, C, img_stride=0):
    _chk(dy, "dy"); _chk(dbias, "dbias")
    call("sod_bias_grad_f32", ptr(dy), ptr(dbias), N, HW, C, img_stride, stream_ptr())
    return dbias


def bias_grad_ml(dys, dbias):
    N, C = dys[0].shape[0], dys[0].shape[-1]
    for g in dys:
        bias_grad(g, dbias, N, g.numel() // (N * C), C)
    return dbias


# ----------------------------------------------------------------------------------------------- input
def preprocess_image(img, out, mean, std):
    if img.dtype not in (torch.uint8, torch.float32):
        raise _C.SlenderHipError("image must be uint8 or float32")
    _chk(img, "image", None); _chk(out, "out")
    C, H, W = img.shape
    Hp, Wp, Cp = out.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    call("sod_preprocess_image_f32", ptr(img), 1 if img.dtype == torch.uint8 else 0, C, H, W, ptr(out), Hp, Wp, Cp, ctypes.cast(m, ctypes.c_void_p),
         ctypes.cast(s, ctypes.c_void_p), stream_ptr())
    return out


def preprocess_batch(imgs, out, mean, std):
    for i, im in enumerate(imgs):
        preprocess_image(im.contiguous() if im.dtype in (torch.uint8, torch.float32) else im.float().contiguous(), out[i], mean, std)
    return out
