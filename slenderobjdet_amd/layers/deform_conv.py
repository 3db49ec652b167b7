"""Deformable convolution modules with the detectron2 / reference names:
``DeformConv`` / ``ModulatedDeformConv`` (detectron2.layers, source absent) and ``DFConv2d``
(slender_det/layers/df_conv.py:6-78).  NHWC bf16 activations, fp32 NHWC offsets (channel 2k = dy, 2k+1 = dx).
"""
import math

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import functional as HF
from .nn import HipConv2d, _arena_of


import os

SAVE_COLS = True
# Fused kernels (csrc/dcn_fused.hip): forward and weight gradient gather the samples into LDS tiles inside the MFMA loop - no
# (N*Ho*Wo, KH*KW*C) column buffer in HBM.  Measured per 256->256 DCN layer over the 5 RepPoints levels (batch 16): forward
# 1.51 -> 0.95 ms, weight gradient 0.53 ms (on kept columns) -> 0.79 ms, i.e. 2.04 -> 1.74 ms and 1.65 GB less memory traffic and
# footprint per layer.  SOD_DCN_FUSED=0 restores the column-buffer path (shapes the fused kernels do not take use it anyway).
FUSED = os.environ.get("SOD_DCN_FUSED", "1") != "0"
# The gradient w.r.t. input / offsets / mask: sod_deform_conv_bwd_fused computes each tile's column gradients on the matrix cores inside
# the scatter kernel; SOD_DCN_BWD_FUSED=0 restores 1x1 dgrad -> bf16 dcols (N*Ho*Wo, 9C) in HBM -> dcn_col2im_tile.
BWD_FUSED = os.environ.get("SOD_DCN_BWD_FUSED", "1") != "0"


# The fused backward keeps dX in an LDS window around each 8x8 output tile; samples outside it are added to HBM with float atomics, eight
# parked lanes per wave instruction (csrc/deform_conv.hip).  The P3 level of RepPoints takes 2.5 / 3.2 / 4.5 / 7.0 ms at offset spreads of
# 0.5 / 2 / 4 / 8 px with 2 px of slack (tools/bench_dcn_bwd_window.py, profiles/r5_dcn_bwd_window.txt: 0 / 4 / 24 / 57 % of the samples
# outside; 2.5 / 5.3 / 18 / 39 ms while every lane added its own 32 scattered atomics, rounds 3-4).  Trained RepPoints offsets reach a few
# pixels per level (rpd.py:637-647: the points of an object of 4-8 strides), so the slack is ADAPTIVE per layer and level: the library
# counts the samples that left the window (sod_deform_conv_set_window_counter), the count of the previous launch is read back
# asynchronously, and the slack goes 2 -> 4 px while more than 10 % of the samples are outside - a wider window admits fewer workgroups
# per CU (2.5 / 2.8 ms with nothing outside), so it is only paid where the offsets ask for it (4.5 -> 3.6 ms at 4 px, 6.9 -> 4.7 ms at
# 8 px: from 4 px of slack the library also stages the offset gradients in LDS); a step that takes back less than 15 % of the outside samples is undone (a diverging run's offsets are beyond any window); every
# PROBE_EVERY launches the narrower window is tried again.  SOD_DCN_ADAPTIVE_WINDOW=0 switches it off.
ADAPTIVE_WINDOW = os.environ.get("SOD_DCN_ADAPTIVE_WINDOW", "1") != "0"
PROBE_EVERY = 200


class _WindowPolicy:
    """Slack of the fused backward's LDS window for ONE (layer, level)."""
    MAX_SLACK = 4

    def __init__(self, device):
        self.slack = 2
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.event = None
        self.pending_slack = 2
        self.calls = 0
        self.probing = False
        self.last_share = 0.0
        self.widened_from = None      # share at the narrower window while the first count at the widened one is pending
        self.hold = 0                 # launches left before widening may be tried again

    def poll(self, lanes):
        """Folds the previous launch's count into the slack, if its read-back has completed (never waits)."""
        if self.event is None or not self.event.query():
            return
        self.event = None
        share = float(self.host[0]) / max(lanes, 1)
        self.last_share = share
        if self.pending_slack != self.slack:        # a count taken at another slack says nothing about this one
            return
        if self.widened_from is not None:           # first count at a widened window: did it pay?
            before, self.widened_from = self.widened_from, None
            if share > 0.85 * before:                # the offsets are far beyond ANY window (a diverging run): the wide window only costs
                self.slack -= 2                      # workgroups per CU - back, and no further attempt for a while
                self.hold = PROBE_EVERY
                return
        # break-even share from tools/bench_dcn_bwd_window.py (P3 level of RepPoints, profiles/r5_dcn_bwd_window.txt: ms at slack 2 / 4 / 6 =
        # 2.5 / 3.1 / 5.5 with nothing outside, + ~0.085 ms per per cent of the samples outside since the wave adds them (0.65 before); a
        # step of the slack takes 3-8x of them back in: slack 4 pays from ~10 % outside, slack 6 never does any more)
        if share > 0.10 and self.slack < self.MAX_SLACK and self.hold == 0:
            self.widened_from = share
            self.slack += 2
            self.probing = False
        elif self.probing:
            self.probing = False                     # the narrower window held: keep it

    def before(self, C, K, dg, k, stride, dil):
        """Slack for this launch (lowered until the library accepts the layer at it); registers the counter.  Returns the slack or None."""
        self.calls += 1
        self.hold = max(0, self.hold - 1)
        if self.slack > 2 and not self.probing and self.calls % PROBE_EVERY == 0:
            self.slack -= 2
            self.probing = True
        r = self.slack
        while r >= 0:
            HF.call("sod_deform_conv_set_window_slack", r)
            if HF.deform_bwd_fused_supported(C, K, dg, (k, k), stride, dil):
                break
            r -= 2
        if r < 0:
            HF.call("sod_deform_conv_set_window_slack", -1)
            return None
        self.slack = min(self.slack, r)             # the layer does not fit at the wider window: stay where it does
        self.pending_slack = r
        HF.call("sod_deform_conv_set_window_counter", HF.ptr(self.counter))
        return r

    def after(self):
        HF.call("sod_deform_conv_set_window_counter", None)
        HF.call("sod_deform_conv_set_window_slack", -1)
        if self.event is None:                       # one read-back in flight at a time
            self.host.copy_(self.counter, non_blocking=True)
            self.counter.zero_()
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.counter.zero_()


def _ceil8(v):
    return (v + 7) // 8 * 8


class _DeformConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, mod, off_ld, mask_ld, mask_is_logit):
        k = mod.kernel_size
        fused = FUSED and HF.deform_fused_supported(x.shape[3], mod.out_channels, mod.deformable_groups)
        cols = None
        if fused:
            y = HF.deform_conv_fwd_fused(x, offset, mask, mod.w_bf16, mod.bias_eff, (k, k), mod.stride, mod.padding, mod.dilation,
                                         mod.deformable_groups, off_ld, mask_ld, mask_is_logit, relu=mod.relu)
        else:
            cols = HF.deform_im2col(x, offset, mask, (k, k), mod.stride, mod.padding, mod.dilation, mod.deformable_groups, off_ld, mask_ld, mask_is_logit)
            y = HF.conv2d_fwd(cols, mod.w_bf16, mod.bias_eff, None, 1, 0, 1, relu=mod.relu)
        ctx.mod, ctx.cfg, ctx.fused = mod, (off_ld, mask_ld, mask_is_logit), fused
        # Column-buffer path only: the sampled columns are needed again by the weight gradient.  They are KEPT across the step
        # (2 DCN layers x 5 levels of RepPoints at batch 16: 3.3 GB of the 288 GB) instead of being re-gathered in backward;
        # SAVE_COLS = False restores the recomputation.
        keep_cols = SAVE_COLS and mod.weight.requires_grad and cols is not None
        ctx.save_for_backward(x, offset, mask, y if mod.relu else None, cols if keep_cols else None)
        arena = _arena_of(mod)
        if arena is not None and mod.weight.requires_grad:
            arena.note_use(mod.weight)
            if mod.bias is not None:
                arena.note_use(mod.bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        mod = ctx.mod
        off_ld, mask_ld, mask_is_logit = ctx.cfg
        x, offset, mask, y, cols = ctx.saved_tensors
        dy = dy.contiguous()
        if mod.relu:
            dy = HF.relu_bwd(dy, y)
        k, dg = mod.kernel_size, mod.deformable_groups
        arena = _arena_of(mod)
        N, Ho, Wo, K = dy.shape
        C = x.shape[3]
        if mod.weight.requires_grad:
            grouped = getattr(mod, "groups", 1) > 1
            # groups > 1: dW of the dense embedding into a scratch tensor on the CURRENT stream, its diagonal blocks added to the arena gradient
            dw = torch.zeros((K, 1, 1, k * k * C), dtype=torch.float32, device=dy.device) if grouped else arena.grad_view(mod.weight).view(K, 1, 1, k * k * C)
            prev_side = HF.WGRAD_SIDE_STREAM
            if grouped:
                HF.WGRAD_SIDE_STREAM = False
            try:
                if ctx.fused:
                    HF.deform_conv_wgrad_fused(dy, x, offset, mask, dw, (k, k), mod.stride, mod.padding, mod.dilation, dg, off_ld, mask_ld, mask_is_logit,
                                               qscale=mod.bn_scale)
                else:
                    if cols is None:
                        cols = HF.deform_im2col(x, offset, mask, (k, k), mod.stride, mod.padding, mod.dilation, dg, off_ld, mask_ld, mask_is_logit)
                    HF.conv2d_wgrad(dy, cols, dw, 1, 1, 1, 0, 1, qscale=mod.bn_scale)
                    del cols
            finally:
                HF.WGRAD_SIDE_STREAM = prev_side
            if grouped:
                arena.grad_view(mod.weight).add_(mod.blocks_of(dw.view(K, k, k, C)))
            arena.mark_ready(mod.weight)
            if mod.bias is not None:
                HF.bias_grad(dy, arena.grad_view(mod.bias), N, Ho * Wo, K)
                arena.mark_ready(mod.bias)
        doff = torch.zeros_like(offset)          # pitched like the offset tensor (mask columns live in the same rows for v2)
        dmask = None
        aliased = mask is not None and mask.untyped_storage().data_ptr() == offset.untyped_storage().data_ptr()
        if mask is not None:
            dmask = doff.view(-1)[mask.storage_offset() - offset.storage_offset():] if aliased else torch.zeros_like(mask)
        pol, slack = None, None
        if BWD_FUSED and ADAPTIVE_WINDOW and not HF.is_f32():
            pols = mod.__dict__.setdefault("_window_policies", {})
            pol = pols.get((N, Ho, Wo))
            if pol is None:
                pol = pols[(N, Ho, Wo)] = _WindowPolicy(dy.device)
            pol.poll(N * Ho * Wo * k * k * (C // 8))
            slack = pol.before(C, K, dg, k, mod.stride, mod.dilation)
        if BWD_FUSED and (slack is not None or (pol is None and HF.deform_bwd_fused_supported(C, K, dg, (k, k), mod.stride, mod.dilation))):
            # one pass: the tile's slice of dcols = dY x W^T is computed inside the scatter kernel (no column-gradient tensor)
            try:
                dx32 = HF.deform_conv_bwd_fused(dy, mod.wt_bf16, x, offset, mask, (k, k), mod.stride, mod.padding, mod.dilation, dg, doff, dmask,
                                                off_ld, mask_ld, mask_is_logit)
            finally:
                if pol is not None:
                    pol.after()
        else:
            dcols = HF.conv2d_dgrad(dy, mod.wt_bf16, (Ho, Wo), 1, 0, 1)
            dx32 = HF.deform_col2im(dcols, x, offset, mask, (k, k), mod.stride, mod.padding, mod.dilation, dg, doff, dmask, off_ld, mask_ld, mask_is_logit)
        dx = HF.f32_to_bf16(dx32) if ctx.needs_input_grad[0] else None
        gmask = None
        if mask is not None and ctx.needs_input_grad[2] and not aliased:
            gmask = dmask if dmask.shape == mask.shape else None
        # (a mask that is a VIEW into the offset tensor's rows - DFConv2d, DeformBottleneckBlock - got its gradient written into ``doff``'s
        # mask columns above; returning it a second time through the view would make autograd add those columns - and, the view being
        # one-dimensional, everything behind them - to the offset tensor's gradient twice)
        return dx, doff, gmask, None, None, None, None, None


class DeformConv(nn.Module):
    """detectron2.layers.DeformConv(in, out, kernel_size, stride, padding, dilation, groups, deformable_groups, bias=False).
    ``relu`` (extension): fuse the ReLU that follows the layer (rpd.py:161-167) into the GEMM epilogue."""
    modulated = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=False,
                 relu=False, frozen_bn=False):
        super().__init__()
        self.relu = relu
        # detectron2's ``groups`` splits the GEMM into independent channel groups (ResNeXt + DCN: DeformBottleneckBlock passes
        # RESNETS.NUM_GROUPS).  No config of the reference sets it, so it takes the simple exact route of HipGroupedConv2d: the master weight
        # has the reference's grouped shape (K, k, k, C / groups), the compute copies are its block-diagonal embedding into a dense weight
        # (every kernel below runs unchanged; groups x the FLOPs of a true grouped GEMM), the weight gradient is taken densely into a
        # scratch tensor and its diagonal blocks are added to the arena.
        if groups < 1 or in_channels % groups or out_channels % groups:
            raise ValueError(f"groups={groups} must divide in_channels={in_channels} and out_channels={out_channels}")
        self.groups = groups
        if groups > 1:
            self.batched_prep_shape = None       # not part of the arena's batched bf16 preparation: prepare() embeds the blocks first
        if bias and not self.modulated:
            raise AssertionError("DeformConv has no bias (detectron2)")
        if isinstance(kernel_size, (tuple, list)):
            assert kernel_size[0] == kernel_size[1]
            kernel_size = kernel_size[0]
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.deformable_groups = stride, padding, dilation, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, in_channels // groups))   # KRSC (C / groups per row)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight.permute(0, 3, 1, 2), nonlinearity="relu")
        # detectron2 wraps a norm around the op inside DeformBottleneckBlock (conv2 = DeformConv(..., norm=FrozenBN)): the frozen affine
        # is folded into the compute copy of the weights (scale) and the epilogue bias (shift), as HipConv2d does
        self.frozen_bn = frozen_bn
        if frozen_bn:
            self.register_buffer("bn_weight", torch.ones(out_channels))
            self.register_buffer("bn_bias", torch.zeros(out_channels))
            self.register_buffer("bn_running_mean", torch.zeros(out_channels))
            self.register_buffer("bn_running_var", torch.ones(out_channels) - 1e-5)
        self._prep_key = None
        self.w_bf16 = self.wt_bf16 = self.bias_eff = self.bn_scale = None

    def dense_weight(self, w):
        """(K, k, k, C / groups) -> block-diagonal (K, k, k, C)."""
        K, R, S, Cg = w.shape
        g = self.groups
        if g == 1:
            return w
        d = w.new_zeros(K, R, S, Cg * g)
        idx = torch.arange(g, device=w.device)
        d.view(g, K // g, R, S, g, Cg)[idx, :, :, :, idx, :] = w.reshape(g, K // g, R, S, Cg)
        return d

    def blocks_of(self, dense):
        """Diagonal blocks of a dense (K, k, k, C) tensor -> (K, k, k, C / groups)."""
        K, R, S, C = dense.shape
        g = self.groups
        idx = torch.arange(g, device=dense.device)
        return dense.view(g, K // g, R, S, g, C // g)[idx, :, :, :, idx, :].reshape(K, R, S, C // g)

    # batched weight preparation (layers/arena.py): the GEMM view is (K, 1, 1, k*k*C)
    def batched_prep_shape(self):
        return self.out_channels, 1, self.kernel_size * self.kernel_size * self.in_channels

    def frozen_bn_scale(self):
        return self.frozen_bn

    def bind_batched_prep(self, krsc, crsk, scale_view):
        K, n = self.out_channels, self.kernel_size * self.kernel_size * self.in_channels
        self._b_krsc, self._b_crsk, self._b_scale = krsc.view(K, 1, 1, n), crsk.view(n, 1, 1, K), scale_view
        self._prep_ver = None
        if self.frozen_bn:
            self._fold_bn(self._b_scale)

    def _bn_state(self):
        return (self.bn_weight._version, self.bn_bias._version, self.bn_running_mean._version, self.bn_running_var._version,
                self.bn_weight.data_ptr(), self.bias._version if self.bias is not None else 0)

    @torch.no_grad()
    def _fold_bn(self, scale_out=None):
        scale = self.bn_weight * torch.rsqrt(self.bn_running_var + 1e-5)
        shift = self.bn_bias - self.bn_running_mean * scale
        if scale_out is not None:
            scale_out.copy_(scale)
            scale = scale_out
        self.bn_scale = scale.contiguous()
        self.bias_eff = (shift + (self.bias.detach() * scale if self.bias is not None else 0)).contiguous()
        self._bn_key = self._bn_state()

    def prepare(self):
        arena = _arena_of(self)
        key = (self.weight._version, arena.generation if (arena is not None and self.weight.requires_grad) else -1, self.weight.data_ptr(),
               self._bn_state() if self.frozen_bn else None, HF.PRECISION)
        if key == self._prep_key:
            return
        # (the batched preparation writes the bf16 arenas; the fp32 validation mode derives its copies per layer, as HipConv2d does)
        batched = arena is not None and getattr(self, "_b_krsc", None) is not None and self.weight.requires_grad and not HF.is_f32()
        if self.frozen_bn:
            if self._bn_state() != getattr(self, "_bn_key", None):      # buffers changed (checkpoint load): re-fold, re-prepare
                self._fold_bn(self._b_scale if batched else None)
                if batched:
                    arena._prep_gen = -1
        else:
            self.bn_scale = None
            self.bias_eff = self.bias.detach() if self.bias is not None else None
        if batched:
            ver = (self.weight._version, self.weight.data_ptr())
            if arena._prep_gen != arena.generation or ver != self._prep_ver:
                arena.prep_all()
            self.w_bf16, self.wt_bf16 = self._b_krsc, self._b_crsk
            self._prep_ver, self._prep_key = ver, key
            return
        K, k, C = self.out_channels, self.kernel_size, self.in_channels
        # the GEMM sees a 1x1 convolution over k*k*C "channels" (tap-major, the layout deform_im2col writes)
        with torch.no_grad():
            dense = self.dense_weight(self.weight.detach()).contiguous()
        self.w_bf16, self.wt_bf16 = HF.weight_prep(dense.view(K, 1, 1, k * k * C), self.bn_scale)
        self._prep_key = key

    def forward(self, x, offset, mask=None, off_ld=0, mask_ld=0, mask_is_logit=False):
        self.prepare()
        if self.modulated and mask is None:
            raise ValueError("ModulatedDeformConv needs a mask")
        return _DeformConvFn.apply(x, offset, mask if self.modulated else None, self.weight, self, off_ld, mask_ld, mask_is_logit)


class ModulatedDeformConv(DeformConv):
    """detectron2.layers.ModulatedDeformConv (DCNv2): samples are multiplied by a mask; optional bias."""
    modulated = True


class DFConv2d(nn.Module):
    """slender_det/layers/df_conv.py:6-78: a regular conv predicts the offsets (and mask logits), then DeformConv /
    ModulatedDeformConv.  The offset conv's output channels are padded to a multiple of 8 (18 -> 24, 27 -> 32) so its
    gradient can feed the MFMA dgrad/wgrad kernels; the mask sigmoid (df_conv.py:76) is applied inside the gather kernel."""

    def __init__(self, in_channels, out_channels, with_modulated_dcn=True, kernel_size=3, stride=1, groups=1, padding=1, dilation=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        base = kernel_size * kernel_size
        self.offset_base_channels = base
        self.with_modulated_dcn = with_modulated_dcn
        n_off = deformable_groups * base * (3 if with_modulated_dcn else 2)
        self.n_off, self.n_off_pad = n_off, _ceil8(n_off)
        self.offset = HipConv2d(in_channels, self.n_off_pad, kernel_size, stride, padding, dilation, bias=True, out_f32=True)
        fan_in = in_channels * kernel_size * kernel_size
        bound = math.sqrt(6.0 / (2.0 * fan_in))       # kaiming_uniform_(a=1)
        with torch.no_grad():
            self.offset.weight.uniform_(-bound, bound)
            self.offset.weight[n_off:].zero_()
            self.offset.bias.zero_()
        block = ModulatedDeformConv if with_modulated_dcn else DeformConv
        self.conv = block(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias=bias)
        self.deformable_groups = deformable_groups

    def forward(self, x):
        assert x.numel() > 0, "only non-empty tensors are supported"
        om = self.offset(x)                      # (N,Ho,Wo,n_off_pad) fp32
        if not self.with_modulated_dcn:
            return self.conv(x, om, None, off_ld=self.n_off_pad)
        split = self.offset_base_channels * 2 * self.deformable_groups
        mask = om.view(-1)[split:]               # same rows, starting at the mask columns
        return self.conv(x, om, mask, off_ld=self.n_off_pad, mask_ld=self.n_off_pad, mask_is_logit=True)
