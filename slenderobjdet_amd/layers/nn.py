"""nn.Module / autograd layer over the HIP ops (NHWC bf16 activations, fp32 master weights in KRSC layout).

Mirrors the building blocks the reference gets from detectron2.layers / torch.nn for this path:
``Conv2d`` (+ FrozenBatchNorm2d, + ReLU, + residual add), ``nn.GroupNorm(32, C)`` + ReLU (fcosv2.py:315-336),
``Scale`` (slender_det/layers/scale.py:5-11, fused into the loss kernel here).  Every forward/backward is a
sequence of C-ABI calls; weight gradients are accumulated by the wgrad kernel straight into the flat gradient
arena (:mod:`slenderobjdet_amd.layers.arena`).
"""
import math

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import functional as HF


def _arena_of(mod):
    return getattr(mod, "_arena", None)


class DeferSlot:
    """Hand-over between a producer node that can fold a consumer's data gradient into its own first backward kernel and that consumer.

    A ResNet stage output feeds the next stage AND an FPN lateral 1x1 conv; autograd would sum the two input gradients with an add
    kernel and the stage would then apply its output ReLU mask in another pass (7 passes over the wide tensor incl. the lateral's own
    dgrad store).  Instead the stage publishes a slot for its output (``offer``); the lateral conv that consumes exactly that tensor
    takes it (``take``), and in backward leaves its output gradient here instead of running its dgrad; the stage's backward - which
    autograd runs after every consumer - runs that dgrad itself with ``accum`` = the other consumers' gradient and ``relu_mask`` =
    the stage output: read accum, read mask, write = 3 passes.  The NEXT stage defers as well when its first block opens with stride-2
    1x1 convolutions (STRIDE_IN_1X1): only the even (h, w) positions of its input receive gradient, so it leaves the COMPACT
    (N, H/2, W/2, C) gradient in ``comp`` instead of scattering it into a zero tensor of the full shape, and the producer's fused
    launch adds it at the even positions (``accum_even``)."""
    _offers = {}

    def __init__(self):
        self.g = self.mod = None
        self.comp = None          # compact (N, H/2, W/2, C) data gradient of the next stage's stride-2 1x1 convs (see resnet.py)

    @classmethod
    def reset(cls):
        """Start of a backbone forward: offers nobody took (outputs without a lateral conv) must not outlive their tensors."""
        cls._offers.clear()

    @classmethod
    def offer(cls, tensor):
        slot = cls()
        cls._offers[(tensor.data_ptr(), tuple(tensor.shape))] = slot
        return slot

    @classmethod
    def take(cls, tensor):
        """The slot offered for ``tensor`` (None if its producer made no offer).  Offers stay registered until the next backbone forward:
        a stage output has up to two deferring consumers, the FPN lateral conv and the next stage."""
        return cls._offers.get((tensor.data_ptr(), tuple(tensor.shape)))


import os as _os

# SOD_GN_EPILOGUE_STATS=0: the GroupNorm statistics of the tower units come from their own pass over the conv output
GN_EPILOGUE_STATS = _os.environ.get("SOD_GN_EPILOGUE_STATS", "1") != "0"
# False: the FPN lateral convs run their own data gradient and autograd sums it with the next stage's (see DeferSlot)
DEFER_LATERAL_DGRAD = True


class HipConv2d(nn.Module):
    """Conv2d with optional folded FrozenBatchNorm2d, fused bias / residual / ReLU epilogue.

    weight: (K, R, S, C) fp32 master copy ("KRSC"; ``weight_kcrs()`` gives the torch layout).
    ``cin_pad``: pad input channels of the bf16 compute copy (stem: 3 -> 8).
    ``mask_input``: the input is a post-ReLU tensor consumed only by this conv, so dgrad applies its ReLU mask
    (the producer is then constructed with ``grad_premasked=True`` and skips its own mask pass).
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True,
                 frozen_bn=False, relu=False, cin_pad=None, mask_input=False, grad_premasked=False, out_f32=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.relu, self.mask_input, self.grad_premasked = relu, mask_input, grad_premasked
        self.cin_pad = cin_pad
        self.out_f32 = out_f32      # fp32 output (offset / prediction convs); its gradient is converted to bf16 for the MFMA kernels
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.frozen_bn = frozen_bn
        if frozen_bn:   # detectron2 FrozenBatchNorm2d buffers, eps 1e-5
            self.register_buffer("bn_weight", torch.ones(out_channels))
            self.register_buffer("bn_bias", torch.zeros(out_channels))
            self.register_buffer("bn_running_mean", torch.zeros(out_channels))
            self.register_buffer("bn_running_var", torch.ones(out_channels) - 1e-5)
        self._prep_key = None
        self.w_bf16 = self.wt_bf16 = self.bias_eff = self.bn_scale = None

    # initialisers used by the model builders
    def init_msra(self):       # c2_msra_fill: kaiming_normal_(fan_out, relu)
        fan_out = self.out_channels * self.kernel_size * self.kernel_size
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / fan_out))
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def init_xavier(self):     # c2_xavier_fill: kaiming_uniform_(a=1)
        fan_in = self.in_channels * self.kernel_size * self.kernel_size
        bound = math.sqrt(6.0 / (2.0 * fan_in))
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def init_normal(self, std=0.01, bias_value=0.0):
        nn.init.normal_(self.weight, 0.0, std)
        if self.bias is not None:
            nn.init.constant_(self.bias, bias_value)

    def weight_kcrs(self):
        return self.weight.detach().permute(0, 3, 1, 2).contiguous()

    # ---- batched weight preparation (layers/arena.py: one launch for all trainable convs instead of one per conv) ----
    def batched_prep_shape(self):
        K, R, S, C = self.weight.shape
        return K, R * S, C

    def frozen_bn_scale(self):
        return self.frozen_bn

    def bind_batched_prep(self, krsc, crsk, scale_view):
        K, R, S, C = self.weight.shape
        self._b_krsc, self._b_crsk, self._b_scale = krsc.view(K, R, S, C), crsk.view(C, R, S, K), scale_view
        self._prep_ver = None
        if self.frozen_bn:          # fold now, so that the first step needs ONE batched launch, not one per FrozenBN conv
            self._fold_bn()

    def _bn_state(self):
        return (self.bn_weight._version, self.bn_bias._version, self.bn_running_mean._version, self.bn_running_var._version,
                self.bn_weight.data_ptr(), self.bias._version if self.bias is not None else 0)

    @torch.no_grad()
    def _fold_bn(self):
        scale = self.bn_weight * torch.rsqrt(self.bn_running_var + 1e-5)
        shift = self.bn_bias - self.bn_running_mean * scale
        self._b_scale.copy_(scale)
        self.bn_scale = self._b_scale
        self.bias_eff = (shift + (self.bias.detach() * scale if self.bias is not None else 0)).contiguous()
        self._bn_key = self._bn_state()

    def _prepare_batched(self, arena, key):
        if self.frozen_bn:
            if self._bn_state() != getattr(self, "_bn_key", None):     # buffers changed (checkpoint load): re-fold and re-prepare
                self._fold_bn()
                arena._prep_gen = -1
        else:
            self.bn_scale = None
            self.bias_eff = self.bias.detach() if self.bias is not None else None
        ver = (self.weight._version, self.weight.data_ptr())
        if arena._prep_gen != arena.generation or ver != self._prep_ver:
            arena.prep_all()
        self.w_bf16, self.wt_bf16 = self._b_krsc, self._b_crsk
        self._prep_ver, self._prep_key = ver, key

    def prepare(self, force=False):
        """(Re)build the bf16 compute copies when the master weights changed."""
        arena = _arena_of(self)
        key = (self.weight._version, self.bias._version if self.bias is not None else 0,
               arena.generation if (arena is not None and self.weight.requires_grad) else -1, self.weight.data_ptr(), HF.PRECISION)
        if not force and key == self._prep_key:
            return
        # (the batched preparation writes the bf16 arenas; the fp32 validation mode derives its copies per convolution)
        if arena is not None and getattr(self, "_b_krsc", None) is not None and self.weight.requires_grad and not HF.is_f32():
            return self._prepare_batched(arena, key)
        w = self.weight.detach()
        if self.frozen_bn:
            bn_key = (self.bn_weight._version, self.bn_bias._version, self.bn_running_mean._version, self.bn_running_var._version,
                      self.bn_weight.data_ptr(), self.bias._version if self.bias is not None else 0, "own")
            if bn_key != getattr(self, "_bn_key", None):   # the folded affine is constant: recompute only when buffers change
                scale = self.bn_weight * torch.rsqrt(self.bn_running_var + 1e-5)
                shift = self.bn_bias - self.bn_running_mean * scale
                self.bn_scale = scale.contiguous()
                self.bias_eff = (shift + (self.bias.detach() * scale if self.bias is not None else 0)).contiguous()
                self._bn_key = bn_key
        else:
            self.bn_scale = None
            self.bias_eff = self.bias.detach() if self.bias is not None else None
        need_t = self.weight.requires_grad or True
        self.w_bf16, self.wt_bf16 = HF.weight_prep(w.contiguous(), self.bn_scale, True, need_t, self.cin_pad)
        self._prep_key = key

    def forward(self, x, res=None, res_up2=False):
        self.prepare()
        # the master weight rides along as a differentiable input so autograd builds the node; its gradient is
        # accumulated by the wgrad kernel directly into the arena (backward returns None for it)
        return _ConvFn.apply(x, res, self.weight, self, res_up2)


# SOD_GROUPED_WINDOW=0: grouped convolutions as block-diagonal DENSE embeddings on every shape (round 3's form; groups x the FLOPs)
WINDOWED = _os.environ.get("SOD_GROUPED_WINDOW", "1") != "0"


class HipGroupedConv2d(HipConv2d):
    """Conv2d with ``groups`` > 1 (ResNeXt bottlenecks: MODEL.RESNETS.NUM_GROUPS / WIDTH_PER_GROUP, e.g.
    configs/ablation_studies/pointset/base_X101.yaml:9-11).  The master weight has the reference's shape (K, R, S, C / groups); the
    compute copies are its block-diagonal embedding into a dense (K, R, S, C) weight, so forward and data gradient run on the same
    implicit-GEMM kernels as every other convolution and are exact (the off-diagonal blocks are zeros).  The weight gradient is taken
    densely into a scratch tensor and its diagonal blocks are added to the arena.  Cost: groups x the FLOPs of a true grouped kernel on
    this one layer type - correctness first; a channel-window variant of the implicit GEMM is the follow-up."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, **kw):
        if in_channels % groups or out_channels % groups:
            raise ValueError(f"groups={groups} must divide in_channels={in_channels} and out_channels={out_channels}")
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, **kw)
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, in_channels // groups))

    batched_prep_shape = None          # not part of the arena's batched bf16 preparation: prepare() embeds the blocks first

    def init_msra(self):               # fan_out of the grouped layer as torch computes it: out_channels * k * k / groups ... of weight (K, C/g, k, k)
        fan_out = self.out_channels * self.kernel_size * self.kernel_size
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / fan_out))

    # ---- channel-window mode (round 4): when C == K are multiples of 128 and a group's channels divide 128, a 128-wide output tile only
    # needs the 128 input channels at the same offset: compute copies (K, R, S, 128) / (C, R, S, 128), block-diagonal inside each
    # window, on the implicit-GEMM kernels' window mode (SOD_CONV_CWIN / sod_conv2d_dgrad_cwin / SOD_WGRAD_DIAG).  FLOPs: 128 / (C / g)
    # times a true grouped kernel's (32x8d: 16x at res2 ... 2x at res5) instead of g = 32 times for the dense embedding.
    def windowed(self):
        C, K, g = self.in_channels, self.out_channels, self.groups
        return (WINDOWED and not HF.is_f32() and C == K and C > HF.CWIN and C % HF.CWIN == 0 and HF.CWIN % (C // g) == 0)

    def window_weight(self, w):
        """(K, R, S, C/g) -> (K, R, S, 128): row q's weights at the position of its group's channels inside q's 128-channel window."""
        K, R, S, Cg = w.shape
        T, gt = K // HF.CWIN, HF.CWIN // Cg          # tiles, groups per tile
        d = w.new_zeros(T, gt, Cg, R, S, gt, Cg)
        idx = torch.arange(gt, device=w.device)
        d[:, idx, :, :, :, idx, :] = w.reshape(T, gt, Cg, R, S, Cg).permute(1, 0, 2, 3, 4, 5)
        return d.reshape(K, R, S, HF.CWIN)

    def window_blocks(self, win):
        """(K, R, S, 128) -> (K, R, S, C/g): the inverse selection (the rest of a window are structural zeros / unused gradients)."""
        K, R, S, _ = win.shape
        Cg = self.in_channels // self.groups
        T, gt = K // HF.CWIN, HF.CWIN // Cg
        idx = torch.arange(gt, device=win.device)
        v = win.reshape(T, gt, Cg, R, S, gt, Cg)
        return v[:, idx, :, :, :, idx, :].permute(1, 0, 2, 3, 4, 5).reshape(K, R, S, Cg)

    def dense_weight(self, w):
        """(K, R, S, C/g) -> block-diagonal (K, R, S, C)."""
        K, R, S, Cg = w.shape
        g = self.groups
        d = w.new_zeros(K, R, S, Cg * g)
        idx = torch.arange(g, device=w.device)
        d.view(g, K // g, R, S, g, Cg)[idx, :, :, :, idx, :] = w.reshape(g, K // g, R, S, Cg)
        return d

    def blocks_of(self, dense):
        """Diagonal blocks of a dense (K, R, S, C) tensor -> (K, R, S, C/g)."""
        K, R, S, C = dense.shape
        g = self.groups
        idx = torch.arange(g, device=dense.device)
        return dense.view(g, K // g, R, S, g, C // g)[idx, :, :, :, idx, :].reshape(K, R, S, C // g)

    def weight_kcrs(self):
        return self.weight.detach().permute(0, 3, 1, 2).contiguous()

    def prepare(self, force=False):
        arena = _arena_of(self)
        key = (self.weight._version, arena.generation if (arena is not None and self.weight.requires_grad) else -1, self.weight.data_ptr(),
               self._bn_state() if self.frozen_bn else None, HF.PRECISION)
        if not force and key == self._prep_key:
            return
        if self.frozen_bn:
            scale = self.bn_weight * torch.rsqrt(self.bn_running_var + 1e-5)
            self.bn_scale = scale.contiguous()
            self.bias_eff = (self.bn_bias - self.bn_running_mean * scale + (self.bias.detach() * scale if self.bias is not None else 0)).contiguous()
        else:
            self.bn_scale = None
            self.bias_eff = self.bias.detach() if self.bias is not None else None
        with torch.no_grad():
            if self.windowed():
                win = self.window_weight(self.weight.detach())
                if self.bn_scale is not None:
                    win = win * self.bn_scale.view(-1, 1, 1, 1)
                T = self.out_channels // HF.CWIN
                # the data gradient's copy: row c holds, per tap, the weights towards the 128 outputs of c's own tile
                win_t = win.reshape(T, HF.CWIN, self.kernel_size, self.kernel_size, HF.CWIN).permute(0, 4, 2, 3, 1).reshape(win.shape)
                self.w_bf16 = HF.weight_prep(win.contiguous(), None, True, False, None)[0]
                self.wt_bf16 = HF.weight_prep(win_t.contiguous(), None, True, False, None)[0]
                self._prep_key = key
                return
            dense = self.dense_weight(self.weight.detach())
        self.w_bf16, self.wt_bf16 = HF.weight_prep(dense.contiguous(), self.bn_scale, True, True, None)
        self._prep_key = key

    def wgrad_into(self, arena, g, x):
        """dW of the dense embedding into a scratch tensor on the CURRENT stream, diagonal blocks added to the arena gradient."""
        # as ONE unit behind whatever an open wgrad_batch has collected (the scratch tensor is consumed right after its launch)
        HF.batch_or_call(lambda: self._wgrad_into_now(arena, g, x))

    def _wgrad_into_now(self, arena, g, x):
        K, k, C = self.out_channels, self.kernel_size, self.in_channels
        win = self.windowed()
        dense = torch.zeros((K, k, k, HF.CWIN if win else C), dtype=torch.float32, device=g.device)
        prev, HF.WGRAD_SIDE_STREAM = HF.WGRAD_SIDE_STREAM, False
        try:
            HF.conv2d_wgrad(g, x, dense, k, k, self.stride, self.padding, self.dilation, qscale=self.bn_scale)
        finally:
            HF.WGRAD_SIDE_STREAM = prev
        arena.grad_view(self.weight).add_(self.window_blocks(dense) if win else self.blocks_of(dense))
        arena.mark_ready(self.weight)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, weight, mod, res_up2):
        y = HF.conv2d_fwd(x, mod.w_bf16, mod.bias_eff, res, mod.stride, mod.padding, mod.dilation, relu=mod.relu, res_up2=res_up2,
                          out_f32=mod.out_f32, c_real=mod.in_channels if mod.cin_pad else None)
        ctx.mod, ctx.res_up2, ctx.has_res = mod, res_up2, res is not None
        # a 1x1 stride-1 conv on a stage output whose producer offered to run this conv's data gradient (DeferSlot)
        ctx.slot = None
        if DEFER_LATERAL_DGRAD and x.requires_grad and mod.kernel_size == 1 and mod.stride == 1 and not mod.mask_input and not mod.relu:
            ctx.slot = DeferSlot.take(x)
        train_w = mod.weight.requires_grad
        if train_w or x.requires_grad or (res is not None and res.requires_grad):
            ctx.save_for_backward(x, y if mod.relu else None)
            if train_w and _arena_of(mod) is not None:
                _arena_of(mod).note_use(mod.weight)
                if mod.bias is not None:
                    _arena_of(mod).note_use(mod.bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        mod = ctx.mod
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype == torch.float32:
            dy = HF.f32_to_bf16(dy)
        g = HF.relu_bwd(dy, y) if (mod.relu and not mod.grad_premasked) else dy
        arena = _arena_of(mod)
        N, H, W, C = x.shape
        HF.release_held()               # weight gradients the head towers parked (SOD_HOLD_HEAD_WGRAD=1; nothing otherwise)
        if mod.weight.requires_grad:
            with HF.wgrad_batch():      # weight + bias gradient: one hand-over to the side stream
                if getattr(mod, "groups", 1) > 1:
                    mod.wgrad_into(arena, g, x)
                else:
                    dw = arena.grad_view(mod.weight)
                    HF.conv2d_wgrad(g, x, dw, mod.kernel_size, mod.kernel_size, mod.stride, mod.padding, mod.dilation, qscale=mod.bn_scale)
                    arena.mark_ready(mod.weight)
                if mod.bias is not None:
                    HF.bias_grad(g, arena.grad_view(mod.bias), N, g.shape[1] * g.shape[2], mod.out_channels)
                    arena.mark_ready(mod.bias)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.slot is not None:      # the producer of x runs this data gradient fused with its own mask / accumulate (DeferSlot)
                ctx.slot.g, ctx.slot.mod = g, mod
            else:
                dx = HF.conv2d_dgrad(g, mod.wt_bf16, (H, W), mod.stride, mod.padding, mod.dilation,
                                     relu_mask=x if mod.mask_input else None)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[1]:
            dres = HF.upsample2x_bwd(g) if ctx.res_up2 else g
        return dx, dres, None, None, None


class HipGroupNorm(nn.Module):
    """Parameters of nn.GroupNorm(num_groups, C); applied fused with the preceding conv by ConvGnRelu."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))


class GradPark:
    """Two-stage detectors: the ROI pooler and the RPN head read the same FPN outputs, and the pooler's backward runs first (its node is
    the younger one).  Instead of handing autograd a dense gradient per level that it then ADDS to the RPN head's (read 2, write 1 over
    the 256-channel P2 ... P5 tensors), the pooler parks its gradients here and the RPN head's last data-gradient launch adds them in its
    epilogue (sod_conv2d_dgrad_ml_accum).  ``current`` is set by GeneralizedRCNN.forward for the duration of one training forward pass."""
    current = None

    def __init__(self):
        self.consumer_ptrs = set()      # data_ptr of every tensor the RPN head node will produce a gradient for (set in ITS forward)
        self.parked = {}
        self.done = False               # the RPN head's backward has run: nobody collects any more
        self.hooked = False

    def put(self, ptr, g):
        self.parked[ptr] = g
        if not self.hooked:
            torch.autograd.Variable._execution_engine.queue_callback(self._check)
            self.hooked = True

    def _check(self):
        self.hooked = False
        if self.parked:
            self.parked.clear()
            raise RuntimeError("GradPark: the ROI pooler parked feature gradients and the RPN head's backward never collected them "
                               "(meta_arch.rcnn.GRAD_PARK = False restores autograd's accumulation)")


class SiblingFold:
    """Shared by the FIRST units of sibling chains that read the same tensors (the two towers of FCOSHead on the FPN outputs,
    fcosv2.py:342-361).  autograd would add the two input gradients level by level (read 2, write 1: five launches on the critical path
    of backward); instead the unit whose backward runs first parks its data gradients here and returns none, and the second adds them
    in the epilogue of its own data-gradient launch (sod_conv2d_dgrad_ml_accum) and returns the sum.  Only for graphs in which BOTH
    siblings receive a gradient (the fused FCOS loss node): a parked gradient that nobody collected by the end of the backward pass
    raises instead of being lost."""

    def __init__(self):
        self.partial = self.event = self.stream = None
        self.hooked = False

    def put(self, dxs):
        self.partial = list(dxs)
        self.stream = torch.cuda.current_stream(dxs[0].device)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        if not self.hooked:
            torch.autograd.Variable._execution_engine.queue_callback(self._check)
            self.hooked = True

    def take(self):
        part, self.partial = self.partial, None
        cur = torch.cuda.current_stream(part[0].device)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_event(self.event)
            for t in part:
                t.record_stream(cur)
        return part

    def _check(self):
        self.hooked = False
        if self.partial is not None:
            self.partial = None
            raise RuntimeError("SiblingFold: one tower parked its input gradient and the other never ran in this backward pass "
                               "(SOD_TOWER_FOLD=0 restores autograd's accumulation)")


class ConvGnRelu(nn.Module):
    """[Conv3x3(bias) -> GroupNorm(32) -> ReLU] unit of the FCOS towers (fcosv2.py:300-336), applied to ALL FPN levels at
    once: the levels share the weights, so the convolution forward / dgrad / wgrad are one multi-level launch each."""

    def __init__(self, channels, num_groups=32):
        super().__init__()
        self.conv = HipConv2d(channels, channels, 3, 1, 1, bias=True)
        self.gn = HipGroupNorm(num_groups, channels)

    def forward(self, xs, fold=None):
        """``fold``: a SiblingFold shared with the other unit that reads the same ``xs``."""
        single = isinstance(xs, torch.Tensor)
        if single:
            xs = [xs]
        self.conv.prepare()
        out = _ConvGnReluFn.apply(self.conv.weight, self, fold, *xs)
        return out[0] if single else list(out)


class _ConvGnReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, unit, fold, *xs):
        conv, gn = unit.conv, unit.gn
        ctx.fold = fold
        gw, gb = gn.weight.detach(), gn.bias.detach()
        relu = True
        if (GN_EPILOGUE_STATS and not HF.DETERMINISTIC and not HF.is_f32() and conv.out_channels == 8 * gn.num_groups
                and conv.stride == 1 and conv.dilation == 1 and 2 * conv.padding == conv.kernel_size - 1):
            # the norm's statistics are gathered in the conv epilogue (float atomics: not for the deterministic mode)
            y1s, y2s, stats = HF.conv_gn_fwd_ml(list(xs), conv.w_bf16, conv.bias_eff, gw, gb, gn.num_groups, gn.eps, relu=relu, pad=conv.padding)
        else:
            y1s = HF.conv2d_fwd_ml(list(xs), conv.w_bf16, conv.bias_eff, conv.stride, conv.padding, conv.dilation)
            y2s, stats = HF.groupnorm_fwd_ml(y1s, gw, gb, gn.num_groups, gn.eps, relu=relu)      # all levels in one launch per pass
        ctx.unit, ctx.nl, ctx.relu = unit, len(xs), relu
        ctx.save_for_backward(*xs, *y1s, stats)
        arena = _arena_of(conv)
        if arena is not None and conv.weight.requires_grad:
            for p in (conv.weight, conv.bias, gn.weight, gn.bias):
                if p is not None:
                    arena.note_use(p)
        return tuple(y2s)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dy2s):
        conv, gn = ctx.unit.conv, ctx.unit.gn
        nl = ctx.nl
        saved = ctx.saved_tensors
        xs, y1s, stats = saved[:nl], saved[nl:2 * nl], saved[2 * nl]
        arena = _arena_of(conv)
        gw, gb = gn.weight.detach(), gn.bias.detach()
        dgw, dgb = arena.grad_view(gn.weight), arena.grad_view(gn.bias)
        relu, k = ctx.relu, conv.kernel_size
        # the conv bias gradient (sum of dy1 over pixels) falls out of the GN backward pass
        dbias = arena.grad_view(conv.bias) if conv.bias is not None else None
        dy1s = HF.groupnorm_bwd_ml([d.contiguous() for d in dy2s], list(y1s), gw, gb, stats, gn.num_groups, dgw, dgb, relu=relu, dxsum=dbias)
        arena.mark_ready(gn.weight)
        arena.mark_ready(gn.bias)
        if conv.bias is not None:
            arena.mark_ready(conv.bias)
        def wgrad():
            HF.conv2d_wgrad_ml(dy1s, list(xs), arena.grad_view(conv.weight), k, k, conv.stride, conv.padding, conv.dilation)
            arena.mark_ready(conv.weight)
        HF.hold_or_call(wgrad)          # (SOD_HOLD_HEAD_WGRAD=1: parked until the FPN backward starts; default: launched here)
        dxs = [None] * nl
        if any(ctx.needs_input_grad[3:]):
            fold, ctx.fold = ctx.fold, None
            hw = [(x.shape[1], x.shape[2]) for x in xs]
            if fold is not None and all(ctx.needs_input_grad[3:]):
                if fold.partial is None:      # first of the two siblings: park the gradient, the other returns the sum
                    fold.put(HF.conv2d_dgrad_ml(dy1s, conv.wt_bf16, hw, conv.stride, conv.padding, conv.dilation))
                else:
                    dxs = HF.conv2d_dgrad_ml(dy1s, conv.wt_bf16, hw, conv.stride, conv.padding, conv.dilation, accums=fold.take())
            else:
                dxs = HF.conv2d_dgrad_ml(dy1s, conv.wt_bf16, hw, conv.stride, conv.padding, conv.dilation)
        return (None, None, None, *dxs)


class _ReluToken:
    """Shared by a ConvReluML unit and the unit that consumes its outputs as their ONLY consumer: the consumer's data gradient applies
    this unit's ReLU mask in its epilogue (sod_conv2d_dgrad_ml_mask), so this unit's backward skips its relu_bwd launches."""

    def __init__(self, outs):
        self.out_ptrs = [(t.data_ptr(), tuple(t.shape)) for t in outs]
        self.premasked = False

    def matches(self, xs):
        return len(xs) == len(self.out_ptrs) and all((x.data_ptr(), tuple(x.shape)) == k for x, k in zip(xs, self.out_ptrs))


# False: every ConvReluML unit masks its own incoming gradient (one relu_bwd launch per level)
RELU_CHAIN = True


class ConvReluML(nn.Module):
    """[Conv3x3(bias) -> ReLU] tower unit of RetinaNetHead (retina_rotated.py:418-430) over all FPN levels in one launch."""

    def __init__(self, channels):
        super().__init__()
        self.conv = HipConv2d(channels, channels, 3, 1, 1, bias=True)
        self._last_token = None

    def forward(self, xs, chained=None):
        """``chained``: the ConvReluML unit whose latest forward produced ``xs`` and that has no other consumer (the previous unit of
        a tower): this unit's data gradient then applies that unit's ReLU mask in its epilogue."""
        self.conv.prepare()
        tok = None
        if RELU_CHAIN and chained is not None and chained._last_token is not None and chained._last_token.matches(xs):
            tok = chained._last_token
        return list(_ConvReluMLFn.apply(self.conv.weight, self, tok, *xs))


class _ConvReluMLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, unit, in_token, *xs):
        conv = unit.conv
        ys = HF.conv2d_fwd_ml(list(xs), conv.w_bf16, conv.bias_eff, 1, 1, 1, relu=True)
        ctx.unit, ctx.nl = unit, len(xs)
        ctx.save_for_backward(*xs, *ys)
        ctx.in_token = in_token
        ctx.token = unit._last_token = None
        arena = _arena_of(conv)
        if arena is not None and conv.weight.requires_grad:
            arena.note_use(conv.weight)
            arena.note_use(conv.bias)
            ctx.token = unit._last_token = _ReluToken(ys)
        if in_token is not None and all(x.requires_grad for x in xs):
            in_token.premasked = True          # decided in forward: the producer's backward runs after ours and relies on it
        else:
            ctx.in_token = None
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        conv, nl = ctx.unit.conv, ctx.nl
        saved = ctx.saved_tensors
        xs, ys = saved[:nl], saved[nl:]
        arena = _arena_of(conv)
        if ctx.token is not None and ctx.token.premasked:      # the consumer's data gradient already applied this unit's ReLU mask
            gs = [dy.contiguous() for dy in dys]
        else:
            gs = [HF.relu_bwd(dy.contiguous(), y) for dy, y in zip(dys, ys)]
        def wgrad():
            with HF.wgrad_batch():
                HF.conv2d_wgrad_ml(gs, list(xs), arena.grad_view(conv.weight), 3, 3, 1, 1, 1)
                arena.mark_ready(conv.weight)
                HF.bias_grad_ml(gs, arena.grad_view(conv.bias))
                arena.mark_ready(conv.bias)
        HF.hold_or_call(wgrad)          # parked until the FPN backward starts (layers/functional.py: HOLD_HEAD_WGRAD)
        dxs = [None] * nl
        if any(ctx.needs_input_grad[3:]):
            hw = [(x.shape[1], x.shape[2]) for x in xs]
            dxs = HF.conv2d_dgrad_ml(gs, conv.wt_bf16, hw, 1, 1, 1, relu_masks=list(xs) if ctx.in_token is not None else None)
        return (None, None, None, *dxs)


class ConvML(nn.Module):
    """A plain Conv2d(bias) shared by all FPN levels, one multi-level launch per pass (RepPoints' 1x1 / 3x3 output convs,
    rpd.py:156-167).  ``relu``: fused ReLU epilogue; ``out_f32``: fp32 output rows (offsets), whose gradient is converted to
    bf16 for the MFMA kernels; ``out_channels`` is padded by the caller to a multiple of 8."""

    def __init__(self, in_channels, out_channels, kernel_size=1, padding=0, relu=False, out_f32=False):
        super().__init__()
        self.conv = HipConv2d(in_channels, out_channels, kernel_size, 1, padding, bias=True)
        self.relu, self.out_f32 = relu, out_f32

    def forward(self, xs):
        self.conv.prepare()
        return list(_ConvMLFn.apply(self.conv.weight, self, *xs))


class _ConvMLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, unit, *xs):
        conv = unit.conv
        ys = HF.conv2d_fwd_ml(list(xs), conv.w_bf16, conv.bias_eff, 1, conv.padding, 1, relu=unit.relu, out_f32=unit.out_f32)
        ctx.unit, ctx.nl = unit, len(xs)
        ctx.save_for_backward(*xs, *(ys if unit.relu else ()))
        arena = _arena_of(conv)
        if arena is not None and conv.weight.requires_grad:
            arena.note_use(conv.weight)
            arena.note_use(conv.bias)
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        unit, nl = ctx.unit, ctx.nl
        conv = unit.conv
        saved = ctx.saved_tensors
        xs, ys = saved[:nl], saved[nl:]
        arena = _arena_of(conv)
        gs = []
        for i, dy in enumerate(dys):
            dy = dy.contiguous()
            if dy.dtype == torch.float32:
                dy = HF.f32_to_bf16(dy)
            gs.append(HF.relu_bwd(dy, ys[i]) if unit.relu else dy)
        k = conv.kernel_size
        with HF.wgrad_batch():
            HF.conv2d_wgrad_ml(gs, list(xs), arena.grad_view(conv.weight), k, k, 1, conv.padding, 1)
            arena.mark_ready(conv.weight)
            HF.bias_grad_ml(gs, arena.grad_view(conv.bias))
            arena.mark_ready(conv.bias)
        dxs = [None] * nl
        if any(ctx.needs_input_grad[2:]):
            dxs = HF.conv2d_dgrad_ml(gs, conv.wt_bf16, [(x.shape[1], x.shape[2]) for x in xs], 1, conv.padding, 1)
        return (None, None, *dxs)


class _AddUp2Fn(torch.autograd.Function):
    """a + nearest-2x-upsample(b) (d2 FPN top-down sum) as its own node for FPN.NORM != ""."""

    @staticmethod
    def forward(ctx, a, b):
        return HF.add_up2(a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (dy if ctx.needs_input_grad[0] else None), (HF.upsample2x_bwd(dy) if ctx.needs_input_grad[1] else None)


def add_up2(a, b):
    return _AddUp2Fn.apply(a, b)


class _GnReluFn(torch.autograd.Function):
    """Stand-alone GroupNorm(+ReLU) (after a deformable tower conv, fcosv2.py:300-336 with USE_DCN_IN_TOWER)."""

    @staticmethod
    def forward(ctx, x, weight, gn, relu):
        y, stats = HF.groupnorm_fwd(x, gn.weight.detach(), gn.bias.detach(), gn.num_groups, gn.eps, relu=relu)
        ctx.gn, ctx.relu = gn, relu
        ctx.save_for_backward(x, stats)
        arena = _arena_of(gn)
        if arena is not None:
            arena.note_use(gn.weight)
            arena.note_use(gn.bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        gn = ctx.gn
        x, stats = ctx.saved_tensors
        arena = _arena_of(gn)
        dx = HF.groupnorm_bwd(dy.contiguous(), x, gn.weight.detach(), gn.bias.detach(), stats, gn.num_groups,
                              arena.grad_view(gn.weight), arena.grad_view(gn.bias), relu=ctx.relu)
        arena.mark_ready(gn.weight)
        arena.mark_ready(gn.bias)
        return dx, None, None, None


def group_norm_relu(x, gn, relu=True):
    return _GnReluFn.apply(x, gn.weight, gn, relu)


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = HF.relu_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return HF.relu_bwd(dy.contiguous(), y)


def relu(x):
    return _ReluFn.apply(x)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if x.requires_grad:
            raise NotImplementedError("max-pool backward is not built: the stem is frozen (MODEL.BACKBONE.FREEZE_AT >= 1)")
        return HF.maxpool3x3s2(x)


def max_pool_3x3_s2(x):
    return _MaxPoolFn.apply(x)


class Scale(nn.Module):
    """slender_det/layers/scale.py:5-11. Kept for API parity; FCOSHead fuses the multiply into the loss kernel."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, input):
        return input * self.scale


def attach_arena(model, arena):
    for m in model.modules():
        m._arena = arena
    # uses are counted only in forwards that will be followed by a backward pass (ParamArena.on_forward)
    model.register_forward_pre_hook(lambda mod, inputs: arena.on_forward(torch.is_grad_enabled() and mod.training))
