"""ResNet (18/34/50/101/152) bottom-up backbone on the HIP conv kernels, NHWC bf16.

Restates detectron2's ``build_resnet_backbone`` semantics (source absent) that
slender_det/modeling/backbone/fpn.py:103 calls: BasicStem (7x7 s2 + FrozenBN + ReLU + max-pool 3x3 s2),
Bottleneck / Basic blocks with the stride on the first 1x1 (``STRIDE_IN_1X1``), FrozenBatchNorm2d folded into the
convolution weights, ``FREEZE_AT`` stages without gradients, MSRA initialisation.
"""
import torch
from torch import nn

from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers.nn import DeferSlot, HipConv2d, HipGroupedConv2d, _arena_of, max_pool_3x3_s2
import os

from ..shape_spec import ShapeSpec

STEM_FUSED = os.environ.get("SOD_STEM_FUSED", "1") != "0"
# frozen 64 -> 256 bottleneck blocks (res2 under FREEZE_AT >= 2) as one kernel each (csrc/bottleneck_fused.hip)
BNECK_FUSED = os.environ.get("SOD_BNECK_FUSED", "1") != "0"
# ReLU masks of the block outputs of a trainable bottleneck stage as 1 bit per element, written by the conv3 epilogue and read by the
# data gradient that folds the mask in (1/16 of the bytes of re-reading the bf16 block output); False re-reads the tensor
RELU_BITS = True
# A stage whose first block opens with stride-2 1x1 convolutions hands its input gradient to the producing stage in compact form
# (layers/nn.py DeferSlot.comp) instead of a zero-stuffed full-resolution tensor; False scatters it as before
COMPACT_S2_GRAD = True
from .build import BACKBONE_REGISTRY, Backbone


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = HipConv2d(in_channels, out_channels, 7, 2, 3, bias=False, frozen_bn=True, relu=True, cin_pad=8)
        self.conv1.init_msra()
        self.out_channels, self.stride = out_channels, 4

    def forward(self, x):
        if isinstance(x, HF.RawImageBatch):
            # frozen stem on raw uint8 pixels: normalise + conv + FrozenBN + ReLU + max-pool in one kernel (csrc/stem_fused.hip)
            c = self.conv1
            if STEM_FUSED and not HF.is_f32() and not c.weight.requires_grad and c.in_channels == 3 and c.out_channels == 64:
                c.prepare()
                key = (c._prep_key, c.w_bf16.data_ptr())
                if getattr(self, "_packed_key", None) != key:
                    self._packed, self._packed_key = HF.stem_pack_weights(c.w_bf16), key
                return HF.stem_fused(x, self._packed, c.bias_eff)
            x = x.materialize()
        return max_pool_3x3_s2(self.conv1(x))


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bottleneck_channels, stride=1, stride_in_1x1=True, dilation=1, num_groups=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = HipConv2d(in_channels, out_channels, 1, stride, 0, bias=False, frozen_bn=True)
        # conv1's / conv2's ReLU outputs have a single consumer, so the consumer's dgrad applies their masks
        self.conv1 = HipConv2d(in_channels, bottleneck_channels, 1, s1, 0, bias=False, frozen_bn=True, relu=True, grad_premasked=True)
        if num_groups > 1:     # ResNeXt: the 3x3 is grouped (detectron2 BottleneckBlock(num_groups=...))
            self.conv2 = HipGroupedConv2d(bottleneck_channels, bottleneck_channels, 3, s3, dilation, dilation, groups=num_groups, bias=False,
                                          frozen_bn=True, relu=True, mask_input=True, grad_premasked=True)
        else:
            self.conv2 = HipConv2d(bottleneck_channels, bottleneck_channels, 3, s3, dilation, dilation, bias=False, frozen_bn=True,
                                   relu=True, mask_input=True, grad_premasked=True)
        self.conv3 = HipConv2d(bottleneck_channels, out_channels, 1, 1, 0, bias=False, frozen_bn=True, relu=True, mask_input=True)
        for m in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if m is not None:
                m.init_msra()

    def forward(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        out = self.conv2(self.conv1(x))
        return self.conv3(out, res=sc)     # relu(conv3 + shortcut) fused in the conv epilogue


class DeformBottleneckBlock(nn.Module):
    """detectron2's DeformBottleneckBlock (MODEL.RESNETS.DEFORM_ON_PER_STAGE; source absent): the bottleneck with its 3x3
    replaced by ``conv2_offset`` (a plain 3x3 conv with bias, zero-initialised, 18 * G offsets - or 27 * G with DEFORM_MODULATED: the
    first 18 * G channels are the offsets, the last 9 * G the mask logits) followed by DeformConv / ModulatedDeformConv with FrozenBN +
    ReLU.  The reference reaches it through configs/fcos/fcos_R_50_FPN_2x_dcnv2.yaml:6-7.  conv1's ReLU output has two consumers here
    (the offset conv and the sampled conv), so it masks its own gradient and autograd sums the two input gradients."""

    def __init__(self, in_channels, out_channels, bottleneck_channels, stride=1, stride_in_1x1=True, dilation=1, deform_modulated=False,
                 deform_num_groups=1, num_groups=1):
        super().__init__()
        from ...layers.deform_conv import DeformConv, ModulatedDeformConv, _ceil8

        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.deform_modulated, self.deform_num_groups = deform_modulated, deform_num_groups
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = HipConv2d(in_channels, out_channels, 1, stride, 0, bias=False, frozen_bn=True)
        self.conv1 = HipConv2d(in_channels, bottleneck_channels, 1, s1, 0, bias=False, frozen_bn=True, relu=True)
        self.n_off = (27 if deform_modulated else 18) * deform_num_groups
        self.n_off_pad = _ceil8(self.n_off)        # padded so that the offset gradient can feed the MFMA kernels; pad rows stay zero
        self.conv2_offset = HipConv2d(bottleneck_channels, self.n_off_pad, 3, s3, dilation, dilation, bias=True, out_f32=True)
        op = ModulatedDeformConv if deform_modulated else DeformConv
        self.conv2 = op(bottleneck_channels, bottleneck_channels, 3, s3, dilation, dilation, groups=num_groups, deformable_groups=deform_num_groups,
                        bias=False, relu=True, frozen_bn=True)
        self.conv3 = HipConv2d(bottleneck_channels, out_channels, 1, 1, 0, bias=False, frozen_bn=True, relu=True)
        for m in (self.conv1, self.conv3, self.shortcut):
            if m is not None:
                m.init_msra()
        with torch.no_grad():
            fan_out = bottleneck_channels * 9
            self.conv2.weight.normal_(0.0, (2.0 / fan_out) ** 0.5)       # c2_msra_fill
            self.conv2_offset.weight.zero_()
            self.conv2_offset.bias.zero_()
        self.conv2_offset.ckpt_rows = self.n_off

    def forward(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        out = self.conv1(x)
        om = self.conv2_offset(out)                       # (N, Ho, Wo, n_off_pad) fp32 rows: offsets, then mask logits
        if self.deform_modulated:
            mask = om.view(-1)[18 * self.deform_num_groups:]
            out = self.conv2(out, om, mask, off_ld=self.n_off_pad, mask_ld=self.n_off_pad, mask_is_logit=True)
        else:
            out = self.conv2(out, om, None, off_ld=self.n_off_pad)
        return self.conv3(out, res=sc)


class _BottleneckStageFn(torch.autograd.Function):
    """A whole stage of bottleneck blocks as ONE autograd node.  Forward is the plain sequence of fused conv launches; the
    hand-written backward chains the blocks so that the ReLU mask of a block output and the residual-gradient sum are applied
    in the dgrad epilogue of the NEXT block's first conv (accumulate + mask), instead of a relu_bwd pass plus an autograd add
    (6 tensor passes -> 2 per block)."""

    @staticmethod
    def forward(ctx, x, weight, stage):
        blocks = list(stage)
        saved, bits = [x], []
        train = blocks[0].conv1.weight.requires_grad
        need_bwd = train or x.requires_grad
        for blk in blocks:
            for m in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut):
                if m is not None:
                    m.prepare()
        for k, blk in enumerate(blocks):
            xin = saved[-1]
            sc = _conv(blk.shortcut, xin, reverse=True) if blk.shortcut is not None else xin
            a = _conv(blk.conv1, xin, reverse=blk.shortcut is None)
            b = _conv(blk.conv2, a)
            c3 = blk.conv3
            if need_bwd and RELU_BITS and not HF.is_f32() and c3.relu and c3.out_channels % 8 == 0:
                N, Hb, Wb, _ = b.shape
                Ho, Wo = HF.conv_out_size(Hb, Wb, c3.kernel_size, c3.kernel_size, c3.stride, c3.padding, c3.dilation)
                bt = torch.empty(N * Ho * Wo * c3.out_channels // 8, dtype=torch.uint8, device=b.device)
                out = HF.conv2d_fwd(b, c3.w_bf16, c3.bias_eff, sc, c3.stride, c3.padding, c3.dilation, relu=True, relu_bits=bt)
                bits.append(bt)
            else:
                out = _conv(c3, b, res=sc)
            saved += [a, b, out]
        ctx.stage = stage
        ctx.slot = None
        # the producer of x may take this stage's input gradient in compact form (DeferSlot.comp): first block = stride-2 1x1 convs
        b0 = blocks[0]
        ctx.in_slot = None
        if (x.requires_grad and COMPACT_S2_GRAD and not HF.is_f32() and b0.shortcut is not None and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
                and all(m.kernel_size == 1 and m.stride == 2 and m.padding == 0 for m in (b0.shortcut, b0.conv1))):
            ctx.in_slot = DeferSlot.take(x)
        ctx.nbits = len(bits) if len(bits) == len(blocks) else 0
        if need_bwd:
            ctx.save_for_backward(*saved, *(bits if ctx.nbits else []))
            ctx.set_materialize_grads(False)
            ctx.slot = DeferSlot.offer(saved[-1])      # an FPN lateral conv on this output may leave its data gradient to us
            arena = _arena_of(blocks[0].conv1)
            if train and arena is not None:
                for blk in blocks:
                    for m in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut):
                        if m is not None:
                            arena.note_use(m.weight)
        return saved[-1]

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        blocks = list(ctx.stage)
        saved = ctx.saved_tensors
        bits = None
        if ctx.nbits:
            saved, bits = saved[:-ctx.nbits], saved[-ctx.nbits:]

        def mask_of(k):      # ReLU mask of block k's output: its bit array if the forward pass recorded one, else the tensor itself
            return dict(relu_bits=bits[k]) if bits is not None else dict(relu_mask=saved[3 * k + 3])

        arena = _arena_of(blocks[0].conv1)
        slot, ctx.slot = ctx.slot, None
        comp = None
        if slot is not None and slot.comp is not None:      # the next stage left its input gradient in compact (even-pixel) form
            comp, slot.comp = slot.comp, None
            if not (slot.g is not None and dout is None and bits is not None):
                dense = _scatter_even(comp, saved[-1].shape)
                dout, comp = dense if dout is None else HF.add_bf16(dout.contiguous(), dense), None
        if slot is not None and slot.g is not None:
            # d(out) = dgrad(lateral 1x1, its output gradient) + the other consumers' gradient, times the ReLU mask of out: one launch
            lat = slot.mod
            if comp is not None:
                g = HF.conv2d_dgrad(slot.g, lat.wt_bf16, (saved[-1].shape[1], saved[-1].shape[2]), 1, 0, 1, accum=comp, accum_even=True,
                                    relu_bits=bits[len(blocks) - 1])
            else:
                g = HF.conv2d_dgrad(slot.g, lat.wt_bf16, (saved[-1].shape[1], saved[-1].shape[2]), 1, 0, 1,
                                    accum=None if dout is None else dout.contiguous(), **mask_of(len(blocks) - 1))
            slot.g = slot.mod = None
        elif dout is None:
            return None, None, None
        else:
            g = HF.relu_bwd(dout.contiguous(), saved[-1])
        dx = None
        for k in range(len(blocks) - 1, -1, -1):
            blk = blocks[k]
            xin, a, b = saved[3 * k], saved[3 * k + 1], saved[3 * k + 2]
            xin = saved[0] if k == 0 else saved[3 * k]        # block input = previous block's output
            _wgrad(blk.conv3, g, b, arena)
            db = _dgrad(blk.conv3, g, b, relu_mask=b, reverse=True)
            _wgrad(blk.conv2, db, a, arena)
            da = _dgrad(blk.conv2, db, a, relu_mask=a)
            with HF.wgrad_batch():       # adjacent launches: one hand-over to the side stream for both
                _wgrad(blk.conv1, da, xin, arena)
                if blk.shortcut is not None:
                    _wgrad(blk.shortcut, g, xin, arena)
            if k == 0:
                if ctx.needs_input_grad[0]:
                    if ctx.in_slot is not None:      # compact gradient for the producer's fused launch; autograd sees no gradient here
                        Ho, Wo = g.shape[1], g.shape[2]
                        part = HF.conv2d_dgrad(da, blk.conv1.wt_bf16, (Ho, Wo), 1, 0, 1)
                        ctx.in_slot.comp = HF.conv2d_dgrad(g, blk.shortcut.wt_bf16, (Ho, Wo), 1, 0, 1, accum=part)
                        ctx.in_slot = None
                    elif blk.shortcut is not None:
                        dx = _dgrad_pair_into_input(blk, g, da, xin)
                    else:
                        dx = _dgrad(blk.conv1, da, xin, accum=g)
            else:   # previous block's output is post-ReLU: fold its mask and the identity-path gradient into the epilogue
                if blk.shortcut is not None:
                    g = _dgrad(blk.shortcut, g, xin, accum=_dgrad(blk.conv1, da, xin), **mask_of(k - 1))
                else:
                    g = _dgrad(blk.conv1, da, xin, accum=g, **mask_of(k - 1))
        return dx, None, None


# The convolutions that read the wide block tensor right after it was written (forward: shortcut / conv1; backward: conv3's data gradient)
# walk their tiles last to first, i.e. start on the part of the tensor the 256 MB Infinity Cache still holds (sod_conv_set_reverse).
# Same results; measured 621.8 / 623.1 -> 624.8 / 624.6 img/s (A/B in one gpurun call).  False restores first-to-last.
CONV_REVERSE = True


def _conv(m, x, res=None, reverse=False):
    if reverse and CONV_REVERSE:
        HF.call("sod_conv_set_reverse", 1)
        try:
            return HF.conv2d_fwd(x, m.w_bf16, m.bias_eff, res, m.stride, m.padding, m.dilation, relu=m.relu)
        finally:
            HF.call("sod_conv_set_reverse", 0)
    return HF.conv2d_fwd(x, m.w_bf16, m.bias_eff, res, m.stride, m.padding, m.dilation, relu=m.relu)


def _wgrad(m, g, x, arena):
    if m.weight.requires_grad and getattr(m, "groups", 1) > 1:
        m.wgrad_into(arena, g, x)
    elif m.weight.requires_grad:
        HF.conv2d_wgrad(g, x, arena.grad_view(m.weight), m.kernel_size, m.kernel_size, m.stride, m.padding, m.dilation, qscale=m.bn_scale)
        arena.mark_ready(m.weight)


def _dgrad(m, g, x, accum=None, relu_mask=None, relu_bits=None, reverse=False):
    if reverse and CONV_REVERSE:
        HF.call("sod_conv_set_reverse", 1)
        try:
            return HF.conv2d_dgrad(g, m.wt_bf16, (x.shape[1], x.shape[2]), m.stride, m.padding, m.dilation, accum=accum, relu_mask=relu_mask,
                                   relu_bits=relu_bits)
        finally:
            HF.call("sod_conv_set_reverse", 0)
    return HF.conv2d_dgrad(g, m.wt_bf16, (x.shape[1], x.shape[2]), m.stride, m.padding, m.dilation, accum=accum, relu_mask=relu_mask,
                           relu_bits=relu_bits)


def _scatter_even(comp, shape):
    """(N, H/2, W/2, C) gradient of the even (h, w) positions -> zero-stuffed (N, H, W, C)."""
    dx = torch.zeros(shape, dtype=comp.dtype, device=comp.device)
    dx[:, ::2, ::2, :] = comp
    return dx


def _dgrad_pair_into_input(blk, g, da, xin):
    """d(block input) = dgrad(shortcut, g) + dgrad(conv1, da).  When both are 1x1 stride-2 convolutions (STRIDE_IN_1X1) only the even
    (h, w) positions of the input receive gradient: the two data-gradients run as stride-1 GEMMs on the (Ho, Wo) grid (4x fewer MFMA
    work than masking the odd positions of the full grid) and the sum is scattered into a zero tensor."""
    sc, c1 = blk.shortcut, blk.conv1
    if all(m.kernel_size == 1 and m.stride == 2 and m.padding == 0 for m in (sc, c1)):
        Ho, Wo = g.shape[1], g.shape[2]
        part = HF.conv2d_dgrad(da, c1.wt_bf16, (Ho, Wo), 1, 0, 1)
        comp = HF.conv2d_dgrad(g, sc.wt_bf16, (Ho, Wo), 1, 0, 1, accum=part)
        dx = torch.zeros(xin.shape, dtype=comp.dtype, device=comp.device)
        dx[:, ::2, ::2, :] = comp
        return dx
    return _dgrad(sc, g, xin, accum=_dgrad(c1, da, xin))


def _block_fusable(blk):
    c1, c2, c3, sc = blk.conv1, blk.conv2, blk.conv3, blk.shortcut
    if not (c1.kernel_size == 1 and c1.stride == 1 and c1.padding == 0 and c1.out_channels == 64 and c1.relu
            and c2.kernel_size == 3 and c2.stride == 1 and c2.padding == 1 and c2.dilation == 1 and c2.out_channels == 64 and c2.relu
            and c3.kernel_size == 1 and c3.stride == 1 and c3.padding == 0 and c3.out_channels == 256 and c3.relu):
        return False
    if any(m.cin_pad for m in (c1, c2, c3)):
        return False
    if sc is None:      # the two block shapes of res2: identity on 256 channels, projection on the 64-channel stem output
        return c1.in_channels == 256
    return (c1.in_channels == 64 and sc.kernel_size == 1 and sc.stride == 1 and sc.padding == 0 and sc.out_channels == 256
            and not sc.relu and not sc.cin_pad)


def _fused_frozen_block(blk, x):
    """relu(conv3(conv2(conv1(x))) + shortcut(x)) of a block without a backward pass in one launch; None if an operand is unsuitable."""
    convs = [m for m in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut) if m is not None]
    for m in convs:
        m.prepare()
    if any(m.w_bf16.data_ptr() % 16 or not m.w_bf16.is_contiguous() for m in convs):
        return None
    key = tuple((m._prep_key, id(m.bias_eff)) for m in convs)
    if getattr(blk, "_fused_key", None) != key:
        def shift(m):
            return m.bias_eff if m.bias_eff is not None else torch.zeros(m.out_channels, dtype=torch.float32, device=x.device)
        b3 = shift(blk.conv3) + (shift(blk.shortcut) if blk.shortcut is not None else 0)
        blk._fused_bias = (shift(blk.conv1).contiguous(), shift(blk.conv2).contiguous(), b3.contiguous())
        blk._fused_key = key
    b1, b2, b3 = blk._fused_bias
    return HF.bottleneck_frozen_fwd(x, blk.conv1.w_bf16, b1, blk.conv2.w_bf16, b2, blk.conv3.w_bf16, b3,
                                    blk.shortcut.w_bf16 if blk.shortcut is not None else None)


class BottleneckStage(nn.Sequential):
    """nn.Sequential of BottleneckBlocks (same parameter names as a plain Sequential) executed as one fused autograd node; a stage
    that needs no backward pass (frozen under FREEZE_AT, or any stage under no_grad) whose blocks have the res2 shape runs one
    kernel per block instead."""

    def forward(self, x):
        no_bwd = not torch.is_grad_enabled() or not (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        # the fused kernel addresses its 256-channel output with 32-bit byte offsets (SOD_ESIZE above 2 GiB, e.g. >= 63 images of
        # 800x1333 per GPU); the generic per-conv path accepts destinations up to 8 GiB (sources stay below 2 GiB model-wide)
        if (BNECK_FUSED and no_bwd and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous()
                and x.shape[0] * x.shape[1] * x.shape[2] * 256 * 2 < 2 ** 31
                and all(_block_fusable(b) for b in self) and self[0].conv1.in_channels == x.shape[-1]):
            y = x
            for blk in self:
                nxt = _fused_frozen_block(blk, y)
                if nxt is None:
                    break
                y = nxt
            else:
                return y
        return _BottleneckStageFn.apply(x, self[0].conv1.weight, self)


class BasicBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = HipConv2d(in_channels, out_channels, 1, stride, 0, bias=False, frozen_bn=True)
        self.conv1 = HipConv2d(in_channels, out_channels, 3, stride, 1, bias=False, frozen_bn=True, relu=True, grad_premasked=True)
        self.conv2 = HipConv2d(out_channels, out_channels, 3, 1, 1, bias=False, frozen_bn=True, relu=True, mask_input=True)
        for m in (self.conv1, self.conv2, self.shortcut):
            if m is not None:
                m.init_msra()

    def forward(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        return self.conv2(self.conv1(x), res=sc)


class FrozenPrefix:
    """What ``ResNet.forward_frozen_prefix`` hands to ``ResNet.forward``: the activation behind the last frozen stage, the number of
    stages it covers and the named outputs produced on the way.  ``shape`` is the input batch's (N, Hp, Wp, C) so that callers that
    only look at ``images.tensor.shape`` keep working."""

    def __init__(self, tensor, n_stages, outputs, input_shape):
        self.tensor, self.n_stages, self.outputs, self.shape = tensor, n_stages, outputs, tuple(input_shape)


class ResNet(Backbone):
    def __init__(self, stem, stages, out_features):
        super().__init__()
        self.stem = stem
        self._out_feature_strides = {"stem": stem.stride}
        self._out_feature_channels = {"stem": stem.out_channels}
        self.stages_and_names = []
        cur_stride = stem.stride
        for i, blocks in enumerate(stages):
            name = "res" + str(i + 2)
            stage = BottleneckStage(*blocks) if all(isinstance(b, BottleneckBlock) for b in blocks) else nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            cur_stride = int(cur_stride * blocks[0].stride)
            self._out_feature_strides[name] = cur_stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        self._out_features = out_features

    def forward(self, x):
        outputs, start = {}, 0
        DeferSlot.reset()
        if isinstance(x, FrozenPrefix):
            outputs.update(x.outputs)
            start, x = x.n_stages, x.tensor
        else:
            x = self.stem(x)
            if "stem" in self._out_features:
                outputs["stem"] = x
        for stage, name in self.stages_and_names[start:]:
            x = stage(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def frozen_prefix_len(self):
        """Number of leading stages (behind a frozen stem) without a trainable parameter; -1 if the stem itself trains."""
        if any(p.requires_grad for p in self.stem.parameters()):
            return -1
        n = 0
        for stage, _ in self.stages_and_names:
            if any(p.requires_grad for p in stage.parameters()):
                break
            n += 1
        return n

    @torch.no_grad()
    def prepare_frozen_prefix(self):
        """Refresh the folded bf16 compute copies of the stem and of the frozen leading stages on the CURRENT stream (a no-op unless a
        weight or FrozenBN buffer changed): ``forward_frozen_prefix`` may then run on any stream without allocating or writing them."""
        n = self.frozen_prefix_len()
        if n < 0:
            return
        mods = list(self.stem.modules())
        for stage, _ in self.stages_and_names[:n]:
            mods += list(stage.modules())
        for m in mods:
            if isinstance(m, HipConv2d):
                m.prepare()

    @torch.no_grad()
    def forward_frozen_prefix(self, x):
        """Stem + the frozen leading stages (FREEZE_AT) of ``x``.  They have no gradient and their weights never change, so a training
        loop may run them for the NEXT batch while the current one is in backward (meta-arch ``prefetch``).  None if nothing is frozen."""
        n = self.frozen_prefix_len()
        if n < 0:
            return None
        shape = x.shape
        outputs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for stage, name in self.stages_and_names[:n]:
            x = stage(x)
            if name in self._out_features:
                outputs[name] = x
        return FrozenPrefix(x, n, outputs, shape)

    def freeze(self, freeze_at=0):
        """FREEZE_AT: 1 = stem, 2 = stem + res2, ..."""
        if freeze_at >= 1:
            for p in self.stem.parameters():
                p.requires_grad = False
        for idx, (stage, _) in enumerate(self.stages_and_names, start=2):
            if freeze_at >= idx:
                for p in stage.parameters():
                    p.requires_grad = False
        return self


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    depth = cfg.MODEL.RESNETS.DEPTH
    out_features = cfg.MODEL.RESNETS.OUT_FEATURES
    norm = cfg.MODEL.RESNETS.NORM
    if norm != "FrozenBN":
        raise NotImplementedError(f"MODEL.RESNETS.NORM={norm}: only FrozenBN (the default the FCOS/RetinaNet configs use) is built")
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    if num_groups != 1 and depth in (18, 34):
        raise NotImplementedError("NUM_GROUPS > 1 needs bottleneck blocks (detectron2 asserts the same for R18 / R34)")
    deform_on = list(cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE)
    if any(deform_on) and depth in (18, 34):
        raise NotImplementedError("DEFORM_ON_PER_STAGE needs bottleneck blocks (detectron2 asserts the same for R18 / R34)")
    stem = BasicStem(input_shape.channels, cfg.MODEL.RESNETS.STEM_OUT_CHANNELS)
    blocks_per_stage = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    in_ch = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
    out_ch = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    bott = cfg.MODEL.RESNETS.NUM_GROUPS * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    if depth in (18, 34):
        assert out_ch == 64, "Must set MODEL.RESNETS.RES2_OUT_CHANNELS = 64 for R18/R34"
    stage_names = ["res2", "res3", "res4", "res5"]
    max_stage = max(stage_names.index(f) + 2 for f in out_features if f in stage_names)
    stages = []
    for idx, stage_idx in enumerate(range(2, max_stage + 1)):
        dilation = cfg.MODEL.RESNETS.RES5_DILATION if stage_idx == 5 else 1
        first_stride = 1 if idx == 0 or (stage_idx == 5 and dilation == 2) else 2
        blocks = []
        for b in range(blocks_per_stage[idx]):
            stride = first_stride if b == 0 else 1
            if depth in (18, 34):
                blocks.append(BasicBlock(in_ch, out_ch, stride))
            elif idx < len(deform_on) and deform_on[idx]:
                blocks.append(DeformBottleneckBlock(in_ch, out_ch, bott, stride, cfg.MODEL.RESNETS.STRIDE_IN_1X1, dilation,
                                                    deform_modulated=cfg.MODEL.RESNETS.DEFORM_MODULATED,
                                                    deform_num_groups=cfg.MODEL.RESNETS.DEFORM_NUM_GROUPS, num_groups=cfg.MODEL.RESNETS.NUM_GROUPS))
            else:
                blocks.append(BottleneckBlock(in_ch, out_ch, bott, stride, cfg.MODEL.RESNETS.STRIDE_IN_1X1, dilation, num_groups=num_groups))
            in_ch = out_ch
        out_ch *= 2
        bott *= 2
        stages.append(blocks)
    return ResNet(stem, stages, out_features).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)
