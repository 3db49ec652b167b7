"""FPN + LastLevelP6P7 and the registered builders of slender_det/modeling/backbone/fpn.py:94-115.

Restates detectron2 ``FPN`` (source absent): per level a 1x1 lateral and a 3x3 output
conv (bias when norm == ""), top-down nearest-2x upsample + sum.  The upsample+add is fused into the lateral conv's
epilogue (``res_up2``), so the top-down path costs no extra memory pass.
"""
import math

import torch
from torch import nn

from ...layers.nn import HipConv2d, HipGroupNorm, add_up2, group_norm_relu, relu
from ..shape_spec import ShapeSpec
from .build import BACKBONE_REGISTRY, Backbone
from .resnet import build_resnet_backbone


class LastLevelP6P7(nn.Module):
    def __init__(self, in_channels, out_channels, in_feature="res5"):
        super().__init__()
        self.num_levels = 2
        self.in_feature = in_feature
        self.p6 = HipConv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = HipConv2d(out_channels, out_channels, 3, 2, 1)
        self.p6.init_xavier()
        self.p7.init_xavier()

    def forward(self, c5):
        p6 = self.p6(c5)
        p7 = self.p7(relu(p6))
        return [p6, p7]


class _SubsampleFn(torch.autograd.Function):
    """F.max_pool2d(x, kernel_size=1, stride=2, padding=0) == x[:, ::2, ::2] (NHWC): a strided copy and its scatter-back."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x[:, ::2, ::2, :].contiguous()

    @staticmethod
    def backward(ctx, dy):
        dx = torch.zeros(ctx.shape, dtype=dy.dtype, device=dy.device)
        dx[:, ::2, ::2, :] = dy
        return dx


class LastLevelMaxPool(nn.Module):
    """d2 LastLevelMaxPool: P6 = max_pool2d(P5, kernel 1, stride 2) (build_resnet_fpn_backbone, used by the R-CNN configs)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [_SubsampleFn.apply(x)]


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        if norm not in ("", "GN"):
            raise NotImplementedError(f"FPN.NORM {norm!r}: only '' and 'GN' (configs/rep-points/*.yaml) are built")
        self.norm = norm
        if fuse_type != "sum":
            raise NotImplementedError("FPN.FUSE_TYPE avg is not built")
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels = [input_shapes[f].channels for f in in_features]
        for i, s in enumerate(strides[1:], 1):
            assert s == 2 * strides[i - 1], f"Strides {s} {strides[i - 1]} are not log2 contiguous"
        lateral, output = [], []
        for idx, ch in enumerate(in_channels):
            use_bias = norm == ""          # d2 FPN: the convs lose their bias when a norm follows
            lat = HipConv2d(ch, out_channels, 1, 1, 0, bias=use_bias)
            out = HipConv2d(out_channels, out_channels, 3, 1, 1, bias=use_bias)
            lat.init_xavier()
            out.init_xavier()
            stage = int(math.log2(strides[idx]))
            self.add_module(f"fpn_lateral{stage}", lat)
            self.add_module(f"fpn_output{stage}", out)
            if norm == "GN":               # get_norm("GN", C) = GroupNorm(32, C), applied as Conv2d(..., norm=...) does
                lat.norm = HipGroupNorm(32, out_channels)
                out.norm = HipGroupNorm(32, out_channels)
            lateral.append(lat)
            output.append(out)
        self.lateral_convs = lateral[::-1]      # top (coarsest) first
        self.output_convs = output[::-1]
        self.top_block = top_block
        self.in_features = in_features
        self.bottom_up = bottom_up
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in strides}
        if top_block is not None:
            stage = int(math.log2(strides[-1]))
            for s in range(stage, stage + top_block.num_levels):
                self._out_feature_strides[f"p{s + 1}"] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        feats = self.bottom_up(x)
        xs = [feats[f] for f in self.in_features[::-1]]
        results = []
        if self.norm == "":
            prev = self.lateral_convs[0](xs[0])
            results.append(self.output_convs[0](prev))
            for feat, lat, out in zip(xs[1:], self.lateral_convs[1:], self.output_convs[1:]):
                prev = lat(feat, res=prev, res_up2=True)
                results.insert(0, out(prev))
        else:
            # (conv + GroupNorm as one node with epilogue statistics measured slower here - 436.4 vs 441.4 img/s on RepPoints R50, round 3 -
            # and left the tree in round 5: these short convolutions lose more to the statistics' atomics than the pass costs)
            cg = {id(c): (lambda t, c=c: group_norm_relu(c(t), c.norm, relu=False)) for c in self.lateral_convs + self.output_convs}
            prev = cg[id(self.lateral_convs[0])](xs[0])
            results.append(cg[id(self.output_convs[0])](prev))
            for feat, lat, out in zip(xs[1:], self.lateral_convs[1:], self.output_convs[1:]):
                prev = add_up2(cg[id(lat)](feat), prev)
                results.insert(0, cg[id(out)](prev))
        if self.top_block is not None:
            src = feats.get(self.top_block.in_feature)
            if src is None:
                src = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(src))
        assert len(self._out_features) == len(results)
        return dict(zip(self._out_features, results))


def _build(cfg, input_shape, p6p7_from_p5):
    bottom_up = build_resnet_backbone(cfg, input_shape)
    out_channels = cfg.MODEL.FPN.OUT_CHANNELS
    if p6p7_from_p5:
        top = LastLevelP6P7(out_channels, out_channels, in_feature="p5")
    else:
        top = LastLevelP6P7(bottom_up.output_shape()["res5"].channels, out_channels, in_feature="res5")
    return FPN(bottom_up, cfg.MODEL.FPN.IN_FEATURES, out_channels, cfg.MODEL.FPN.NORM, top, cfg.MODEL.FPN.FUSE_TYPE)


@BACKBONE_REGISTRY.register()
def build_retinanet_resnet_fpn_backbone_use_p5(cfg, input_shape: ShapeSpec):
    """slender_det/modeling/backbone/fpn.py:94-115 (what configs/fcos/Base-Fcos.yaml:4 selects)."""
    return _build(cfg, input_shape, True)


@BACKBONE_REGISTRY.register()
def build_retinanet_resnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    """detectron2's builder (P6 from res5), used by configs/retina/Base-RetinaNet.yaml."""
    return _build(cfg, input_shape, False)


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    """detectron2's builder for the R-CNN family (configs/rotated/Base-RRCNN-FPN.yaml:5): FPN over res2..res5 + LastLevelMaxPool."""
    bottom_up = build_resnet_backbone(cfg, input_shape)
    return FPN(bottom_up, cfg.MODEL.FPN.IN_FEATURES, cfg.MODEL.FPN.OUT_CHANNELS, cfg.MODEL.FPN.NORM, LastLevelMaxPool(), cfg.MODEL.FPN.FUSE_TYPE)
