"""Backbone registry + base class (detectron2.modeling.backbone surface re-exported at
slender_det/modeling/backbone/__init__.py:1-10)."""
from torch import nn

from ...utils.registry import Registry
from ..shape_spec import ShapeSpec

BACKBONE_REGISTRY = Registry("BACKBONE")


class Backbone(nn.Module):
    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone
