"""``slender_det.modeling`` surface for the hot path (reference: slender_det/modeling/__init__.py:1-30)."""
from .backbone import BACKBONE_REGISTRY, Backbone, build_backbone
from .meta_arch import META_ARCH_REGISTRY, build_model
from .shape_spec import ShapeSpec
