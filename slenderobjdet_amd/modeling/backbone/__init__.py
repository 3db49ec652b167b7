from .build import BACKBONE_REGISTRY, Backbone, build_backbone
from .fpn import FPN, LastLevelP6P7, build_retinanet_resnet_fpn_backbone, build_retinanet_resnet_fpn_backbone_use_p5
from .resnet import ResNet, build_resnet_backbone
