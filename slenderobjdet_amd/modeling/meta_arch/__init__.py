from .build import META_ARCH_REGISTRY, build_model
from .fcos import FCOS, FCOSV2, FCOSHead
from .retinanet import RetinaNet, RetinaNetHead
from .reppoints import RepPointsDetector
from .rcnn import GeneralizedRCNN, ProposalNetwork, ProposalVisibleRCNN
from .meta import MEAT_HEADS_REGISTRY, AblationMetaArch, AnchorHead, LRTBHead, LRTBTopkHead, PointSetHead
