"""``import slenderobjdet_amd.dropin`` makes the reference's import paths resolve to this package, so its ``train_net.py``
(train_net.py:23-38 imports ``detectron2.*`` and ``slender_det.*``) and ``configs/*`` run unchanged on MI355X:

    python -c "import slenderobjdet_amd.dropin, runpy, sys; sys.argv = ['train_net.py', '--config-file', 'configs/fcos/fcos_R_50_FPN_1x.yaml', \
               '--num-gpus', '8']; runpy.run_path('train_net.py', run_name='__main__')"

Only the training hot path is backed by real code; evaluation / TTA / dataset names resolve to objects that raise
``NotImplementedError`` when used (out of scope).
"""
import sys
import types

from . import config as _config
from . import engine as _engine
from . import modeling as _modeling
from . import solver as _solver
from . import structures as _structures
from .engine import hooks as _hooks
from .layers import deform_conv as _dcn
from .layers import losses as _losses
from .layers import nms as _nms
from .layers import nn as _nn
from .modeling import backbone as _backbone
from .modeling import meta_arch as _meta_arch
from .modeling import postprocessing as _post
from .modeling.shape_spec import ShapeSpec
from .utils import comm as _comm
from .utils import registry as _registry


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _unavailable(what):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{what} is outside the training hot path covered by slenderobjdet_amd")

    _Missing.__name__ = what.split(".")[-1]
    return _Missing


def _cat(tensors, dim=0):
    import torch

    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


class _Metadata(dict):
    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)


class _MetadataCatalog:
    _d = {}

    @classmethod
    def get(cls, name):
        return cls._d.setdefault(name, _Metadata(name=name, evaluator_type="coco"))


def install():
    # ---- detectron2 ----
    _mod("detectron2")
    _mod("detectron2.config", CfgNode=_config.CfgNode, get_cfg=_config.get_cfg)
    _mod("detectron2.utils")
    sys.modules["detectron2.utils.comm"] = _comm
    sys.modules["detectron2.utils"].comm = _comm
    _mod("detectron2.utils.registry", Registry=_registry.Registry)
    _mod("detectron2.utils.events", get_event_storage=lambda: types.SimpleNamespace(put_scalar=lambda *a, **k: None, put_image=lambda *a, **k: None))
    _mod("detectron2.engine", default_argument_parser=_engine.default_argument_parser, launch=_engine.launch, hooks=_hooks,
         DefaultTrainer=_engine.DefaultTrainer, default_setup=_engine.default_setup)
    sys.modules["detectron2.engine.hooks"] = _hooks
    _mod("detectron2.data", MetadataCatalog=_MetadataCatalog)
    _mod("detectron2.evaluation", **{n: _unavailable("detectron2.evaluation." + n) for n in
                                     ("COCOEvaluator", "DatasetEvaluator", "DatasetEvaluators", "RotatedCOCOEvaluator")},
         print_csv_format=lambda *a, **k: None, verify_results=lambda *a, **k: None)
    _mod("detectron2.structures", Boxes=_structures.Boxes, Instances=_structures.Instances, ImageList=_structures.ImageList,
         pairwise_iou=_structures.pairwise_iou)
    _mod("detectron2.layers", ShapeSpec=ShapeSpec, cat=_cat, batched_nms=_nms.batched_nms, DeformConv=_dcn.DeformConv,
         ModulatedDeformConv=_dcn.ModulatedDeformConv, Conv2d=_nn.HipConv2d)
    _mod("detectron2.modeling", META_ARCH_REGISTRY=_meta_arch.META_ARCH_REGISTRY, BACKBONE_REGISTRY=_backbone.BACKBONE_REGISTRY,
         build_model=_meta_arch.build_model, build_backbone=_backbone.build_backbone, GeneralizedRCNNWithTTA=_unavailable("detectron2.modeling.GeneralizedRCNNWithTTA"))
    _mod("detectron2.modeling.meta_arch", META_ARCH_REGISTRY=_meta_arch.META_ARCH_REGISTRY, build_model=_meta_arch.build_model)
    _mod("detectron2.modeling.backbone", BACKBONE_REGISTRY=_backbone.BACKBONE_REGISTRY, build_backbone=_backbone.build_backbone,
         Backbone=_backbone.Backbone, FPN=_backbone.FPN, build_resnet_backbone=_backbone.build_resnet_backbone)
    _mod("detectron2.modeling.postprocessing", detector_postprocess=_post.detector_postprocess)
    # ---- fvcore ----
    _mod("fvcore")
    _mod("fvcore.nn", sigmoid_focal_loss_jit=_losses.sigmoid_focal_loss_jit, sigmoid_focal_loss=_losses.sigmoid_focal_loss)
    # ---- slender_det ----
    _mod("slender_det")
    sys.modules["slender_det.config"] = _config
    sys.modules["slender_det"].config = _config
    _mod("slender_det.engine", BaseTrainer=_engine.BaseTrainer, default_setup=_engine.default_setup, hooks=_hooks)
    _mod("slender_det.modeling", build_model=_meta_arch.build_model, build_backbone=_backbone.build_backbone,
         META_ARCH_REGISTRY=_meta_arch.META_ARCH_REGISTRY, BACKBONE_REGISTRY=_backbone.BACKBONE_REGISTRY)
    _mod("slender_det.modeling.backbone", build_backbone=_backbone.build_backbone, BACKBONE_REGISTRY=_backbone.BACKBONE_REGISTRY)
    _mod("slender_det.solver", build_optimizer=_solver.build_optimizer, get_default_optimizer_params=_solver.get_default_optimizer_params)
    _mod("slender_det.layers", Scale=_nn.Scale, iou_loss=_losses.iou_loss, DFConv2d=_dcn.DFConv2d)
    _mod("slender_det.checkpoint", DetectionCheckpointer=_engine.defaults._Checkpointer)
    _mod("slender_det.evaluation", COCOEvaluator=_unavailable("slender_det.evaluation.COCOEvaluator"),
         inference_on_dataset=_unavailable("slender_det.evaluation.inference_on_dataset"))


install()
