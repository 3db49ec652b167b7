"""``build_model`` + META_ARCH_REGISTRY (detectron2.modeling.meta_arch surface re-exported by
slender_det/modeling/meta_arch/__init__.py:1 and used by slender_det/engine/defaults.py:137-149)."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    """``META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)`` moved to ``cfg.MODEL.DEVICE``; on a GPU the
    trainable parameters are then re-homed into the flat arena (layers/arena.py)."""
    from ...layers.arena import ParamArena
    from ...layers.nn import attach_arena

    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    if torch.device(cfg.MODEL.DEVICE).type == "cuda":
        arena = ParamArena(model)
        attach_arena(model, arena)
        arena.setup_batched_prep(model)
        model.arena = arena
    return model
