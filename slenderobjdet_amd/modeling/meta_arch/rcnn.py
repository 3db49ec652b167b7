"""GeneralizedRCNN on the HIP kernels — the two-stage meta-architecture that BASELINE config 5
(configs/rotated/faster_R_101.yaml over Base-RRCNN-FPN.yaml) selects; detectron2's source is absent, the contract is the one the
reference's own subclass relies on (slender_det/modeling/meta_arch/rcnn/pvrcnn.py:17-64): ``forward(batched_inputs)`` returns
``{loss_rpn_cls, loss_rpn_loc, loss_cls, loss_box_reg}`` in training and ``[{"instances": Instances}]`` in eval."""
import os

import torch
from torch import nn

from ...layers import nn as _nn
from ..backbone import build_backbone
from ..proposal_generator import build_proposal_generator
from ..roi_heads import build_roi_heads
from .build import META_ARCH_REGISTRY
from .fcos import FCOSV2


# GRAD_PARK = False: autograd adds the ROI pooler's and the RPN head's feature gradients itself (one elementwise pass per level)
GRAD_PARK = True


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.proposal_generator = build_proposal_generator(cfg, shapes)
        self.roi_heads = build_roi_heads(cfg, shapes)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    preprocess_image = FCOSV2.preprocess_image
    prefetch, _take_prefetched = FCOSV2.prefetch, FCOSV2._take_prefetched
    postprocess = FCOSV2.postprocess

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        features = self.backbone(images.tensor)
        # both heads read the FPN outputs: the ROI pooler's feature gradients ride in the RPN head's data-gradient launch (layers/nn.py GradPark)
        _nn.GradPark.current = _nn.GradPark() if GRAD_PARK and torch.is_grad_enabled() else None
        try:
            proposals, proposal_losses = self.proposal_generator(images, features, gt_instances)
            _, detector_losses = self.roi_heads(images, features, proposals, gt_instances)
        finally:
            _nn.GradPark.current = None
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        proposals, _ = self.proposal_generator(images, features, None)
        results, _ = self.roi_heads(images, features, proposals, None)
        if do_postprocess:
            return self.postprocess(results, batched_inputs, images.image_sizes)
        return results


@META_ARCH_REGISTRY.register()
class ProposalVisibleRCNN(GeneralizedRCNN):
    """slender_det/modeling/meta_arch/rcnn/pvrcnn.py:10-63: inference results carry the (rescaled) proposals next to the instances."""

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        from ..postprocessing import detector_postprocess

        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        proposals, _ = self.proposal_generator(images, features, None)
        results, _ = self.roi_heads(images, features, proposals, None)
        if not do_postprocess:
            return results
        out = []
        for res, prop, inp, size in zip(results, proposals, batched_inputs, images.image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            out.append({"instances": detector_postprocess(res, h, w), "proposals": detector_postprocess(prop, h, w)})
        return out


@META_ARCH_REGISTRY.register()
class ProposalNetwork(nn.Module):
    """d2 ProposalNetwork: backbone + proposal generator only (loss_rpn_cls / loss_rpn_loc)."""

    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    preprocess_image = FCOSV2.preprocess_image
    prefetch, _take_prefetched = FCOSV2.prefetch, FCOSV2._take_prefetched

    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        gt = [x["instances"].to(self.device) for x in batched_inputs] if "instances" in batched_inputs[0] else None
        proposals, losses = self.proposal_generator(images, features, gt)
        if self.training:
            return losses
        return [{"proposals": p} for p in proposals]
