"""AblationMetaArch + PointSetHead on the HIP kernels.

Mirror of slender_det/modeling/meta_arch/meta/meta.py:25-152 (``AblationMetaArch``: backbone + a head from ``MEAT_HEADS_REGISTRY``
selected by ``cfg.MODEL.META_ARCH.NAME``) and meta/heads/pointset_head.py:19-581 with meta_head.py:21-104 (``PointSetHead``: unified
cls / loc towers, init points, feature adaption in {Empty, Unsupervised Offset, Split Unsup Offset, Supervised Offset}, refine
points; losses ``loss_cls`` / ``loss_pts_init`` / ``loss_pts_refine``).

The head is RepPoints-shaped, so it reuses the RepPoints machinery (modeling/meta_arch/reppoints.py): multi-level tower launches,
the point-row kernels (dcn offsets, points2bbox with arg indices), the fused point matcher, IoU + MaxIoU assignment through the
anchor-match kernel, and the one-node loss function.  Differences that are reproduced (pointset_head.py line numbers):
  * "Supervised Offset" subtracts the dcn base from the UN-flipped points (:137-140), unlike rpd.py:621-635;
  * points are scaled by the FPN stride itself (:232-234) and normalised by POINT_BASE_SCALE * stride (:228);
  * refine labels come from MaxIoU(pos 0.5, neg 0.4, gt_max_matching) on init boxes clamped to >= 0 (:415-470): the 0.4-0.5 band
    is background, not ignored; nothing is masked by the image size;
  * normalisers are the per-batch positive counts (:301-312), no EMA.
``LRTBHead`` / ``LRTBTopkHead`` / ``AnchorHead`` are not built yet.
"""
import math

import torch
from torch import nn

from ...layers import functional as HF
from ...layers.deform_conv import DeformConv
from ...layers.nn import ConvGnRelu, ConvML, ConvReluML, HipConv2d
from ...structures import Boxes, Instances
from ...utils.registry import Registry
from ..backbone import build_backbone
from .build import META_ARCH_REGISTRY
from .fcos import FCOSV2
from .reppoints import RepPointsDetector, _DcnOffsetFn, _RepPointsLossFn

MEAT_HEADS_REGISTRY = Registry("META_HEADS")      # (sic) the reference's spelling, meta_head.py:9
FEAT_ADAPTION_METHODS = ["Empty", "Unsupervised Offset", "Supervised Offset", "Split Unsup Offset"]


@MEAT_HEADS_REGISTRY.register()
class PointSetHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        h = cfg.MODEL.META_ARCH
        self.in_channels = input_shape[0].channels
        self.in_features, self.fpn_strides = list(h.IN_FEATURES), list(h.FPN_STRIDES)
        self.strides = self.fpn_strides
        self.num_classes, self.feat_channels, self.stacked_convs, self.norm = h.NUM_CLASSES, h.FEAT_CHANNELS, h.STACK_CONVS, h.NORM
        self.feat_adaption, self.res_refine = h.FEAT_ADAPTION, h.RES_REFINE
        self.loc_feat_channels, self.gradient_mul, self.prior_prob = h.LOC_FEAT_CHANNELS, h.GRADIENT_MUL, h.PRIOR_PROB
        self.focal_loss_gamma, self.focal_loss_alpha = h.FOCAL_LOSS_GAMMA, h.FOCAL_LOSS_ALPHA
        self.loss_cls_weight, self.loss_init_weight, self.loss_refine_weight = h.LOSS_CLS_WEIGHT, h.LOSS_LOC_INIT_WEIGHT, h.LOSS_LOC_REFINE_WEIGHT
        self.score_threshold, self.topk_candidates, self.nms_threshold = h.SCORE_THRESH_TEST, h.TOPK_CANDIDATES_TEST, h.NMS_THRESH_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.num_points, self.point_base_scale = h.NUM_POINTS, h.POINT_BASE_SCALE
        if h.TRANSFORM_METHOD != "minmax":
            raise NotImplementedError(f"META_ARCH.TRANSFORM_METHOD {h.TRANSFORM_METHOD!r}: only 'minmax' (the default) is built")
        if self.feat_adaption not in FEAT_ADAPTION_METHODS:
            raise AssertionError(f"{self.feat_adaption} {type(self.feat_adaption)}")
        if self.norm not in ("GN", ""):
            raise NotImplementedError(f"META_ARCH.NORM {self.norm!r}: only 'GN' and '' are built")
        C = self.feat_channels
        assert self.in_channels == C == self.loc_feat_channels == 256 and self.num_classes % 8 == 0, "PointSetHead is built for 256-channel features"
        assert self.num_points == 9, "3x3 deformable kernels: NUM_POINTS must be 9"
        self.pts_ld = (2 * self.num_points + 7) // 8 * 8
        self.point_scales = self.fpn_strides               # pts * stride + centre (pointset_head.py:232-234)
        self.smooth_l1_beta = 0.11
        unit = ConvGnRelu if self.norm == "GN" else ConvReluML
        self.cls_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_init_conv = ConvML(C, C, 3, 1, relu=True)                    # + F.relu_ (:116)
        self.loc_init_out = ConvML(C, self.pts_ld, 1, 0, out_f32=True)
        if self.feat_adaption == "Empty":                                    # + F.relu_ before the output convs (:145-149)
            self.cls_conv = ConvML(C, C, 3, 1, relu=True)
            self.loc_refine_conv = ConvML(C, C, 3, 1, relu=True)
        else:
            self.cls_conv = DeformConv(C, C, 3, 1, 1, relu=True)
            self.loc_refine_conv = DeformConv(C, C, 3, 1, 1, relu=True)
        if self.feat_adaption == "Unsupervised Offset":
            self.offset_conv = ConvML(C, self.pts_ld, 1, 0, out_f32=True)
        elif self.feat_adaption == "Split Unsup Offset":
            self.offset_conv_cls = ConvML(C, self.pts_ld, 1, 0, out_f32=True)
            self.offset_conv_loc = ConvML(C, self.pts_ld, 1, 0, out_f32=True)
        self.logits = HipConv2d(C, self.num_classes, 1, 1, 0, bias=True)             # cls_out
        self.offsets_refine = HipConv2d(C, self.pts_ld, 1, 1, 0, bias=True)          # loc_refine_out
        npt = 2 * self.num_points
        with torch.no_grad():            # meta_head.py:64-72, pointset_head.py:66-88 (DeformConv keeps its own init)
            for u in list(self.cls_subnet) + list(self.loc_subnet):
                u.conv.init_normal(0.01, 0.0)
            mods = [self.loc_init_conv.conv, self.loc_init_out.conv, self.logits, self.offsets_refine]
            mods += [m.conv for m in (self.cls_conv, self.loc_refine_conv) if isinstance(m, ConvML)]
            mods += [getattr(self, n).conv for n in ("offset_conv", "offset_conv_cls", "offset_conv_loc") if hasattr(self, n)]
            for m in mods:
                m.init_normal(0.01, 0.0)
            for m in (self.loc_init_out.conv, self.offsets_refine) + tuple(getattr(self, n).conv for n in ("offset_conv", "offset_conv_cls",
                                                                                                       "offset_conv_loc") if hasattr(self, n)):
                m.weight[npt:].zero_()
            self.logits.bias.fill_(-math.log((1 - self.prior_prob) / self.prior_prob))
        self.register_buffer("loss_normalizer", torch.zeros(1))     # holds the refine positive count of the last step (no EMA here)
        self.loss_normalizer_momentum = 0.0
        self._grid_cache = {}
        self.last_targets = None

    @property
    def device(self):
        return self.loss_normalizer.device

    # the RepPoints node reads these
    point_grid = RepPointsDetector.point_grid
    predict = RepPointsDetector.predict

    def box_norm(self, strides):            # normalize_term = POINT_BASE_SCALE * stride (:228); the kernel divides by 4 * value
        return strides * (self.point_base_scale / 4.0)

    def normalizer_images(self, n):         # max(1, num_pos_refine) (:301-303)
        return 1

    def run_head(self, features):
        cls_f, loc_f = list(features), list(features)
        for u in self.cls_subnet:
            cls_f = u(cls_f)
        for u in self.loc_subnet:
            loc_f = u(loc_f)
        oi = self.loc_init_out(self.loc_init_conv(loc_f))
        if self.feat_adaption == "Empty":
            return oi, self.cls_conv(cls_f), self.loc_refine_conv(loc_f)
        nl = len(features)
        if self.feat_adaption == "Unsupervised Offset":
            off = self.offset_conv(loc_f)
            off_c = off_l = off
        elif self.feat_adaption == "Split Unsup Offset":
            off_c, off_l = self.offset_conv_cls(loc_f), self.offset_conv_loc(loc_f)
        else:   # Supervised Offset: grad_mul(loc_out_init) - dcn_base_offset, channel order untouched (:137-140)
            off_c = off_l = [_DcnOffsetFn.apply(oi[l], self.num_points, self.gradient_mul, False) for l in range(nl)]
        cf = [self.cls_conv(cls_f[l], off_c[l], off_ld=self.pts_ld) for l in range(nl)]
        rf = [self.loc_refine_conv(loc_f[l], off_l[l], off_ld=self.pts_ld) for l in range(nl)]
        return oi, cf, rf

    @torch.no_grad()
    def get_ground_truth(self, centers, strides, lvl_start, init_boxes, gt_instances, image_sizes):
        """point_targets (:352-412) for the init boxes, bbox_targets (:415-470) on the clamped init boxes for cls / refine."""
        dev = centers.device
        N, X = init_boxes.shape[:2]
        counts = [len(g) for g in gt_instances]
        if min(counts) == 0:
            raise ValueError("No gt or bboxes")                      # pointset_head.py:366-367
        box_off = torch.tensor([0] + counts, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev, non_blocking=True)
        boxes = torch.cat([g.gt_boxes.tensor for g in gt_instances]).float().contiguous()
        classes = torch.cat([g.gt_classes for g in gt_instances]).to(torch.int32).contiguous()
        obj, init_lab = HF.reppoints_point_match(centers, strides, lvl_start, boxes, box_off, N, max(counts), "points", float(self.point_base_scale))
        cand = init_boxes.clamp(min=0).contiguous()                  # candidate_bboxes[:, k].clamp_(min=0)  (:436-439)
        vals = torch.empty((N, X), dtype=torch.float32, device=dev)
        matches = torch.empty((N, X), dtype=torch.int32, device=dev)
        mlab = torch.empty((N, X), dtype=torch.int8, device=dev)
        b0 = 0
        for i, c in enumerate(counts):
            HF.anchor_match(boxes[b0:b0 + c], cand[i], [0.4, 0.5], [0, 0, 1], True, out=(vals[i], matches[i], mlab[i]))
            b0 += c
        never = torch.full((N, 2), 3.0e38, dtype=torch.float32, device=dev)      # no masking by the image size in this head
        cls, refine_lab = HF.reppoints_labels(matches, mlab, boxes, classes, box_off, centers, never, self.num_classes, None)
        return obj, init_lab, cls, refine_lab

    def forward(self, images, features, gt_instances=None):
        oi, cf, rf = self.run_head(features)
        if self.training:
            out3 = _RepPointsLossFn.apply(self, self.logits.weight, gt_instances, images.image_sizes, *oi, *cf, *rf)
            return {"loss_cls": out3[0] * self.loss_cls_weight, "loss_pts_init": out3[1], "loss_pts_refine": out3[2] * self.loss_refine_weight}
        with torch.no_grad():
            logits, _, _, _, refine_boxes, _, geo = self.predict(oi, cf, rf)
            return self.inference(logits, refine_boxes, geo, images.image_sizes)

    @torch.no_grad()
    def inference(self, logits, refine_boxes, geo, image_sizes):
        """pointset_head.py:472-581: per level sigmoid over (HW x K), top-k, threshold, clamp to the image, class-aware NMS."""
        from ...layers.nms import batched_nms

        hw, offs, X = geo
        bounds = list(offs) + [X]
        K = self.num_classes
        results = []
        for i, image_size in enumerate(image_sizes):
            B, S, C = [], [], []
            for l in range(len(hw)):
                sl = slice(bounds[l], bounds[l + 1])
                box = refine_boxes[i, sl].clone()
                box[:, 0::2].clamp_(min=0, max=image_size[1])
                box[:, 1::2].clamp_(min=0, max=image_size[0])
                p = logits[i, sl].flatten().sigmoid()
                k = min(self.topk_candidates, p.shape[0])
                prob, idx = p.sort(descending=True)
                prob, idx = prob[:k], idx[:k]
                keep = prob > self.score_threshold
                prob, idx = prob[keep], idx[keep]
                B.append(box[idx // K]); S.append(prob); C.append(idx % K)
            B, S, C = torch.cat(B), torch.cat(S), torch.cat(C)
            keep = batched_nms(B, S, C, self.nms_threshold)[: self.max_detections_per_image]
            r = Instances(tuple(image_size))
            r.pred_boxes, r.scores, r.pred_classes = Boxes(B[keep]), S[keep], C[keep]
            results.append(r)
        return results


def build_meta_head(cfg, input_shape):
    return MEAT_HEADS_REGISTRY.get(cfg.MODEL.META_ARCH.NAME)(cfg, input_shape)


@META_ARCH_REGISTRY.register()
class AblationMetaArch(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.head = build_meta_head(cfg, [shapes[f] for f in cfg.MODEL.META_ARCH.IN_FEATURES])
        self.input_format, self.vis_period = cfg.INPUT.FORMAT, cfg.VIS_PERIOD
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    preprocess_image = FCOSV2.preprocess_image
    postprocess = FCOSV2.postprocess

    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        features = [features[f] for f in self.head.in_features]
        if not self.training:
            return self.postprocess(self.head(images, features), batched_inputs, images.image_sizes)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        return self.head(images, features, gt_instances)
