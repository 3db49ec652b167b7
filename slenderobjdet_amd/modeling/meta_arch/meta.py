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


def _run_tower(units, xs):
    """A tower of [conv -> (GroupNorm) -> ReLU] units: a [conv -> ReLU] unit is told which unit produced its input (that unit's ONLY
    consumer), so that its data gradient can take over the producer's ReLU backward (layers/nn.py _ReluToken)."""
    prev = None
    for u in units:
        xs = u(xs, chained=prev if isinstance(prev, ConvReluML) else None) if isinstance(u, ConvReluML) else u(xs)
        prev = u
    return xs
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
        self.transform_method, self.moment_mul = h.TRANSFORM_METHOD, float(h.MOMENT_MUL)     # pointset_head.py:26-31
        if self.transform_method not in ("minmax", "partial_minmax", "moment"):
            raise ValueError(f"META_ARCH.TRANSFORM_METHOD {self.transform_method!r}")
        self.box_points = 4 if self.transform_method == "partial_minmax" else self.num_points
        self.moment_transfer = nn.Parameter(torch.zeros(2)) if self.transform_method == "moment" else None
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
        cls_f, loc_f = _run_tower(self.cls_subnet, cls_f), _run_tower(self.loc_subnet, loc_f)
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
    prefetch, _take_prefetched = FCOSV2.prefetch, FCOSV2._take_prefetched
    postprocess = FCOSV2.postprocess

    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        features = [features[f] for f in self.head.in_features]
        if not self.training:
            return self.postprocess(self.head(images, features), batched_inputs, images.image_sizes)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        return self.head(images, features, gt_instances)


# ------------------------------------------------------------------------------------------------ LRTBHead
class _ScaleMulFn(torch.autograd.Function):
    """``Scale`` (slender_det/layers/scale.py:5-11) of one FPN level with the five scalars stored as one arena parameter."""

    @staticmethod
    def forward(ctx, x, scales, level, owner):
        ctx.level, ctx.owner = level, owner
        ctx.save_for_backward(x, scales)
        arena = getattr(owner, "_arena", None)
        if arena is not None:
            arena.note_use(scales)
        return x * scales[level]

    @staticmethod
    def backward(ctx, dy):
        x, scales = ctx.saved_tensors
        arena = getattr(ctx.owner, "_arena", None)
        g = (dy * x).sum()
        if arena is not None:
            arena.grad_view(scales)[ctx.level] += g
            arena.mark_ready(scales)
            return dy * scales[ctx.level], None, None, None
        ds = torch.zeros_like(scales)
        ds[ctx.level] = g
        return dy * scales[ctx.level], ds, None, None


class _LrtbLossFn(torch.autograd.Function):
    """LRTBHead.losses (meta/heads/lrtb_head.py:190-258): focal + two centerness-weighted IoU losses + centerness BCE."""

    @staticmethod
    def forward(ctx, head, cls, ctr, init, refine, labels, reg_t, ctr_t, stats, inv_world, init_labels=None, init_ctr=None):
        """``init_labels`` / ``init_ctr`` (LRTBTopkHead): the init-box loss runs over another row selection with its own weights;
        ``stats`` then carries a third entry, the sum of those weights."""
        K = head.num_classes
        lab, rt, ct = labels.view(-1), reg_t.view(-1, 4), ctr_t.view(-1)
        lab_i = lab if init_labels is None else init_labels.view(-1)
        ct_i = ct if init_ctr is None else init_ctr.view(-1)
        cls2, init2, ref2, ctr2 = cls.reshape(-1, K).contiguous(), init.reshape(-1, 4).contiguous(), refine.reshape(-1, 4).contiguous(), ctr.reshape(-1).contiguous()
        focal, _ = HF.focal_loss_fwd(cls2, lab, None, head.focal_loss_alpha, head.focal_loss_gamma)
        s_init, _ = HF.iou_loss_fwd(init2, rt, ct_i, head.iou_loss_type, mask=lab_i, mask_bg=K)
        s_ref, _ = HF.iou_loss_fwd(ref2, rt, ct, head.iou_loss_type, mask=lab, mask_bg=K)
        s_ctr = HF.bce_logits_soft_fwd(ctr2, ct, lab, K)
        npos = torch.clamp(stats[0:1] * inv_world, min=1.0)
        sctr = torch.where(stats[0:1] > 0, stats[1:2] * inv_world, torch.ones_like(stats[1:2]))   # no positives: the sums are 0 anyway
        sctr_i = sctr if init_labels is None else torch.where(stats[0:1] > 0, stats[2:3] * inv_world, torch.ones_like(stats[2:3]))
        ctx.head, ctx.inv_world, ctx.shapes = head, inv_world, (cls.shape, ctr.shape, init.shape, refine.shape)
        ctx.save_for_backward(cls2, ctr2, init2, ref2, lab, rt, ct, stats, npos, sctr, lab_i, ct_i, sctr_i)
        return torch.cat([focal / npos, s_init / sctr_i, s_ref / sctr, s_ctr / npos])

    @staticmethod
    def backward(ctx, g4):
        head = ctx.head
        cls2, ctr2, init2, ref2, lab, rt, ct, stats, npos, sctr, lab_i, ct_i, sctr_i = ctx.saved_tensors
        K = head.num_classes
        g4 = g4.contiguous().float()
        dcls = HF.focal_loss_bwd(cls2, lab, None, head.focal_loss_alpha, head.focal_loss_gamma, scale_num=g4[0:1], scale_den=stats[0:1],
                                 den_mul=ctx.inv_world, den_min=1.0)
        dinit = HF.iou_loss_bwd(init2, rt, ct_i, head.iou_loss_type, mask=lab_i, mask_bg=K, grad_scale=(g4[1:2] / sctr_i).contiguous())
        dref = HF.iou_loss_bwd(ref2, rt, ct, head.iou_loss_type, mask=lab, mask_bg=K, grad_scale=(g4[2:3] / sctr).contiguous())
        dctr = HF.bce_logits_soft_bwd(ctr2, ct, lab, K, (g4[3:4] / npos).contiguous())
        s = ctx.shapes
        return None, dcls.view(s[0]), dctr.view(s[1]), dinit.view(s[2]), dref.view(s[3]), None, None, None, None, None, None, None


@MEAT_HEADS_REGISTRY.register()
class LRTBHead(nn.Module):
    """slender_det/modeling/meta_arch/meta/heads/lrtb_head.py:24-375: FCOS-style left/right/top/bottom distances predicted twice
    (init, then refined on adapted features), FCOS targets (fcos/utils.py), centerness-weighted IoU losses, centerness BCE.
    Quirks reproduced: ``lrtb_to_points`` (heads/utils.py:20-23) reads the channels as (l, r, t, b) although the targets are
    (l, t, r, b); "Supervised Offset" only controls taps 0 and 8 of the 3x3 kernel, the other seven come from ``offset_conv_extend``."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        h = cfg.MODEL.META_ARCH
        self.in_channels = input_shape[0].channels
        self.in_features, self.fpn_strides = list(h.IN_FEATURES), list(h.FPN_STRIDES)
        self.num_classes, self.feat_channels, self.stacked_convs, self.norm = h.NUM_CLASSES, h.FEAT_CHANNELS, h.STACK_CONVS, h.NORM
        self.feat_adaption, self.res_refine = h.FEAT_ADAPTION, h.RES_REFINE
        self.gradient_mul, self.prior_prob = h.GRADIENT_MUL, h.PRIOR_PROB
        self.focal_loss_gamma, self.focal_loss_alpha = h.FOCAL_LOSS_GAMMA, h.FOCAL_LOSS_ALPHA
        self.loss_cls_weight, self.loss_loc_init_weight, self.loss_loc_refine_weight = h.LOSS_CLS_WEIGHT, h.LOSS_LOC_INIT_WEIGHT, h.LOSS_LOC_REFINE_WEIGHT
        self.score_threshold, self.topk_candidates, self.nms_threshold = h.SCORE_THRESH_TEST, h.TOPK_CANDIDATES_TEST, h.NMS_THRESH_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        assert h.NUM_POINTS == 2, "LRTBHead: NUM_POINTS must be 2 (lrtb_head.py:32)"
        self.center_sampling_radius, self.norm_reg_targets = h.CENTER_SAMPLING_RADIUS, h.NORM_REG_TARGETS
        self.centerness_on_loc, self.iou_loss_type = h.CENTERNESS_ON_LOC, h.IOU_LOSS_TYPE
        self.slender_centerness = h.SLENDER_CENTERNESS
        if self.feat_adaption not in FEAT_ADAPTION_METHODS:
            raise AssertionError(f"{self.feat_adaption} {type(self.feat_adaption)}")
        if self.norm not in ("GN", ""):
            raise NotImplementedError(f"META_ARCH.NORM {self.norm!r}: only 'GN' and '' are built")
        C = self.feat_channels
        assert self.in_channels == C == h.LOC_FEAT_CHANNELS == 256, "LRTBHead is built for 256-channel features"
        unit = ConvGnRelu if self.norm == "GN" else ConvReluML
        self.cls_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_init_conv = ConvML(C, C, 3, 1, relu=True)
        self.loc_init_out = ConvML(C, 8, 1, 0, out_f32=True)                  # 4 distances (+4 pad)
        if self.feat_adaption == "Empty":
            self.cls_conv = ConvML(C, C, 3, 1, relu=True)
            self.loc_refine_conv = ConvML(C, C, 3, 1, relu=True)
        else:
            self.cls_conv = DeformConv(C, C, 3, 1, 1, relu=True)
            self.loc_refine_conv = DeformConv(C, C, 3, 1, 1, relu=True)
        if self.feat_adaption == "Unsupervised Offset":
            self.offset_conv = ConvML(C, 24, 1, 0, out_f32=True)
        elif self.feat_adaption == "Split Unsup Offset":
            self.offset_conv_cls = ConvML(C, 24, 1, 0, out_f32=True)
            self.offset_conv_loc = ConvML(C, 24, 1, 0, out_f32=True)
        elif self.feat_adaption == "Supervised Offset":
            self.offset_conv_extend = ConvML(C, 16, 1, 0, out_f32=True)      # 14 offsets (+2 pad)
        K = self.num_classes
        self.kc = K + (0 if self.centerness_on_loc else 1)                   # cls_out (+ ctn_out) fused, like FCOSHead.cls_pred
        self.cls_pred = ConvML(C, (self.kc + 7) // 8 * 8, 1, 0, out_f32=True)
        self.box_pred = ConvML(C, 8, 1, 0, out_f32=True)                      # loc_refine_out (4) (+ ctn_out when CENTERNESS_ON_LOC)
        self.scales_init = nn.Parameter(torch.ones(len(self.fpn_strides)))
        self.scales_refine = nn.Parameter(torch.ones(len(self.fpn_strides)))
        with torch.no_grad():
            for u in list(self.cls_subnet) + list(self.loc_subnet):
                u.conv.init_normal(0.01, 0.0)
            named = [self.loc_init_conv, self.loc_init_out, self.cls_pred, self.box_pred]
            named += [m for m in (self.cls_conv, self.loc_refine_conv) if isinstance(m, ConvML)]
            named += [getattr(self, n) for n in ("offset_conv", "offset_conv_cls", "offset_conv_loc", "offset_conv_extend") if hasattr(self, n)]
            for m in named:
                m.conv.init_normal(0.01, 0.0)
            self.loc_init_out.conv.weight[4:].zero_()
            self.cls_pred.conv.weight[self.kc:].zero_()
            self.box_pred.conv.weight[5 if self.centerness_on_loc else 4:].zero_()
            self.cls_pred.conv.bias[:K].fill_(-math.log((1 - self.prior_prob) / self.prior_prob))
            for n, rows in (("offset_conv", 18), ("offset_conv_cls", 18), ("offset_conv_loc", 18), ("offset_conv_extend", 14)):
                if hasattr(self, n):
                    getattr(self, n).conv.weight[rows:].zero_()
        self.register_buffer("_dev_probe", torch.zeros(1))
        self.last_targets = None

    @property
    def device(self):
        return self._dev_probe.device

    def _decode(self, raw, scales, level):
        z = _ScaleMulFn.apply(raw, scales, level, self)
        return torch.relu(z) * self.fpn_strides[level] if self.norm_reg_targets else torch.exp(z)

    def run_head(self, features):
        """-> per level: cls logits (N,H,W,K), centerness logits (N,H,W), init / refine distances (N,H,W,4), all fp32."""
        nl, K = len(features), self.num_classes
        cls_f, loc_f = list(features), list(features)
        cls_f, loc_f = _run_tower(self.cls_subnet, cls_f), _run_tower(self.loc_subnet, loc_f)
        raw_init = self.loc_init_out(self.loc_init_conv(loc_f))
        init = [self._decode(raw_init[l][..., :4], self.scales_init, l) for l in range(nl)]
        if self.feat_adaption == "Empty":
            cf, lf = self.cls_conv(cls_f), self.loc_refine_conv(loc_f)
        else:
            if self.feat_adaption == "Unsupervised Offset":
                off_c = off_l = self.offset_conv(loc_f)
            elif self.feat_adaption == "Split Unsup Offset":
                off_c, off_l = self.offset_conv_cls(loc_f), self.offset_conv_loc(loc_f)
            else:
                ext = self.offset_conv_extend(loc_f)
                off_c = []
                for l in range(nl):
                    gm = (1 - self.gradient_mul) * init[l].detach() + self.gradient_mul * init[l]
                    # lrtb_to_points reads (l, r, t, b): [-ch0, -ch2, ch1, ch3]; dcn_base_offset[[0, 1, -2, -1]] = [-1, -1, 1, 1]
                    d = torch.stack((-gm[..., 0], -gm[..., 2], gm[..., 1], gm[..., 3]), dim=-1) / self.fpn_strides[l]
                    d = d - d.new_tensor([-1.0, -1.0, 1.0, 1.0])
                    pad = d.new_zeros(d.shape[:-1] + (6,))
                    off_c.append(torch.cat((d[..., 0:2], ext[l][..., :14], d[..., 2:4], pad), dim=-1).contiguous())
                off_l = off_c
            cf = [self.cls_conv(cls_f[l], off_c[l], off_ld=24) for l in range(nl)]
            lf = [self.loc_refine_conv(loc_f[l], off_l[l], off_ld=24) for l in range(nl)]
        cp, bp = self.cls_pred(cf), self.box_pred(lf)
        cls = [cp[l][..., :K] for l in range(nl)]
        ctr = [bp[l][..., 4] if self.centerness_on_loc else cp[l][..., K] for l in range(nl)]
        refine = []
        for l in range(nl):
            r = self._decode(bp[l][..., :4], self.scales_refine, l)
            refine.append(r + init[l].detach() if self.res_refine else r)
        return cls, ctr, init, refine

    def forward(self, images, features, gt_instances=None):
        import torch.distributed as dist

        from ...utils import comm
        from .fcos import SIZES_OF_INTEREST

        N = features[0].shape[0]
        hw = [(f.shape[1], f.shape[2]) for f in features]
        cls, ctr, init, refine = self.run_head(features)
        K = self.num_classes
        cat = lambda ts, c: torch.cat([t.reshape(N, -1, c) if c > 1 else t.reshape(N, -1) for t in ts], dim=1)
        cls_all, ctr_all, init_all, ref_all = cat(cls, K), cat(ctr, 1), cat(init, 4), cat(refine, 4)
        if not self.training:
            with torch.no_grad():
                return self.inference(hw, cls_all, ctr_all, ref_all, images.image_sizes)
        dev = cls_all.device
        counts = [len(g) for g in gt_instances]
        offs = torch.tensor([0] + counts, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev, non_blocking=True)
        if sum(counts) > 0:
            boxes = torch.cat([g.gt_boxes.tensor for g in gt_instances]).float().contiguous()
            classes = torch.cat([g.gt_classes for g in gt_instances]).to(torch.int32).contiguous()
        else:
            boxes, classes = torch.zeros((1, 4), dtype=torch.float32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)
        with torch.no_grad():
            labels, reg_t, ctr_t, stats = HF.fcos_assign(boxes, classes, offs, N, hw, self.fpn_strides, SIZES_OF_INTEREST,
                                                         self.center_sampling_radius, K)
            ctr_std = ctr_t
            if self.slender_centerness:
                # compute_slender_centerness_targets (fcos/utils.py:302-312): centerness ** (0.5 * min(w/h, h/w)) on the positives;
                # the assignment kernel returns sqrt(centerness), so raise it to the ratio itself
                fg = (labels >= 0) & (labels != K)
                r = (reg_t[..., 0] + reg_t[..., 2]) / (reg_t[..., 1] + reg_t[..., 3])
                ratio = torch.minimum(r, 1.0 / r)
                ctr_t = torch.where(fg, torch.pow(ctr_t, ratio), torch.zeros_like(ctr_t)).contiguous()
                stats = torch.stack((stats[0], ctr_t.sum()))
            init_labels, init_ctr, stats = self.init_selection(hw, labels, reg_t, ctr_std, stats)
            world = comm.get_world_size()
            if world > 1:
                dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        self.last_targets = (labels, reg_t, ctr_t, stats)
        out = _LrtbLossFn.apply(self, cls_all, ctr_all, init_all, ref_all, labels, reg_t, ctr_t, stats, 1.0 / float(world), init_labels, init_ctr)
        return {"loss_cls": out[0] * self.loss_cls_weight, "centerness_loss": out[3] * self.loss_cls_weight,
                "loss_loc_init": out[1] * self.loss_loc_init_weight, "loss_loc_refine": out[2] * self.loss_loc_refine_weight}

    def init_selection(self, hw, labels, reg_t, ctr_std, stats):
        """Rows and weights of the init-box loss: LRTBHead uses the foreground rows and the (possibly slender) centerness."""
        return None, None, stats

    @torch.no_grad()
    def inference(self, hw, cls_all, ctr_all, ref_all, image_sizes):
        """lrtb_head.py:283-375: per level score threshold on the class probability, times centerness, top-k, decode, sqrt, NMS."""
        from ...layers.nms import batched_nms

        dev = cls_all.device
        bounds = [0]
        for h, w in hw:
            bounds.append(bounds[-1] + h * w)
        locs = []
        for (h, w), s in zip(hw, self.fpn_strides):
            ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=dev)
            xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=dev)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            locs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + s // 2)
        results = []
        for i, image_size in enumerate(image_sizes):
            B, S, C = [], [], []
            for l in range(len(hw)):
                sl = slice(bounds[l], bounds[l + 1])
                p = cls_all[i, sl].sigmoid()
                keep = p > self.score_threshold
                p = p * ctr_all[i, sl].sigmoid()[:, None]
                sc = p[keep]
                idx = keep.nonzero()
                loc_i, class_i = idx[:, 0], idx[:, 1]
                reg_i, locs_i = ref_all[i, sl][loc_i], locs[l][loc_i]
                n_keep = int(keep.sum())
                top_n = min(n_keep, self.topk_candidates)
                if n_keep > top_n:
                    sc, ti = sc.topk(top_n, sorted=False)
                    class_i, reg_i, locs_i = class_i[ti], reg_i[ti], locs_i[ti]
                B.append(torch.stack([locs_i[:, 0] - reg_i[:, 0], locs_i[:, 1] - reg_i[:, 1], locs_i[:, 0] + reg_i[:, 2], locs_i[:, 1] + reg_i[:, 3]], dim=1))
                S.append(torch.sqrt(sc))
                C.append(class_i)
            B, S, C = torch.cat(B), torch.cat(S), torch.cat(C)
            keep = batched_nms(B, S, C, self.nms_threshold)[: self.max_detections_per_image]
            r = Instances(tuple(image_size))
            r.pred_boxes, r.scores, r.pred_classes = Boxes(B[keep]), S[keep], C[keep]
            results.append(r)
        return results


@MEAT_HEADS_REGISTRY.register()
class LRTBTopkHead(LRTBHead):
    """slender_det/modeling/meta_arch/meta/heads/lrtb_topk_head.py:23-368: LRTBHead whose INIT boxes are only supervised at the top-k
    (5) positive locations of every gt box, ranked by the standard centerness target (fcos/utils.py:215-292,
    ``compute_topk_targets_for_locations``), with those centerness values as weights and their sum as the normaliser.  Inference
    thresholds on PRE_NMS_THRESH / PRE_NMS_TOP_N (:322,:336) instead of SCORE_THRESH_TEST / TOPK_CANDIDATES_TEST."""
    topk_per_box = 5

    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        h = cfg.MODEL.META_ARCH
        self.score_threshold, self.topk_candidates = h.PRE_NMS_THRESH, h.PRE_NMS_TOP_N
        self.last_topk = None

    def init_selection(self, hw, labels, reg_t, ctr_std, stats):
        K, dev = self.num_classes, labels.device
        N, L = labels.shape
        locs = []
        for (h, w), s in zip(hw, self.fpn_strides):
            ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=dev)
            xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=dev)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            locs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + s // 2)
        locs = torch.cat(locs)
        topk = torch.zeros((N, L), dtype=torch.bool, device=dev)
        for i in range(N):
            fg = ((labels[i] >= 0) & (labels[i] != K)).nonzero().squeeze(1)
            if fg.numel() == 0:
                continue
            r = reg_t[i, fg]
            # the gt box a positive location regresses to identifies its gt (locations_to_gt_inds of the reference)
            box = torch.stack((locs[fg, 0] - r[:, 0], locs[fg, 1] - r[:, 1], locs[fg, 0] + r[:, 2], locs[fg, 1] + r[:, 3]), dim=1)
            uniq, inv = torch.unique((box * 8).round().to(torch.int64), dim=0, return_inverse=True)
            score = ctr_std[i, fg]
            for g in range(uniq.shape[0]):
                rows = (inv == g).nonzero().squeeze(1)
                if rows.numel() > self.topk_per_box:
                    rows = rows[torch.topk(score[rows], self.topk_per_box, sorted=False)[1]]
                topk[i, fg[rows]] = True
        self.last_topk = topk
        init_labels = torch.where(topk, labels, torch.full_like(labels, K)).contiguous()
        stats = torch.cat((stats, (ctr_std * topk).sum().reshape(1)))
        return init_labels, ctr_std.contiguous(), stats


# ------------------------------------------------------------------------------------------------ AnchorHead
class _InitBoxLossFn(torch.autograd.Function):
    """smooth_l1(pred[fg] / (4 stride), gt[fg] / (4 stride), 0.11, "sum") / max(#fg, 1)   (meta/heads/anchor_head.py:353-361)."""

    @staticmethod
    def forward(ctx, pred, target, obj, strides):
        pred = pred.contiguous()
        sums = HF.reppoints_box_loss_fwd(pred, target, obj, strides, -1, 0.11)
        ctx.save_for_backward(pred, target, obj, strides, sums)
        return sums[0:1] / torch.clamp(sums[1:2], min=1.0)

    @staticmethod
    def backward(ctx, g):
        pred, target, obj, strides, sums = ctx.saved_tensors
        return HF.reppoints_box_loss_bwd(pred, target, obj, strides, -1, 0.11, g.contiguous().float(), sums[1:2], 1.0, 1.0), None, None, None


@MEAT_HEADS_REGISTRY.register()
class AnchorHead(nn.Module):
    """slender_det/modeling/meta_arch/meta/heads/anchor_head.py:25-527: a RetinaNet head (A anchors per location, 3x3 ``cls_out`` /
    ``loc_refine_out``) on the unified GN towers with a feature-adaption layer in {none, unsupervised, split, supervised}, plus an
    auxiliary "init box" (two corner points per location, scaled by 1/2/4/8/16, supervised at the nearest point of every gt box,
    ``nearest_point_match``).  Anchor labelling, focal / smooth-L1 / GIoU losses and the EMA normaliser are RetinaNet's
    (retinanet.py: ``_RetinaLossFn``); the init loss is the stride-normalised smooth-L1 kernel of the RepPoints path.
    ``RES_REFINE`` must be False, as in the reference's configs (its residual add mixes 4- and 4A-channel tensors)."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        from .retinanet import RetinaNetHead

        h = cfg.MODEL.META_ARCH
        self.in_channels = input_shape[0].channels
        self.in_features, self.fpn_strides = list(h.IN_FEATURES), list(h.FPN_STRIDES)
        self.strides = [s.stride for s in input_shape]
        self.num_classes, self.feat_channels, self.stacked_convs, self.norm = h.NUM_CLASSES, h.FEAT_CHANNELS, h.STACK_CONVS, h.NORM
        self.feat_adaption, self.res_refine = h.FEAT_ADAPTION, h.RES_REFINE
        self.gradient_mul, self.prior_prob = h.GRADIENT_MUL, h.PRIOR_PROB
        self.focal_loss_gamma, self.focal_loss_alpha = h.FOCAL_LOSS_GAMMA, h.FOCAL_LOSS_ALPHA
        self.loss_cls_weight, self.loss_loc_init_weight, self.loss_loc_refine_weight = h.LOSS_CLS_WEIGHT, h.LOSS_LOC_INIT_WEIGHT, h.LOSS_LOC_REFINE_WEIGHT
        self.score_threshold, self.topk_candidates, self.nms_threshold = h.SCORE_THRESH_TEST, h.TOPK_CANDIDATES_TEST, h.NMS_THRESH_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.box_reg_loss_type = h.BBOX_REG_LOSS_TYPE
        if self.box_reg_loss_type not in ("smooth_l1", "giou"):
            raise ValueError(f"Invalid bbox reg loss type '{self.box_reg_loss_type}'")
        if self.feat_adaption not in (None, "none", "unsupervised", "split", "supervised"):
            raise AssertionError(self.feat_adaption)                    # anchor_head.py:104
        if self.res_refine:
            raise NotImplementedError("AnchorHead with RES_REFINE adds a 4-channel tensor to 4A channels in the reference; not built")
        if self.norm not in ("GN", ""):
            raise NotImplementedError(f"META_ARCH.NORM {self.norm!r}: only 'GN' and '' are built")
        ag = cfg.MODEL.ANCHOR_GENERATOR
        self.anchor_sizes, self.anchor_ratios, self.anchor_offset = [list(s) for s in ag.SIZES], [list(a) for a in ag.ASPECT_RATIOS], ag.OFFSET
        self.num_anchors = len(self.anchor_sizes[0]) * len(self.anchor_ratios[0])
        self.bbox_reg_weights = tuple(h.BBOX_REG_WEIGHTS)
        self.iou_thresholds, self.iou_labels = list(h.IOU_THRESHOLDS), list(h.IOU_LABELS)
        self.smooth_l1_loss_beta = 0.11
        self.scale_clamp = math.log(1000.0 / 16)
        C = self.feat_channels
        assert self.in_channels == C == h.LOC_FEAT_CHANNELS == 256, "AnchorHead is built for 256-channel features"
        unit = ConvGnRelu if self.norm == "GN" else ConvReluML
        self.cls_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_subnet = nn.ModuleList([unit(C) for _ in range(self.stacked_convs)])
        self.loc_init_conv = ConvML(C, C, 3, 1, relu=True)
        self.loc_init_out = ConvML(C, 8, 3, 1, out_f32=True)                 # 3x3 -> (x1, y1, x2, y2) offsets (+4 pad)
        fa = self.feat_adaption
        if fa in (None, "none"):
            self.cls_conv, self.loc_refine_conv = ConvML(C, C, 3, 1, relu=True), ConvML(C, C, 3, 1, relu=True)
        else:
            self.cls_conv, self.loc_refine_conv = DeformConv(C, C, 3, 1, 1, relu=True), DeformConv(C, C, 3, 1, 1, relu=True)
            if fa == "unsupervised":
                self.offset_conv = ConvML(C, 24, 1, 0, out_f32=True)
            elif fa == "split":
                self.offset_conv_cls, self.offset_conv_loc = ConvML(C, 24, 1, 0, out_f32=True), ConvML(C, 24, 1, 0, out_f32=True)
            else:
                self.offset_conv = ConvML(C, 16, 1, 0, out_f32=True)         # 14 offsets (+2 pad)
        self.kc = self.num_anchors * self.num_classes
        assert self.kc % 8 == 0
        self.box_pitch = (self.num_anchors * 4 + 7) // 8 * 8
        self.cls_score = HipConv2d(C, self.kc, 3, 1, 1, bias=True)           # cls_out
        self.bbox_pred = HipConv2d(C, self.box_pitch, 3, 1, 1, bias=True)    # loc_refine_out
        with torch.no_grad():         # anchor_head.py:123-137 (offset convs and DeformConv keep their default init there)
            for u in list(self.cls_subnet) + list(self.loc_subnet):
                u.conv.init_normal(0.01, 0.0)
            mods = [self.loc_init_conv.conv, self.loc_init_out.conv, self.cls_score, self.bbox_pred]
            mods += [m.conv for m in (self.cls_conv, self.loc_refine_conv) if isinstance(m, ConvML)]
            for m in mods:
                m.init_normal(0.01, 0.0)
            self.loc_init_out.conv.weight[4:].zero_()
            self.bbox_pred.weight[self.num_anchors * 4:].zero_()
            self.cls_score.bias.fill_(-math.log((1 - self.prior_prob) / self.prior_prob))
            for n, rows in (("offset_conv", 14 if fa == "supervised" else 18), ("offset_conv_cls", 18), ("offset_conv_loc", 18)):
                if hasattr(self, n):
                    conv = getattr(self, n).conv
                    bound = 1.0 / math.sqrt(C)                               # nn.Conv2d default init (kaiming_uniform_(a=sqrt(5)))
                    conv.weight.uniform_(-bound, bound)
                    conv.bias.uniform_(-bound, bound)
                    conv.weight[rows:].zero_()
                    conv.bias[rows:].zero_()
        self.register_buffer("loss_normalizer", torch.tensor([100.0]))      # anchor_head.py:62-63
        self.loss_normalizer_momentum = 0.9
        self._anchor_cache, self._grid_cache = {}, {}
        self._predict = RetinaNetHead.predict
        self.last_targets = None

    # attribute names _RetinaLossFn reads from ``model`` and ``model.head``
    @property
    def head(self):
        return self

    @property
    def device(self):
        return self.loss_normalizer.device

    def predict(self, cls_t, box_t):
        return self._predict(self, cls_t, box_t)

    point_grid = RepPointsDetector.point_grid

    def anchors_for(self, level_hw):
        from ..anchor_generator import grid_anchors

        key = tuple(level_hw)
        if key not in self._anchor_cache:
            per_level = grid_anchors(level_hw, self.strides, self.anchor_sizes, self.anchor_ratios, self.anchor_offset, self.device)
            self._anchor_cache[key] = torch.cat(per_level).contiguous()
        return self._anchor_cache[key]

    def run_head(self, features):
        """-> cls towers, box towers (inputs of the 3x3 prediction convs), raw init offsets per level (N,H,W,8)."""
        nl = len(features)
        cls_f, loc_f = list(features), list(features)
        cls_f, loc_f = _run_tower(self.cls_subnet, cls_f), _run_tower(self.loc_subnet, loc_f)
        raw = self.loc_init_out(self.loc_init_conv(loc_f))
        fa = self.feat_adaption
        if fa in (None, "none"):
            return self.cls_conv(cls_f), self.loc_refine_conv(loc_f), raw
        if fa == "unsupervised":
            off_c = off_l = self.offset_conv(loc_f)
        elif fa == "split":
            off_c, off_l = self.offset_conv_cls(loc_f), self.offset_conv_loc(loc_f)
        else:       # supervised (anchor_head.py:192-209): [flip_xy(grad_mul(init)), offset_conv(loc_feat)] - dcn_base_offset
            ext = self.offset_conv(loc_f)
            base = torch.tensor([[i, j] for i in (-1.0, 0.0, 1.0) for j in (-1.0, 0.0, 1.0)], device=raw[0].device).reshape(-1)
            off_c = []
            for l in range(nl):
                r4 = raw[l][..., :4]
                gm = (1 - self.gradient_mul) * r4.detach() + self.gradient_mul * r4
                flipped = torch.stack((gm[..., 1], gm[..., 0], gm[..., 3], gm[..., 2]), dim=-1)
                off = torch.cat((flipped, ext[l][..., :14]), dim=-1) - base
                off_c.append(torch.cat((off, off.new_zeros(off.shape[:-1] + (6,))), dim=-1).contiguous())
            off_l = off_c
        cf = [self.cls_conv(cls_f[l], off_c[l], off_ld=24) for l in range(nl)]
        lf = [self.loc_refine_conv(loc_f[l], off_l[l], off_ld=24) for l in range(nl)]
        return cf, lf, raw

    def init_boxes(self, raw, hw):
        """loc_out_init * factor + (cx, cy, cx, cy)  (anchor_head.py:213-219) concatenated to (N, X, 4)."""
        centers, _, _ = self.point_grid(hw)
        N = raw[0].shape[0]
        out, o = [], 0
        for l, (h, w) in enumerate(hw):
            c = centers[o:o + h * w]
            out.append(raw[l][..., :4].reshape(N, h * w, 4) * float(2 ** l) + torch.cat((c, c), dim=1))
            o += h * w
        return torch.cat(out, dim=1)

    @torch.no_grad()
    def label_anchors(self, anchors, gt_instances):
        """anchor_head.py:394-434 (RetinaNet.label_anchors); for "giou" the second output holds the matched gt boxes."""
        N, R = len(gt_instances), anchors.shape[0]
        labels = torch.empty((N, R), dtype=torch.int32, device=anchors.device)
        deltas = torch.empty((N, R, 4), dtype=torch.float32, device=anchors.device)
        for i, g in enumerate(gt_instances):
            boxes = g.gt_boxes.tensor.float().contiguous()
            classes = g.gt_classes.to(torch.int32).contiguous()
            _, matches, mlab = HF.anchor_match(boxes, anchors, self.iou_thresholds, self.iou_labels, True)
            HF.retina_targets(anchors, boxes, classes, matches, mlab, self.num_classes, self.bbox_reg_weights, labels[i], deltas[i])
            if self.box_reg_loss_type == "giou":
                deltas[i] = boxes[matches.long()] if len(boxes) else 0.0
        return labels, deltas

    @torch.no_grad()
    def init_targets(self, hw, gt_instances, image_sizes):
        """get_ground_truth (anchor_head.py:241-283): nearest_point_match per image, off-image centres switched off."""
        centers, strides, lvl_start = self.point_grid(hw)
        counts = [len(g) for g in gt_instances]
        if min(counts) == 0:
            raise ValueError("No gt or bboxes")
        dev = centers.device
        box_off = torch.tensor([0] + counts, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev, non_blocking=True)
        boxes = torch.cat([g.gt_boxes.tensor for g in gt_instances]).float().contiguous()
        obj, lab = HF.reppoints_point_match(centers, strides, lvl_start, boxes, box_off, len(counts), max(counts), "nearest_points", 4.0)
        hwt = torch.tensor([[float(h), float(w)] for h, w in image_sizes], dtype=torch.float32).to(dev, non_blocking=True)
        invalid = (centers[None, :, 0] >= hwt[:, 1:2]) | (centers[None, :, 1] >= hwt[:, 0:1])
        return obj.masked_fill(invalid, 0).contiguous(), lab, strides

    def forward(self, images, features, gt_instances=None):
        from .retinanet import _RetinaLossFn

        hw = [(f.shape[1], f.shape[2]) for f in features]
        cls_t, box_t, raw = self.run_head(features)
        if not self.training:
            with torch.no_grad():
                cls_buf, box_buf, _, offs = self.predict(cls_t, box_t)
                return self.inference(hw, cls_buf, box_buf, offs, images.image_sizes)
        anchors = self.anchors_for(hw)
        gt_labels, gt_deltas = self.label_anchors(anchors, gt_instances)
        obj, init_lab, strides = self.init_targets(hw, gt_instances, images.image_sizes)
        self.last_targets = (gt_labels, gt_deltas, obj, init_lab)
        out = _RetinaLossFn.apply(self, self.cls_score.weight, gt_labels, gt_deltas, *cls_t, *box_t)
        init = _InitBoxLossFn.apply(self.init_boxes(raw, hw), init_lab, obj, strides)
        return {"loss_cls": out[0] * self.loss_cls_weight, "loss_loc_init": init[0] * self.loss_loc_init_weight,
                "loss_loc_refine": out[1] * self.loss_loc_refine_weight}

    @torch.no_grad()
    def inference(self, level_hw, cls_buf, box_buf, offs, image_sizes):
        from .retinanet import RetinaNet

        return RetinaNet.inference(self, level_hw, cls_buf, box_buf, offs, image_sizes)
