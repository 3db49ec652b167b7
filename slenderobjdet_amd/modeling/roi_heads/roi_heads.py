"""StandardROIHeads / RROIHeads on the HIP kernels (BASELINE config 5: configs/rotated/Base-RRCNN-FPN.yaml selects ``RROIHeads`` with a
``FastRCNNConvFCHead`` (2 FC) box head over ``ROIAlignRotated``; the reference's subclass roi_heads/roi_heads.py:27-66 rebuilds the same
pooler/head/predictor triple).  detectron2's sources are absent; semantics restated from SURVEY.md §2.3 / C.5-C.6, C.13:

  label_and_sample_proposals: append gt boxes, IoU + Matcher([0.5], [0, 1]), 512 samples per image at <= 25 % foreground;
  ROIPooler: level = floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to the pyramid, ROIAlign(aligned) / ROIAlignRotated 7x7;
  FastRCNNConvFCHead: flatten -> FC 1024 -> ReLU -> FC 1024 -> ReLU;  FastRCNNOutputLayers: cls (K+1), class-specific deltas (K*box_dim);
  losses: ``loss_cls`` = mean cross-entropy, ``loss_box_reg`` = smooth-L1(sum, beta 0) over foreground rows / number of sampled rows.

MI355X-first: the FC layers are the implicit-GEMM MFMA kernel on (R, 1, 1, C) "images" (fused bias + ReLU); pooled features are NHWC,
so the flatten order is (h, w, c) and the FC1 weight is stored in that order; both losses are one kernel each over all rows.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers.nms import batched_nms, batched_nms_rotated
from ...layers.nn import HipConv2d
from ...structures import Boxes, Instances, RotatedBoxes
from ...utils.registry import Registry
from ...layers import nn as _nn
from ..box_regression import Box2BoxTransform, Box2BoxTransformRotated

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


def _ceil8(v):
    return (v + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ pooler
class _RoiPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooler, rois, order, counts, *feats):
        PH, PW = pooler.output_size
        M, C = rois.shape[0], feats[0].shape[3]
        out = torch.zeros((M, PH, PW, C), dtype=torch.float32, device=rois.device)
        start = 0
        for l, cnt in enumerate(counts):
            if cnt:
                idx = order[start:start + cnt]
                out[idx] = HF.roi_align_fwd(feats[l], rois[idx].contiguous(), (PH, PW), pooler.scales[l], pooler.sampling_ratio, pooler.rotated)
            start += cnt
        ctx.pooler, ctx.counts, ctx.shapes = pooler, counts, [tuple(f.shape) for f in feats]
        ctx.park, ctx.ptrs = _nn.GradPark.current, [f.data_ptr() for f in feats]
        ctx.save_for_backward(rois, order)
        return HF.f32_to_bf16(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        pooler = ctx.pooler
        rois, order = ctx.saved_tensors
        dout = dout.float()
        grads, start = [], 0
        for l, cnt in enumerate(ctx.counts):
            g = None
            if ctx.needs_input_grad[4 + l]:
                if cnt:
                    idx = order[start:start + cnt]
                    g = HF.f32_to_bf16(HF.roi_align_bwd(dout[idx].contiguous(), rois[idx].contiguous(), ctx.shapes[l], pooler.scales[l],
                                                        pooler.sampling_ratio, pooler.rotated))
                else:
                    g = torch.zeros(ctx.shapes[l], dtype=HF.ACT_DTYPE, device=dout.device)
                park = ctx.park
                if park is not None and not park.done and ctx.ptrs[l] in park.consumer_ptrs:
                    # the RPN head's data gradient of this level adds it in its epilogue (layers/nn.py GradPark); a level without ROIs parks nothing
                    if cnt:
                        park.put(ctx.ptrs[l], g)
                    g = None
            grads.append(g)
            start += cnt
        return (None, None, None, None, *grads)


class ROIPooler(nn.Module):
    """detectron2.modeling.poolers.ROIPooler for ROIAlignV2 (aligned=True) and ROIAlignRotated."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        if pooler_type not in ("ROIAlignV2", "ROIAlignRotated"):
            raise NotImplementedError(f"POOLER_TYPE {pooler_type}: only ROIAlignV2 and ROIAlignRotated are built")
        self.output_size, self.scales, self.sampling_ratio = tuple(output_size), list(scales), sampling_ratio
        self.rotated = pooler_type == "ROIAlignRotated"
        min_level, max_level = -math.log2(scales[0]), -math.log2(scales[-1])
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1, "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"
        self.canonical_box_size, self.canonical_level = canonical_box_size, canonical_level

    def assign_levels(self, boxes_cat_area):
        sizes = torch.sqrt(boxes_cat_area)
        lv = torch.floor(self.canonical_level + torch.log2(sizes / self.canonical_box_size + 1e-8))
        return (torch.clamp(lv, min=self.min_level, max=self.max_level) - self.min_level).to(torch.int64)

    def pooler_format(self, box_lists):
        """(M, 1 + box_dim): batch index first."""
        parts = []
        for i, b in enumerate(box_lists):
            t = b.tensor
            parts.append(torch.cat((torch.full((len(t), 1), float(i), dtype=t.dtype, device=t.device), t), dim=1))
        return torch.cat(parts, dim=0).float().contiguous()

    def forward(self, x, box_lists):
        rois = self.pooler_format(box_lists)
        if len(x) == 1:
            levels = torch.zeros(rois.shape[0], dtype=torch.int64, device=rois.device)
        else:
            levels = self.assign_levels(torch.cat([b.area() for b in box_lists]))
        order = torch.sort(levels, stable=True).indices
        counts = torch.bincount(levels, minlength=len(x)).tolist()      # one host sync for the per-level sizes
        return _RoiPoolFn.apply(self, rois, order, counts, *x)


# ------------------------------------------------------------------------------------------------ box head / predictor
@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    def __init__(self, cfg, in_channels, height, width):
        super().__init__()
        b = cfg.MODEL.ROI_BOX_HEAD
        if b.NUM_CONV != 0 or b.NORM != "":
            raise NotImplementedError("FastRCNNConvFCHead with conv layers / norm is not built (the configs on this path use NUM_FC only)")
        assert b.NUM_FC > 0
        dim = in_channels * height * width
        self.fcs = nn.ModuleList()
        for i in range(b.NUM_FC):
            fc = HipConv2d(dim, b.FC_DIM, 1, 1, 0, bias=True, relu=True)     # Linear + ReLU as a 1x1 conv over (R,1,1,dim)
            fc.init_xavier()
            fc.is_linear = True                  # checkpoint interchange (checkpoint.py): a reference file stores nn.Linear (out, in)
            if i == 0:
                fc.fc_input_chw = (in_channels, height, width)      # ... whose input is flattened CHW there, HWC here
            self.fcs.append(fc)
            dim = b.FC_DIM
        self.output_size = dim

    def forward(self, x):
        x = x.reshape(x.shape[0], 1, 1, -1)
        for fc in self.fcs:
            x = fc(x)
        return x


class _FastRcnnLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, scores, deltas, gt_classes, gt_deltas):
        R = scores.shape[0]
        s2, d2 = scores.view(R, -1), deltas.view(R, -1)
        K, D = pred.num_classes, pred.box_dim
        s_cls = HF.softmax_ce_fwd(s2, gt_classes, K + 1)
        s_box = HF.fastrcnn_box_loss_fwd(d2, gt_classes, gt_deltas, K, pred.smooth_l1_beta)
        ctx.pred = pred
        ctx.save_for_backward(s2, d2, gt_classes, gt_deltas)
        return torch.cat([s_cls, s_box]) / max(R, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g2):
        pred = ctx.pred
        s2, d2, gt_classes, gt_deltas = ctx.saved_tensors
        R = s2.shape[0]
        g2 = g2.contiguous().float()
        ds = HF.softmax_ce_bwd(s2, gt_classes, pred.num_classes + 1, g2[0:1], 1.0 / max(R, 1))
        dd = HF.fastrcnn_box_loss_bwd(d2, gt_classes, gt_deltas, pred.num_classes, pred.smooth_l1_beta, g2[1:2], 1.0 / max(R, 1))
        return None, ds.view(R, 1, 1, -1), dd.view(R, 1, 1, -1), None, None


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, cfg, input_size, box2box_transform):
        super().__init__()
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.box2box_transform = box2box_transform
        self.box_dim = box2box_transform.box_dim
        if cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG:
            raise NotImplementedError("CLS_AGNOSTIC_BBOX_REG is not built")
        self.smooth_l1_beta = cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA
        self.test_score_thresh, self.test_nms_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST, cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.cls_pad, self.box_out = _ceil8(self.num_classes + 1), _ceil8(self.num_classes * self.box_dim)
        self.cls_score = HipConv2d(input_size, self.cls_pad, 1, 1, 0, bias=True, out_f32=True)
        self.bbox_pred = HipConv2d(input_size, self.box_out, 1, 1, 0, bias=True, out_f32=True)
        self.cls_score.is_linear = self.bbox_pred.is_linear = True
        self.cls_score.ckpt_rows, self.bbox_pred.ckpt_rows = self.num_classes + 1, self.num_classes * self.box_dim     # without the pad
        with torch.no_grad():
            self.cls_score.init_normal(0.01, 0.0)
            self.bbox_pred.init_normal(0.001, 0.0)
            self.cls_score.weight[self.num_classes + 1:].zero_()
            self.bbox_pred.weight[self.num_classes * self.box_dim:].zero_()

    def forward(self, x):
        return self.cls_score(x), self.bbox_pred(x)          # (R,1,1,cls_pad) / (R,1,1,box_out) fp32

    def losses(self, predictions, proposals):
        scores, deltas = predictions
        gt_classes = torch.cat([p.gt_classes for p in proposals]).to(torch.int32).contiguous()
        boxes = torch.cat([p.proposal_boxes.tensor for p in proposals]).float().contiguous()
        gt_boxes = torch.cat([p.gt_boxes.tensor for p in proposals]).float().contiguous()
        with torch.no_grad():
            gt_deltas = self.box2box_transform.get_deltas(boxes, gt_boxes)
        out = _FastRcnnLossFn.apply(self, scores, deltas, gt_classes, gt_deltas)
        return {"loss_cls": out[0], "loss_box_reg": out[1]}

    @torch.no_grad()
    def inference(self, predictions, proposals):
        scores, deltas = predictions
        R = scores.shape[0]
        K, D = self.num_classes, self.box_dim
        probs = torch.softmax(scores.view(R, -1)[:, : K + 1], dim=-1)
        boxes = torch.cat([p.proposal_boxes.tensor for p in proposals]).float().contiguous()
        pred = self.box2box_transform.apply_deltas(deltas.view(R, -1)[:, : K * D].contiguous(), boxes)
        sizes = [len(p) for p in proposals]
        rotated = D == 5
        results = []
        for pr, bx, p in zip(probs.split(sizes), pred.split(sizes), proposals):
            results.append(fast_rcnn_inference_single_image(bx, pr, p.image_size, self.test_score_thresh, self.test_nms_thresh,
                                                            self.test_topk_per_image, rotated))
        return results


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image, rotated=False):
    """d2 fast_rcnn_inference_single_image(_rotated): per-class score threshold, class-aware NMS, top-k."""
    D = 5 if rotated else 4
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    K = scores.shape[1]
    BoxT = RotatedBoxes if rotated else Boxes
    b = BoxT(boxes.reshape(-1, D).clone())
    b.clip(image_shape)
    boxes = b.tensor.view(-1, K, D)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes, scores = boxes[mask], scores[mask]
    keep = (batched_nms_rotated if rotated else batched_nms)(boxes, scores, inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    res = Instances(tuple(image_shape))
    res.pred_boxes = BoxT(boxes[keep])
    res.scores = scores[keep]
    res.pred_classes = inds[keep, 1]
    return res


# ------------------------------------------------------------------------------------------------ ROI heads
@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(nn.Module):
    rotated = False

    def __init__(self, cfg, input_shape):
        super().__init__()
        r, b = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        if cfg.MODEL.MASK_ON or cfg.MODEL.KEYPOINT_ON:
            raise NotImplementedError("mask / keypoint heads are outside the hot path")
        self.num_classes = r.NUM_CLASSES
        self.batch_size_per_image, self.positive_fraction = r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION
        self.proposal_append_gt = r.PROPOSAL_APPEND_GT
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        assert len(self.iou_thresholds) == 1 and len(self.iou_labels) == 2, "ROI_HEADS uses a single IoU threshold"
        self.box_in_features = list(r.IN_FEATURES)
        in_channels = [input_shape[f].channels for f in self.box_in_features]
        assert len(set(in_channels)) == 1, in_channels
        pooler_type = b.POOLER_TYPE
        if self.rotated:
            assert pooler_type == "ROIAlignRotated", pooler_type
        self.box_pooler = ROIPooler(b.POOLER_RESOLUTION, tuple(1.0 / input_shape[k].stride for k in self.box_in_features),
                                    b.POOLER_SAMPLING_RATIO, pooler_type)
        self.box_head = ROI_BOX_HEAD_REGISTRY.get(b.NAME)(cfg, in_channels[0], b.POOLER_RESOLUTION, b.POOLER_RESOLUTION)
        transform = (Box2BoxTransformRotated if self.rotated else Box2BoxTransform)(weights=b.BBOX_REG_WEIGHTS)
        self.box_predictor = FastRCNNOutputLayers(cfg, self.box_head.output_size, transform)
        self.last_sampled = None

    def _box_type(self):
        return RotatedBoxes if self.rotated else Boxes

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets):
        """detectron2 ROIHeads.label_and_sample_proposals: append the ground truth, match, label, draw batch_size_per_image samples with
        positive_fraction foreground.  Matching runs per image (its size differs); the random draw of the WHOLE batch is one launch
        (sod_sample_labels + sod_compact_samples) and ONE host read (how many samples every image got) instead of two nonzero() and two
        randperm() per image."""
        BoxT = self._box_type()
        gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        if self.num_classes > 126:
            raise NotImplementedError("label sampling packs class indices into int8")
        # Everything below runs on padded (N, R) tensors of the whole batch: one concatenation of the proposals and of the ground truth,
        # one scatter each into the padded rows, ONE matching / labelling launch (sod_roi_label_batched), the draw, one gather per field -
        # ~25 launches per step where the per-image formulation (cat, match, index, mask, index ... for every image) took ~370.
        N = len(proposals)
        D = 5 if self.rotated else 4
        dev = proposals[0].proposal_boxes.tensor.device
        S = self.batch_size_per_image
        P = [len(p) for p in proposals]
        G = [len(t) for t in targets]
        g_off = [0]
        for g in G:
            g_off.append(g_off[-1] + g)
        Gtot = g_off[-1]
        gt_cat = torch.cat([t.gt_boxes.tensor.float() for t in targets]).contiguous() if Gtot else torch.zeros((0, D), dtype=torch.float32, device=dev)
        gtc_cat = torch.cat([t.gt_classes for t in targets]) if Gtot else torch.zeros((0,), dtype=torch.int64, device=dev)
        app = self.proposal_append_gt
        cnt = [p + (g if app else 0) for p, g in zip(P, G)]
        R = max(1, max(cnt))
        boxes_all = torch.zeros((N * R, D), dtype=torch.float32, device=dev)
        logits_all = torch.zeros((N * R,), dtype=torch.float32, device=dev)
        host = np.concatenate([i * R + np.arange(p, dtype=np.int64) for i, p in enumerate(P)] +
                              ([i * R + p + np.arange(g, dtype=np.int64) for i, (p, g) in enumerate(zip(P, G))] if app else []) +
                              [np.asarray(cnt, dtype=np.int64), np.asarray(g_off, dtype=np.int64)])
        meta = torch.from_numpy(host).to(dev, non_blocking=True)          # the one upload: destination rows, counts, gt offsets
        np_, ng_ = sum(P), (Gtot if app else 0)
        dest_p, dest_g = meta[:np_], meta[np_:np_ + ng_]
        counts, gt_off = meta[np_ + ng_:np_ + ng_ + N].to(torch.int32), meta[np_ + ng_ + N:].to(torch.int32)
        if np_:
            boxes_all[dest_p] = torch.cat([p.proposal_boxes.tensor for p in proposals]).float()
            logits_all[dest_p] = torch.cat([p.objectness_logits for p in proposals]).float()
        if ng_:                     # add_ground_truth_to_proposals
            boxes_all[dest_g] = gt_cat
            logits_all[dest_g] = gt_logit
        boxes_all = boxes_all.view(N, R, D)
        t = self.iou_thresholds[0]
        matches, cls8 = HF.roi_label_batched(boxes_all, counts, gt_cat, gtc_cat.to(torch.int32).contiguous(), gt_off, t, self.iou_labels, self.num_classes)
        mask, _ = HF.sample_labels(cls8, S, self.positive_fraction, self.num_classes)
        idx_all, num = HF.compact_samples(mask, S)
        num = num.cpu().tolist()             # the one host read: Instances have a host-side length
        safe = idx_all.clamp(min=0).long()
        boxes_s = torch.gather(boxes_all, 1, safe[:, :, None].expand(-1, -1, D))
        logits_s = torch.gather(logits_all.view(N, R), 1, safe)
        cls_s = torch.gather(cls8, 1, safe).to(torch.int64)
        if Gtot:
            rows = (gt_off[:-1].long()[:, None] + torch.gather(matches, 1, safe).long()).clamp(max=Gtot - 1)
            has_gt = (gt_off[1:] > gt_off[:-1]).to(torch.float32)[:, None, None]
            gtb_s = gt_cat[rows.reshape(-1)].view(N, S, D) * has_gt        # images without boxes: zero rows (never read: no foreground)
        else:
            gtb_s = torch.zeros((N, S, D), dtype=torch.float32, device=dev)
        out, sampled_rec = [], []
        for i, prop in enumerate(proposals):
            k = int(num[i])
            res = Instances(prop.image_size)
            res.proposal_boxes = BoxT(boxes_s[i, :k])
            res.objectness_logits = logits_s[i, :k]
            res.gt_classes = cls_s[i, :k]
            res.gt_boxes = BoxT(gtb_s[i, :k])
            out.append(res)
            sampled_rec.append(safe[i, :k])
        self.last_sampled = sampled_rec
        return out

    def forward(self, images, features, proposals, targets=None):
        feats = [features[f] for f in self.box_in_features]
        if self.training:
            proposals = self.label_and_sample_proposals(proposals, targets)
            self.last_proposals = proposals
        pooled = self.box_pooler(feats, [p.proposal_boxes for p in proposals])
        predictions = self.box_predictor(self.box_head(pooled))
        if self.training:
            return proposals, self.box_predictor.losses(predictions, proposals)
        return self.box_predictor.inference(predictions, proposals), {}


@ROI_HEADS_REGISTRY.register()
class RROIHeads(StandardROIHeads):
    """d2 RROIHeads: pairwise_iou_rotated matching, ROIAlignRotated pooling, 5-parameter box regression, rotated NMS."""
    rotated = True


@ROI_HEADS_REGISTRY.register()
class ProposalVisibleHead(StandardROIHeads):
    """slender_det/modeling/meta_arch/rcnn/pvrcnn.py:66-97: StandardROIHeads whose inference also hands the proposals back."""

    def forward(self, images, features, proposals, targets=None):
        out, losses = super().forward(images, features, proposals, targets)
        if self.training:
            return out, losses
        return out, {"proposals": proposals}


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)
