"""Oracle two-stage detector (CPU, fp32, NCHW, plain torch / numpy) — TEST INFRASTRUCTURE, never imported by the product path.

Restates detectron2's GeneralizedRCNN + RPN / RRPN + StandardROIHeads / RROIHeads (BASELINE config 5 =
configs/rotated/faster_R_101.yaml over Base-RRCNN-FPN.yaml; the reference's own subclasses proposal_generator/rpn.py:26-356,
roi_heads/roi_heads.py:27-66 build on them).  detectron2's source exists nowhere in this environment: every function here is
"parity unpinned" against upstream and follows the contracts written down in SURVEY.md §2.3 / Appendix C.4-C.7, C.13-C.15.
"""
import math

import torch
import torch.nn.functional as F

from . import detection as od
from . import losses as ol
from .model import _RoundSTE
from .reppoints import OracleRepPoints
from .nn import relu as _band_relu      # torch.relu unless a ReluBand context is active (oracle/nn.py)
from .nn import relu_at as _relu_at       # ... or teacher-forced decisions for this position (ForcedMasks)

SCALE_CLAMP = math.log(1000.0 / 16)


# ------------------------------------------------------------------------------------------------ anchors / box coding
def anchors(level_hw, strides, sizes, ratios, angles=None, offset=0.0):
    """DefaultAnchorGenerator / RotatedAnchorGenerator: per level (H*W*A, 4|5), cell order size -> ratio (-> angle)."""
    out = []
    for lvl, ((h, w), s) in enumerate(zip(level_hw, strides)):
        sz = sizes[lvl] if len(sizes) > 1 else sizes[0]
        ar = ratios[lvl] if len(ratios) > 1 else ratios[0]
        cell = []
        for size in sz:
            area = size ** 2.0
            for r in ar:
                ww = math.sqrt(area / r)
                hh = r * ww
                if angles is None:
                    cell.append([-ww / 2.0, -hh / 2.0, ww / 2.0, hh / 2.0])
                else:
                    an = angles[lvl] if len(angles) > 1 else angles[0]
                    cell.extend([0.0, 0.0, ww, hh, float(a)] for a in an)
        cell = torch.tensor(cell, dtype=torch.float32)
        sx = torch.arange(offset * s, w * s, step=s, dtype=torch.float32)
        sy = torch.arange(offset * s, h * s, step=s, dtype=torch.float32)
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        gx, gy = gx.reshape(-1), gy.reshape(-1)
        z = torch.zeros_like(gx)
        shifts = torch.stack((gx, gy, gx, gy), 1) if angles is None else torch.stack((gx, gy, z, z, z), 1)
        D = shifts.shape[1]
        out.append((shifts.view(-1, 1, D) + cell.view(1, -1, D)).reshape(-1, D))
    return out


def get_deltas(src, tgt, weights):
    if src.shape[1] == 4:
        sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
        scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
        tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
        tcx, tcy = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
        wx, wy, ww, wh = weights
        return torch.stack((wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), 1)
    wx, wy, ww, wh, wa = weights
    da = tgt[:, 4] - src[:, 4]
    da = (da + 180.0) % 360.0 - 180.0
    return torch.stack((wx * (tgt[:, 0] - src[:, 0]) / src[:, 2], wy * (tgt[:, 1] - src[:, 1]) / src[:, 3],
                        ww * torch.log(tgt[:, 2] / src[:, 2]), wh * torch.log(tgt[:, 3] / src[:, 3]), wa * da * math.pi / 180.0), 1)


def apply_deltas(deltas, boxes, weights):
    """deltas (N, k*D), boxes (N, D) -> (N, k*D)."""
    D = boxes.shape[1]
    d = deltas.view(deltas.shape[0], -1, D)
    if D == 4:
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
        dx, dy = d[..., 0] / weights[0], d[..., 1] / weights[1]
        dw, dh = (d[..., 2] / weights[2]).clamp(max=SCALE_CLAMP), (d[..., 3] / weights[3]).clamp(max=SCALE_CLAMP)
        pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
        pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
        return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), -1).reshape(deltas.shape[0], -1)
    dx, dy = d[..., 0] / weights[0], d[..., 1] / weights[1]
    dw, dh = (d[..., 2] / weights[2]).clamp(max=SCALE_CLAMP), (d[..., 3] / weights[3]).clamp(max=SCALE_CLAMP)
    da = d[..., 4] / weights[4]
    pa = da * 180.0 / math.pi + boxes[:, 4:5]
    pa = (pa + 180.0) % 360.0 - 180.0
    return torch.stack((dx * boxes[:, 2:3] + boxes[:, 0:1], dy * boxes[:, 3:4] + boxes[:, 1:2], torch.exp(dw) * boxes[:, 2:3],
                        torch.exp(dh) * boxes[:, 3:4], pa), -1).reshape(deltas.shape[0], -1)


def iou_matrix(gt, boxes):
    return od.pairwise_iou_rotated(gt, boxes) if gt.shape[1] == 5 else od.pairwise_iou(gt, boxes)


def clip_boxes(b, size):
    h, w = size
    b = b.clone()
    if b.shape[1] == 4:
        b[:, 0::2] = b[:, 0::2].clamp(0, w)
        b[:, 1::2] = b[:, 1::2].clamp(0, h)
        return b
    b[:, 4] = (b[:, 4] + 180.0) % 360.0 - 180.0
    idx = torch.where(b[:, 4].abs() <= 1.0)[0]
    x1, y1 = (b[idx, 0] - b[idx, 2] / 2).clamp(0, w), (b[idx, 1] - b[idx, 3] / 2).clamp(0, h)
    x2, y2 = (b[idx, 0] + b[idx, 2] / 2).clamp(0, w), (b[idx, 1] + b[idx, 3] / 2).clamp(0, h)
    b[idx, 0], b[idx, 1] = (x1 + x2) / 2, (y1 + y2) / 2
    b[idx, 2], b[idx, 3] = torch.min(b[idx, 2], x2 - x1), torch.min(b[idx, 3], y2 - y1)
    return b


def nonempty(b, thr=0.0):
    if b.shape[1] == 4:
        return ((b[:, 2] - b[:, 0]) > thr) & ((b[:, 3] - b[:, 1]) > thr)
    return (b[:, 2] > thr) & (b[:, 3] > thr)


def batched_nms_any(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if boxes.shape[1] == 4:
        return od.batched_nms(boxes, scores, idxs, thr)
    mx = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    mn = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.max(boxes[:, 2], boxes[:, 3]) / 2).min()
    sh = boxes.clone()
    sh[:, :2] += (idxs.to(boxes) * (mx - mn + 1))[:, None]
    return od.nms_rotated(sh, scores, thr)


# ------------------------------------------------------------------------------------------------ RPN
def rpn_match(anchors_cat, gt_boxes, thresholds=(0.3, 0.7), labels=(0, -1, 1)):
    """Matcher part of RPN.label_and_sample_anchors (before the random subsampling): matches, labels in {0, -1, 1}."""
    q = iou_matrix(gt_boxes, anchors_cat) if len(gt_boxes) else torch.zeros(0, len(anchors_cat))
    return od.matcher(q, list(thresholds), list(labels), True)


def rpn_losses(logits, deltas, gt_labels, gt_deltas, batch_size_per_image=256, beta=0.0):
    """logits (N,R), deltas (N,R,D), gt_labels (N,R) in {-1,0,1}."""
    pos, valid = gt_labels == 1, gt_labels >= 0
    loc = ol.smooth_l1_loss(deltas[pos], gt_deltas[pos], beta, "sum")
    cls = F.binary_cross_entropy_with_logits(logits[valid], gt_labels[valid].float(), reduction="sum")
    norm = batch_size_per_image * logits.shape[0]
    return {"loss_rpn_cls": cls / norm, "loss_rpn_loc": loc / norm}


def find_top_proposals(props_l, logits_l, image_sizes, nms_thresh, pre_topk, post_topk, min_size=0.0):
    """props_l / logits_l: per level (N, HWA, D) / (N, HWA). Returns per image (boxes, scores)."""
    N = logits_l[0].shape[0]
    S, P, L = [], [], []
    for lvl, (p, lg) in enumerate(zip(props_l, logits_l)):
        num = min(pre_topk, lg.shape[1])
        sc, idx = lg.sort(descending=True, dim=1)
        sc, idx = sc[:, :num], idx[:, :num]
        S.append(sc)
        P.append(p[torch.arange(N)[:, None], idx])
        L.append(torch.full((num,), lvl, dtype=torch.int64))
    S, P, L = torch.cat(S, 1), torch.cat(P, 1), torch.cat(L)
    out = []
    for n, size in enumerate(image_sizes):
        b, s, l = clip_boxes(P[n], size), S[n], L
        keep = nonempty(b, min_size)
        b, s, l = b[keep], s[keep], l[keep]
        keep = batched_nms_any(b, s, l, nms_thresh)[:post_topk]
        out.append((b[keep], s[keep]))
    return out


# ------------------------------------------------------------------------------------------------ ROI heads
def assign_levels(areas, min_level, max_level, canonical_size=224, canonical_level=4):
    lv = torch.floor(canonical_level + torch.log2(torch.sqrt(areas) / canonical_size + 1e-8))
    return (lv.clamp(min_level, max_level) - min_level).long()


def roi_pool(feats, rois, scales, out_size=7, sampling_ratio=0):
    """feats list of (N,C,H,W); rois (M, 1+D) pooler format. Returns (M, C, out, out)."""
    D = rois.shape[1] - 1
    areas = rois[:, 3] * rois[:, 4] if D == 5 else (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    min_level, max_level = int(-math.log2(scales[0])), int(-math.log2(scales[-1]))
    lv = assign_levels(areas, min_level, max_level) if len(feats) > 1 else torch.zeros(len(rois), dtype=torch.long)
    out = torch.zeros(len(rois), feats[0].shape[1], out_size, out_size, dtype=feats[0].dtype)
    for l, (f, s) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(lv == l).squeeze(1)
        if len(idx):
            ra = od.roi_align_vec if od.ROI_ALIGN_IMPL == "vec" else od.roi_align
            out = out.index_put((idx,), ra(f, rois[idx], (out_size, out_size), s, sampling_ratio, rotated=(D == 5)))
    return out


def roi_match(gt_boxes, gt_classes, boxes, num_classes, thr=0.5):
    """IoU + Matcher([thr], [0, 1]) + class assignment of ROIHeads.label_and_sample_proposals (before the random subsampling)."""
    if len(gt_boxes) == 0:
        return torch.zeros(len(boxes), dtype=torch.int64), torch.full((len(boxes),), num_classes, dtype=torch.int64)
    m, lab = od.matcher(iou_matrix(gt_boxes, boxes), [thr], [0, 1], False)
    cls = gt_classes[m].clone().long()
    cls[lab == 0] = num_classes
    return m, cls


def fast_rcnn_losses(scores, deltas, gt_classes, gt_deltas, num_classes, beta=0.0):
    """scores (R,K+1), deltas (R,K*D)."""
    D = gt_deltas.shape[1]
    loss_cls = F.cross_entropy(scores, gt_classes.long(), reduction="mean")
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    cols = D * gt_classes[fg].long()[:, None] + torch.arange(D)
    loss_box = ol.smooth_l1_loss(deltas[fg[:, None], cols], gt_deltas[fg], beta, "sum") / max(gt_classes.numel(), 1)
    return {"loss_cls": loss_cls, "loss_box_reg": loss_box}


def fast_rcnn_inference_single_image(boxes, probs, image_size, score_thresh, nms_thresh, topk):
    """boxes (R, K*D), probs (R, K+1)."""
    K = probs.shape[1] - 1
    D = boxes.shape[1] // K
    scores = probs[:, :-1]
    b = clip_boxes(boxes.reshape(-1, D), image_size).view(-1, K, D)
    mask = scores > score_thresh
    inds = mask.nonzero()
    b, s = b[mask], scores[mask]
    keep = batched_nms_any(b, s, inds[:, 1], nms_thresh)[:topk]
    return b[keep], s[keep], inds[keep, 1]


# ------------------------------------------------------------------------------------------------ whole model
class OracleRCNN(OracleRepPoints):
    """Functional GeneralizedRCNN; the random anchor / proposal subsampling and the proposals themselves are taken from the run
    under test (``targets``), everything differentiable is recomputed here."""

    @classmethod
    def from_hip_model(cls, model, emulate_bf16=False):
        from slenderobjdet_amd.layers.nn import HipConv2d

        params, buffers = {}, {}
        for name, m in model.named_modules():
            if isinstance(m, HipConv2d):
                params[name + ".weight"] = m.weight.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(m.weight.requires_grad)
                if m.bias is not None:
                    params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(m.bias.requires_grad)
                if m.frozen_bn:
                    scale = m.bn_weight.float().cpu() * torch.rsqrt(m.bn_running_var.float().cpu() + 1e-5)
                    buffers[name + ".scale"], buffers[name + ".shift"] = scale, m.bn_bias.float().cpu() - m.bn_running_mean.float().cpu() * scale
        bu = model.backbone.bottom_up
        res_names = [name for _, name in bu.stages_and_names]
        rpn, roi = model.proposal_generator, model.roi_heads
        c = dict(
            blocks={n: len(getattr(bu, n)) for n in res_names}, bottleneck=any(k.endswith("conv3.weight") for k in params),
            mean=[float(v) for v in model.pixel_mean.flatten()], std=[float(v) for v in model.pixel_std.flatten()],
            size_div=model.backbone.size_divisibility, fpn_in=list(model.backbone.in_features), fpn_norm=model.backbone.norm,
            stride_in_1x1={n: [blk.conv1.stride for blk in getattr(bu, n)] for n in res_names},
            block_stride={n: [blk.stride for blk in getattr(bu, n)] for n in res_names},
            rpn_in=list(rpn.in_features), roi_in=list(roi.box_in_features), A=rpn.head.num_anchors, D=rpn.box_dim,
            rpn_weights=rpn.box2box_transform.weights, roi_weights=roi.box_predictor.box2box_transform.weights,
            rpn_bs=rpn.batch_size_per_image, rpn_beta=rpn.smooth_l1_beta, roi_beta=roi.box_predictor.smooth_l1_beta,
            K=roi.num_classes, scales=list(roi.box_pooler.scales), pool=roi.box_pooler.output_size[0],
            sampling_ratio=roi.box_pooler.sampling_ratio, num_fc=len(roi.box_head.fcs),
        )
        return cls(params, buffers, c, emulate_bf16)

    def features(self, batched_inputs):
        c = self.c
        x = self.preprocess(batched_inputs)
        feats = self._bottom_up(x)
        prev, outs = None, {}
        for n in c["fpn_in"][::-1]:
            s = int(n[-1])
            up = F.interpolate(prev, scale_factor=2, mode="nearest") if prev is not None else None
            prev = self._conv(f"backbone.fpn_lateral{s}", feats[n], 1, 0, res=up)
            outs[f"p{s}"] = self._conv(f"backbone.fpn_output{s}", prev, 1, 1)
        outs["p6"] = F.max_pool2d(outs["p5"], kernel_size=1, stride=2, padding=0)
        return outs

    def rpn_outputs(self, feats):
        c = self.c
        A, D = c["A"], c["D"]
        logits, deltas = [], []
        for f in (feats[k] for k in c["rpn_in"]):
            N = f.shape[0]
            t = self._conv("proposal_generator.head.conv.conv", f, 1, 1, relu=True)
            lg = F.conv2d(t, self._w("proposal_generator.head.objectness_logits.conv.weight")[:A], self.p["proposal_generator.head.objectness_logits.conv.bias"][:A])
            dl = F.conv2d(t, self._w("proposal_generator.head.anchor_deltas.conv.weight")[:A * D], self.p["proposal_generator.head.anchor_deltas.conv.bias"][:A * D])
            logits.append(lg.permute(0, 2, 3, 1).reshape(N, -1))
            deltas.append(dl.view(N, A, D, *dl.shape[-2:]).permute(0, 3, 4, 1, 2).reshape(N, -1, D))
        return logits, deltas

    def box_head(self, pooled):
        """pooled (M, C, 7, 7) -> scores (M, K+1), deltas (M, K*D).  FC1 of the product path is stored in (h, w, c) flatten order."""
        c = self.c
        K, D = c["K"], c["D"]
        M, C, P, _ = pooled.shape
        x = self._act(pooled).permute(0, 2, 3, 1).reshape(M, -1)          # the product path pools to bf16 NHWC
        for i in range(c["num_fc"]):
            w = self._w(f"roi_heads.box_head.fcs.{i}.weight").flatten(1)
            x = self._act(_relu_at(F.linear(x, w, self.p[f"roi_heads.box_head.fcs.{i}.bias"]), f"roi_heads.box_head.fcs.{i}"))
        s = F.linear(x, self._w("roi_heads.box_predictor.cls_score.weight").flatten(1)[: K + 1], self.p["roi_heads.box_predictor.cls_score.bias"][: K + 1])
        d = F.linear(x, self._w("roi_heads.box_predictor.bbox_pred.weight").flatten(1)[: K * D], self.p["roi_heads.box_predictor.bbox_pred.bias"][: K * D])
        return s, d

    def losses(self, batched_inputs, rpn_labels, rpn_gt_deltas, rois, roi_classes, roi_gt_boxes):
        """rpn_labels (N,R) / rpn_gt_deltas (N,R,D): the sampled anchor targets; rois (M, 1+D) pooler format, roi_classes (M,),
        roi_gt_boxes (M, D): the sampled proposals — all taken from the run under test."""
        c = self.c
        feats = self.features(batched_inputs)
        logits, deltas = self.rpn_outputs(feats)
        out = rpn_losses(torch.cat(logits, 1), torch.cat(deltas, 1), rpn_labels, rpn_gt_deltas, c["rpn_bs"], c["rpn_beta"])
        pooled = roi_pool([feats[k] for k in c["roi_in"]], rois, c["scales"], c["pool"], c["sampling_ratio"])
        s, d = self.box_head(pooled)
        gt_d = get_deltas(rois[:, 1:], roi_gt_boxes, c["roi_weights"])
        out.update(fast_rcnn_losses(s, d, roi_classes, gt_d, c["K"], c["roi_beta"]))
        return out
