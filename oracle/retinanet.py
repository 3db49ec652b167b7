"""Oracle RetinaNet losses (CPU fp32): detectron2 RetinaNet.label_anchors / losses as documented by the reference's in-tree copy
slender_det/modeling/meta_arch/retina/retina_rotated.py:185-295 (axis-aligned boxes), with Box2BoxTransform.get_deltas
(SURVEY.md C.6), Matcher (C.5), anchors (C.7). Third-party parts "parity unpinned"."""
import torch

from . import detection as od
from . import losses as ol


def get_deltas(src, tgt, weights):
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tcx, tcy = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


def label_anchors(anchors, gt_boxes, gt_classes, thresholds, labels, num_classes):
    out_l, out_b = [], []
    for b, c in zip(gt_boxes, gt_classes):
        q = od.pairwise_iou(b, anchors) if len(b) else torch.zeros(0, len(anchors))
        matches, mlab = od.matcher(q, thresholds, labels, True)
        if len(b):
            mb = b[matches]
            gl = c[matches].clone()
            gl[mlab == 0] = num_classes
            gl[mlab == -1] = -1
        else:
            mb = torch.zeros_like(anchors)
            gl = torch.zeros_like(matches) + num_classes
        out_l.append(gl)
        out_b.append(mb)
    return torch.stack(out_l), torch.stack(out_b)


def apply_deltas(deltas, boxes, weights, clamp=4.135166556742356):
    """Box2BoxTransform.apply_deltas on XYXY boxes."""
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    dx, dy = deltas[:, 0] / weights[0], deltas[:, 1] / weights[1]
    dw, dh = (deltas[:, 2] / weights[2]).clamp(max=clamp), (deltas[:, 3] / weights[3]).clamp(max=clamp)
    pcx, pcy, pw, ph = dx * w + cx, dy * h + cy, torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), 1)


def losses(anchors, pred_logits, pred_deltas, gt_labels, matched_boxes, num_classes, alpha, gamma, beta, weights, normalizer, momentum=0.9,
           box_reg_loss_type="smooth_l1"):
    """pred_logits (N,R,K), pred_deltas (N,R,4). Returns (dict, new_normalizer)."""
    gt_deltas = torch.stack([get_deltas(anchors, k, weights) for k in matched_boxes])
    valid = gt_labels >= 0
    pos = valid & (gt_labels != num_classes)
    normalizer = momentum * normalizer + (1 - momentum) * max(int(pos.sum()), 1)
    target = torch.nn.functional.one_hot(gt_labels[valid].long(), num_classes + 1)[:, :-1].to(pred_logits.dtype)
    loss_cls = ol.sigmoid_focal_loss(pred_logits[valid], target, alpha, gamma, "sum")
    if box_reg_loss_type == "giou":       # retina_rotated.py:236-242
        pred_boxes = torch.stack([apply_deltas(k, anchors, weights) for k in pred_deltas])
        loss_box = ol.giou_loss_xyxy(pred_boxes[pos], (matched_boxes if torch.is_tensor(matched_boxes) else torch.stack(list(matched_boxes)))[pos], "sum")
    else:
        loss_box = ol.smooth_l1_loss(pred_deltas[pos], gt_deltas[pos], beta, "sum")
    return {"loss_cls": loss_cls / normalizer, "loss_box_reg": loss_box / normalizer}, normalizer
