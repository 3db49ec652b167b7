"""Oracle losses (CPU, fp32/fp64).  Each function cites what it restates."""
import torch
import torch.nn.functional as F


def sigmoid_focal_loss(logits, targets, alpha=-1.0, gamma=2.0, reduction="none"):
    """fvcore.nn.sigmoid_focal_loss (source absent; formula SURVEY.md C.1); call site
    slender_det/modeling/meta_arch/fcos/fcosv2.py:124-127."""
    prob = torch.sigmoid(logits)
    bce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_true = prob * targets + (1 - prob) * (1 - targets)
    out = bce * (1 - p_true) ** gamma
    if alpha >= 0:
        out = out * (alpha * targets + (1 - alpha) * (1 - targets))
    if reduction == "sum":
        return out.sum()
    if reduction == "mean":
        return out.mean()
    return out


def one_hot_from_labels(labels, num_classes):
    """fcosv2.py:120-121: rows whose label is a foreground class get a 1 in that column."""
    t = torch.zeros(labels.numel(), num_classes, dtype=torch.float32)
    fg = (labels >= 0) & (labels != num_classes)
    t[fg.nonzero().squeeze(1), labels[fg].long()] = 1.0
    return t


def iou_loss_ltrb(pred, target, weight=None, loss_type="iou", reduce=True):
    """slender_det/layers/iou_loss.py:4-37 on (left, top, right, bottom) distances."""
    pl, pt, pr, pb = pred.unbind(dim=1)
    tl, tt, tr, tb = target.unbind(dim=1)
    area_t = (tl + tr) * (tt + tb)
    area_p = (pl + pr) * (pt + pb)
    inter_w = torch.minimum(pl, tl) + torch.minimum(pr, tr)
    inter_h = torch.minimum(pb, tb) + torch.minimum(pt, tt)
    hull_w = torch.maximum(pl, tl) + torch.maximum(pr, tr)
    hull_h = torch.maximum(pb, tb) + torch.maximum(pt, tt)
    hull = hull_w * hull_h + 1e-7
    inter = inter_w * inter_h
    union = area_t + area_p - inter
    iou = (inter + 1.0) / (union + 1.0)
    giou = iou - (hull - union) / hull
    if loss_type == "iou":
        per = -torch.log(iou)
    elif loss_type == "linear_iou":
        per = 1 - iou
    elif loss_type == "giou":
        per = 1 - giou
    else:
        raise NotImplementedError(loss_type)
    if weight is not None:
        per = per * weight
    return per.sum() if reduce else per


def giou_loss_xyxy(boxes1, boxes2, reduction="none", eps=1e-7):
    """fvcore.nn.giou_loss (source absent)."""
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    loss = 1 - (iou - (area_c - union) / (area_c + eps))
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def smooth_l1_loss(inp, target, beta, reduction="none"):
    """fvcore.nn.smooth_l1_loss == slender_det/layers/smooth_l1_loss_with_weight.py:3-18 w/o weight."""
    d = (inp - target).abs()
    loss = d if beta < 1e-5 else torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def centerness_targets(reg):
    """slender_det/modeling/meta_arch/fcos/utils.py:295-300."""
    lr = reg[:, [0, 2]]
    tb = reg[:, [1, 3]]
    c = (lr.min(dim=-1).values / lr.max(dim=-1).values) * (tb.min(dim=-1).values / tb.max(dim=-1).values)
    return torch.sqrt(c)


def fcos_losses(labels, reg_targets, cls_logits, box_pred, ctr_logits, num_classes, alpha, gamma, iou_type, world=1,
                total_pos=None, total_ctr=None):
    """FCOSV2.losses (fcosv2.py:104-148) on already flattened predictions:
    labels (M,), reg_targets (M,4), cls_logits (M,K), box_pred (M,4) [after exp/Scale], ctr_logits (M,).
    total_pos / total_ctr: all-reduced sums (default: this rank's)."""
    fg = (labels >= 0) & (labels != num_classes)
    n_pos = int(fg.sum()) if total_pos is None else total_pos
    pos_avg = max(n_pos / float(world), 1.0)
    onehot = one_hot_from_labels(labels, num_classes).to(cls_logits.dtype)
    cls_loss = sigmoid_focal_loss(cls_logits, onehot, alpha, gamma, "sum") / pos_avg
    if int(fg.sum()) > 0:
        ctr_t = centerness_targets(reg_targets[fg])
        ctr_sum = float(ctr_t.sum()) if total_ctr is None else total_ctr
        reg_loss = iou_loss_ltrb(box_pred[fg], reg_targets[fg], ctr_t, iou_type) / (ctr_sum / float(world))
        ctr_loss = F.binary_cross_entropy_with_logits(ctr_logits[fg], ctr_t, reduction="sum") / pos_avg
    else:
        reg_loss = box_pred[fg].sum()
        ctr_loss = ctr_logits[fg].sum()
    return {"cls_loss": cls_loss, "reg_loss": reg_loss, "centerness_loss": ctr_loss}
