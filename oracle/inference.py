"""Oracle FCOS inference (CPU): FCOSV2.inference_single_image + postprocess, slender_det/modeling/meta_arch/fcos/fcosv2.py:194-266
(threshold -> x centerness -> top-k -> decode -> sqrt -> class-aware NMS -> top-100 -> rescale/clip)."""
import torch

from . import detection as od


def fcos_inference_single_image(locations, box_cls, box_reg, ctr, image_size, pre_nms_thresh=0.05, pre_nms_top_n=1000, nms_thresh=0.6,
                                max_det=100):
    """Per-level lists: locations (HW,2), box_cls (HW,K) logits, box_reg (HW,4) distances, ctr (HW,1) logits."""
    boxes_all, scores_all, cls_all = [], [], []
    for cls_i, reg_i, loc_i, ctr_i in zip(box_cls, box_reg, locations, ctr):
        p = cls_i.sigmoid()
        keep = p > pre_nms_thresh
        p = p * ctr_i.sigmoid()
        sc = p[keep]
        idx = keep.nonzero()
        li, ci = idx[:, 0], idx[:, 1]
        r, l = reg_i[li], loc_i[li]
        n = int(keep.sum())
        k = min(n, pre_nms_top_n)
        if n > k:
            sc, ti = sc.topk(k, sorted=False)
            ci, r, l = ci[ti], r[ti], l[ti]
        boxes_all.append(torch.stack([l[:, 0] - r[:, 0], l[:, 1] - r[:, 1], l[:, 0] + r[:, 2], l[:, 1] + r[:, 3]], dim=1))
        scores_all.append(torch.sqrt(sc))
        cls_all.append(ci)
    boxes, scores, classes = torch.cat(boxes_all), torch.cat(scores_all), torch.cat(cls_all)
    keep = od.batched_nms(boxes, scores, classes, nms_thresh)[:max_det]
    return boxes[keep], scores[keep], classes[keep], keep


def detector_postprocess(boxes, image_size, out_h, out_w):
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0].clamp_(0, out_w); b[:, 2].clamp_(0, out_w); b[:, 1].clamp_(0, out_h); b[:, 3].clamp_(0, out_h)
    ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    return b, ne
