"""batched_nms with the torchvision / detectron2 contract: class-offset trick, greedy suppression
of IoU > threshold in stable descending-score order, kept indices returned in score order."""
import torch


def nms(boxes, scores, iou_threshold):
    from . import functional as HF

    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    return HF.nms(boxes.float().contiguous(), scores.float().contiguous(), float(iou_threshold))


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def nms_rotated(boxes, scores, iou_threshold):
    from . import functional as HF

    return HF.nms_rotated(boxes.float().contiguous(), scores.float().contiguous(), float(iou_threshold))


def batched_nms_rotated(boxes, scores, idxs, iou_threshold):
    """detectron2.layers.batched_nms_rotated: shift centres per class so classes never overlap."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    max_coordinate = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    min_coordinate = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.max(boxes[:, 2], boxes[:, 3]) / 2).min()
    offsets = idxs.to(boxes) * (max_coordinate - min_coordinate + 1)
    shifted = boxes.clone()
    shifted[:, :2] += offsets[:, None]
    return nms_rotated(shifted, scores, iou_threshold)
