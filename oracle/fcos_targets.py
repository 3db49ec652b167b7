"""Oracle for FCOS location grids and target assignment (CPU, fp32).

Restates slender_det/modeling/meta_arch/fcos/utils.py:82-212 (compute_locations, get_sample_region,
compute_targets_for_locations) and FCOSV2.get_ground_truth (fcosv2.py:150-172; the reference's
missing ``INF`` import there is taken from fcos/utils.py:7).
"""
import torch

INF = 100000000
SIZES_OF_INTEREST = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]


def locations(level_hw, strides):
    """utils.py:82-105: per level (H*W, 2) fp32 (x, y) = (j*s + s//2, i*s + s//2), row-major."""
    out = []
    for (h, w), s in zip(level_hw, strides):
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32)
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        out.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + s // 2)
    return out


def _center_region_mask(boxes, strides, pts_per_level, xs, ys, radius):
    """utils.py:108-157 (get_sample_region)."""
    L, G = xs.numel(), boxes.shape[0]
    cx = ((boxes[:, 0] + boxes[:, 2]) / 2)[None].expand(L, G)
    cy = ((boxes[:, 1] + boxes[:, 3]) / 2)[None].expand(L, G)
    if cx[..., 0].sum() == 0:   # utils.py:121-122 ("no gt"): first box centred on x == 0 disables sampling
        return torch.zeros(L, G, dtype=torch.bool)
    rad = torch.cat([torch.full((n,), float(s * radius), dtype=torch.float32) for n, s in zip(pts_per_level, strides)])[:, None]
    x0 = torch.maximum(cx - rad, boxes[None, :, 0].expand(L, G))
    y0 = torch.maximum(cy - rad, boxes[None, :, 1].expand(L, G))
    x1 = torch.minimum(cx + rad, boxes[None, :, 2].expand(L, G))
    y1 = torch.minimum(cy + rad, boxes[None, :, 3].expand(L, G))
    d = torch.stack((xs[:, None] - x0, ys[:, None] - y0, x1 - xs[:, None], y1 - ys[:, None]), dim=-1)
    return d.min(dim=-1).values > 0


def targets_for_image(locs, pts_per_level, strides, boxes, classes, radius, num_classes, sizes=SIZES_OF_INTEREST, return_inds=False):
    """One image of compute_targets_for_locations (utils.py:160-212).
    Returns labels (L,) int64 (background = num_classes) and reg targets (L,4) fp32."""
    xs, ys = locs[:, 0], locs[:, 1]
    L, G = xs.numel(), boxes.shape[0]
    if G == 0:   # the reference would raise on min over an empty dim; we define: all background
        return torch.full((L,), num_classes, dtype=torch.int64), torch.zeros(L, 4)
    ltrb = torch.stack((xs[:, None] - boxes[None, :, 0], ys[:, None] - boxes[None, :, 1],
                        boxes[None, :, 2] - xs[:, None], boxes[None, :, 3] - ys[:, None]), dim=2)   # (L,G,4)
    if radius > 0:
        inside = _center_region_mask(boxes, strides, pts_per_level, xs, ys, radius)
    else:
        inside = ltrb.min(dim=2).values > 0
    soi = torch.cat([torch.tensor(sizes[i], dtype=torch.float32)[None].expand(n, 2) for i, n in enumerate(pts_per_level)])
    big = ltrb.max(dim=2).values
    cared = (big >= soi[:, [0]]) & (big <= soi[:, [1]])
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))[None].repeat(L, 1)
    area[~inside] = INF
    area[~cared] = INF
    best, idx = area.min(dim=1)    # first minimum wins
    labels = classes.long()[idx].clone()
    labels[best == INF] = num_classes
    reg = ltrb[torch.arange(L), idx]
    if return_inds:
        return labels, reg, idx
    return labels, reg


def targets_for_batch(level_hw, strides, gt_boxes, gt_classes, radius, num_classes):
    """FCOSV2.get_ground_truth: returns labels (N,L), reg (N,L,4)."""
    locs = locations(level_hw, strides)
    pts = [len(l) for l in locs]
    allp = torch.cat(locs, dim=0)
    labs, regs = [], []
    for b, c in zip(gt_boxes, gt_classes):
        l, r = targets_for_image(allp, pts, strides, b.float(), c, radius, num_classes)
        labs.append(l)
        regs.append(r)
    return torch.stack(labs), torch.stack(regs)
