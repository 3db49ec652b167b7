"""detectron2 DefaultAnchorGenerator (source absent): per level cell anchors (sizes outer, aspect ratios inner)
shifted over the grid, order (H, W, A)."""
import math

import torch


def cell_anchors(sizes, aspect_ratios):
    out = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(level_hw, strides, sizes, aspect_ratios, offset=0.0, device="cpu"):
    """Returns a list (per level) of (H*W*A, 4) anchors."""
    if len(sizes) == 1:
        sizes = list(sizes) * len(level_hw)
    if len(aspect_ratios) == 1:
        aspect_ratios = list(aspect_ratios) * len(level_hw)
    out = []
    for (h, w), s, sz, ar in zip(level_hw, strides, sizes, aspect_ratios):
        cell = cell_anchors(sz, ar).to(device)
        sx = torch.arange(offset * s, w * s, step=s, dtype=torch.float32, device=device)
        sy = torch.arange(offset * s, h * s, step=s, dtype=torch.float32, device=device)
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((gx.reshape(-1), gy.reshape(-1), gx.reshape(-1), gy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return out
