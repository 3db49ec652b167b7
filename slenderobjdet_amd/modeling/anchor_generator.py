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


def rotated_cell_anchors(sizes, aspect_ratios, angles):
    """d2 RotatedAnchorGenerator.generate_cell_anchors: size -> ratio -> angle, (0, 0, w, h, a)."""
    out = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            out.extend([0.0, 0.0, w, h, float(a)] for a in angles)
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors_rotated(level_hw, strides, sizes, aspect_ratios, angles, offset=0.0, device="cpu"):
    n = len(level_hw)
    sizes = list(sizes) * n if len(sizes) == 1 else sizes
    aspect_ratios = list(aspect_ratios) * n if len(aspect_ratios) == 1 else aspect_ratios
    angles = list(angles) * n if len(angles) == 1 else angles
    out = []
    for (h, w), s, sz, ar, an in zip(level_hw, strides, sizes, aspect_ratios, angles):
        cell = rotated_cell_anchors(sz, ar, an).to(device)
        sx = torch.arange(offset * s, w * s, step=s, dtype=torch.float32, device=device)
        sy = torch.arange(offset * s, h * s, step=s, dtype=torch.float32, device=device)
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        z = torch.zeros_like(gx.reshape(-1))
        shifts = torch.stack((gx.reshape(-1), gy.reshape(-1), z, z, z), dim=1)
        out.append((shifts.view(-1, 1, 5) + cell.view(1, -1, 5)).reshape(-1, 5))
    return out


class DefaultAnchorGenerator:
    """cfg-driven wrapper with the d2 interface bits the RPN needs (``num_cell_anchors``, ``box_dim``, call on level sizes)."""
    box_dim = 4

    def __init__(self, cfg, input_shape):
        ag = cfg.MODEL.ANCHOR_GENERATOR
        self.strides = [s.stride for s in input_shape]
        self.sizes, self.ratios, self.offset = [list(s) for s in ag.SIZES], [list(a) for a in ag.ASPECT_RATIOS], ag.OFFSET
        self._cache = {}

    @property
    def num_cell_anchors(self):
        return [len(self.sizes[0]) * len(self.ratios[0])] * len(self.strides)

    def _make(self, level_hw, device):
        return grid_anchors(level_hw, self.strides, self.sizes, self.ratios, self.offset, device)

    def __call__(self, level_hw, device):
        key = (tuple(level_hw), str(device))
        if key not in self._cache:
            self._cache[key] = [a.contiguous() for a in self._make(level_hw, device)]
        return self._cache[key]


class RotatedAnchorGenerator(DefaultAnchorGenerator):
    box_dim = 5

    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        self.angles = [list(a) for a in cfg.MODEL.ANCHOR_GENERATOR.ANGLES]

    @property
    def num_cell_anchors(self):
        return [len(self.sizes[0]) * len(self.ratios[0]) * len(self.angles[0])] * len(self.strides)

    def _make(self, level_hw, device):
        return grid_anchors_rotated(level_hw, self.strides, self.sizes, self.ratios, self.angles, self.offset, device)


ANCHOR_GENERATORS = {"DefaultAnchorGenerator": DefaultAnchorGenerator, "RotatedAnchorGenerator": RotatedAnchorGenerator}


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATORS[cfg.MODEL.ANCHOR_GENERATOR.NAME](cfg, input_shape)
