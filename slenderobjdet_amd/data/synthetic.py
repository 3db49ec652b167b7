"""Synthetic COCO-shaped batches (there is no dataset on the box): the ``batched_inputs`` contract of
``model.forward`` (list[dict] with "image" CHW uint8 and "instances" carrying gt_boxes / gt_classes —
slender_det/modeling/meta_arch/fcos/fcosv2.py:63-82), generated as SURVEY.md §8(d) specifies:
uint8 images, G ~ clamp(Poisson(7), 1, 50) boxes, log-uniform 16..600 px sides, 20 % slender (1:5 .. 1:10), 80 classes.
"""
import torch

from ..structures import Boxes, Instances, RotatedBoxes


def synthetic_batch(n_images, height, width, seed=1234, num_classes=80, device="cpu", max_boxes=50, rotated=False):
    """``rotated``: gt boxes as RotatedBoxes (cx, cy, w, h, angle ~ U(-90, 90)) — SURVEY.md §8(d), the rotated R-CNN configuration."""
    g = torch.Generator().manual_seed(int(seed))
    out = []
    for _ in range(n_images):
        img = torch.randint(0, 256, (3, height, width), dtype=torch.uint8, generator=g)
        G = int(torch.poisson(torch.tensor([7.0]), generator=g).clamp(1, max_boxes).item())
        cx = torch.rand(G, generator=g) * width
        cy = torch.rand(G, generator=g) * height
        w = torch.pow(2.0, torch.rand(G, generator=g) * 5.2 + 4.0)
        h = torch.pow(2.0, torch.rand(G, generator=g) * 5.2 + 4.0)
        slender = torch.rand(G, generator=g) < 0.2
        ratio = torch.randint(5, 11, (G,), generator=g).float()
        tall = torch.rand(G, generator=g) < 0.5
        h = torch.where(slender & tall, w * ratio, h)
        w = torch.where(slender & ~tall, h * ratio, w)
        x1, y1 = (cx - w / 2).clamp(0, width), (cy - h / 2).clamp(0, height)
        x2, y2 = (cx + w / 2).clamp(0, width), (cy + h / 2).clamp(0, height)
        x2 = torch.maximum(x2, (x1 + 2).clamp(max=width))
        y2 = torch.maximum(y2, (y1 + 2).clamp(max=height))
        x1 = torch.minimum(x1, x2 - 2)
        y1 = torch.minimum(y1, y2 - 2)
        inst = Instances((height, width))
        inst.gt_boxes = Boxes(torch.stack([x1, y1, x2, y2], dim=1))
        inst.gt_classes = torch.randint(0, num_classes, (G,), generator=g)
        if rotated:
            ang = torch.rand(G, generator=g) * 180.0 - 90.0
            inst.gt_boxes = RotatedBoxes(torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1, ang], dim=1))
        if device != "cpu":
            img = img.to(device)
            inst = inst.to(device)
        out.append({"image": img, "instances": inst, "height": height, "width": width})
    return out


class SyntheticCocoBatches:
    """Infinite iterator of per-rank batches; seed = base + rank*1000 + iteration (SURVEY.md §8 d)."""

    def __init__(self, images_per_rank, height=800, width=1333, rank=0, base_seed=1234, device="cpu", pool=4, rotated=False):
        self.n, self.h, self.w, self.rank, self.base, self.device = images_per_rank, height, width, rank, base_seed, device
        self._pool = [synthetic_batch(self.n, self.h, self.w, self.base + rank * 1000 + i, device=device, rotated=rotated) for i in range(pool)]
        self._it = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self._pool[self._it % len(self._pool)]
        self._it += 1
        return b
