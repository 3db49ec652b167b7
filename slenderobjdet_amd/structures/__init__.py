"""``detectron2.structures`` subset the hot path touches (reference call sites: fcosv2.py:14, fcos/utils.py:170-175):
Boxes, Instances, ImageList (semantics restated from SURVEY.md Appendix C.4 / C.8)."""
import itertools
import math
from typing import Any, Dict, List, Tuple

import torch


class Boxes:
    """(N,4) XYXY absolute fp32 boxes."""

    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1: Boxes, boxes2: Boxes):
    """SURVEY.md C.4."""
    a1, a2 = boxes1.area(), boxes2.area()
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))


class Instances:
    """Per-image container of equally long fields (gt_boxes, gt_classes, pred_boxes, scores, ...)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, f"Adding a field of length {data_len} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]):
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret

    def __repr__(self):
        s = self.__class__.__name__ + "("
        s += "num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self) if len(self._fields) else 0, self._image_size[0], self._image_size[1],
            ", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s


class ImageList:
    """Batched, padded images + original sizes. ``tensor`` layout is whatever the producer chose
    (NCHW for the reference-compatible ``from_tensors``; the HIP preprocess path builds NHWC(8) bf16 directly)."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def padded_size(sizes, size_divisibility=0):
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh = (mh + (d - 1)) // d * d
            mw = (mw + (d - 1)) // d * d
        return mh, mw

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert len(tensors) > 0
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        mh, mw = ImageList.padded_size(sizes, size_divisibility)
        batch = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (mh, mw), pad_value)
        for img, pad_img in zip(tensors, batch):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batch.contiguous(), sizes)


class RotatedBoxes(Boxes):
    """detectron2.structures.RotatedBoxes (source absent): (N,5) = (cx, cy, w, h, angle in degrees, CCW positive)."""

    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 5)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 5, tensor.size()
        self.tensor = tensor

    def clone(self):
        return RotatedBoxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return RotatedBoxes(self.tensor.to(*args, **kwargs))

    def area(self):
        return self.tensor[:, 2] * self.tensor[:, 3]

    def normalize_angles(self):
        self.tensor[:, 4] = (self.tensor[:, 4] + 180.0) % 360.0 - 180.0

    def clip(self, box_size, clip_angle_threshold: float = 1.0):
        """Only nearly-horizontal boxes are clipped (to XYXY, clamp, back), as d2 does."""
        h, w = box_size
        self.normalize_angles()
        idx = torch.where(torch.abs(self.tensor[:, 4]) <= clip_angle_threshold)[0]
        x1 = self.tensor[idx, 0] - self.tensor[idx, 2] / 2.0
        y1 = self.tensor[idx, 1] - self.tensor[idx, 3] / 2.0
        x2 = self.tensor[idx, 0] + self.tensor[idx, 2] / 2.0
        y2 = self.tensor[idx, 1] + self.tensor[idx, 3] / 2.0
        x1.clamp_(min=0, max=w); y1.clamp_(min=0, max=h); x2.clamp_(min=0, max=w); y2.clamp_(min=0, max=h)
        self.tensor[idx, 0] = (x1 + x2) / 2.0
        self.tensor[idx, 1] = (y1 + y2) / 2.0
        self.tensor[idx, 2] = torch.min(self.tensor[idx, 2], x2 - x1)
        self.tensor[idx, 3] = torch.min(self.tensor[idx, 3], y2 - y1)

    def nonempty(self, threshold: float = 0.0):
        return (self.tensor[:, 2] > threshold) & (self.tensor[:, 3] > threshold)

    def get_centers(self):
        return self.tensor[:, :2]

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0] *= scale_x
        self.tensor[:, 1] *= scale_y
        theta = self.tensor[:, 4] * math.pi / 180.0
        c, s = torch.cos(theta), torch.sin(theta)
        self.tensor[:, 2] *= torch.sqrt((scale_x * c) ** 2 + (scale_y * s) ** 2)
        self.tensor[:, 3] *= torch.sqrt((scale_x * s) ** 2 + (scale_y * c) ** 2)
        self.tensor[:, 4] = torch.atan2(scale_x * s, scale_y * c) * 180 / math.pi

    def __getitem__(self, item):
        if isinstance(item, int):
            return RotatedBoxes(self.tensor[item].view(1, -1))
        return RotatedBoxes(self.tensor[item])

    def __repr__(self):
        return "RotatedBoxes(" + str(self.tensor) + ")"

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))


def pairwise_iou_rotated(boxes1: RotatedBoxes, boxes2: RotatedBoxes):
    from ..layers import functional as HF

    return HF.box_iou_rotated(boxes1.tensor.float().contiguous(), boxes2.tensor.float().contiguous())
