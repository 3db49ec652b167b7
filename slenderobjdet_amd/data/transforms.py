"""Augmentation geometry of the reference's input pipeline, computed on the host, applied on the device.

``build_augmentation`` (slender_det/data/utils.py:29-50) = detectron2 ``ResizeShortestEdge`` (train) / the reference's
``ResizeLongestEdge`` (test) followed by ``RandomFlip`` (train); the dataset mapper (slender_det/data/mappers/base.py:158-252) applies
them to the decoded uint8 HWC image with PIL and to the boxes, then ``preprocess_image`` normalises and pads on the device.  Here the
host only draws the random choices and builds the small filter tables; ``DeviceInputPipeline`` runs resize + flip + normalise + pad +
NHWC(8) bf16 for the whole batch in ONE kernel (sod_resize_flip_preprocess_batch) and transforms the boxes with the same numbers.

``pil_bilinear_coeffs`` restates Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c [upstream
knowledge; Pillow is not installed in this image]) - the triangle filter whose support widens with the down-scaling factor.
"""
import collections
import ctypes
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def resize_shortest_edge_size(h, w, size, max_size):
    """detectron2 ResizeShortestEdge.get_transform: output (newh, neww)."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def pil_bilinear_coeffs(in_size, out_size):
    """-> bounds int32 [out][2] (first source index, taps), coefficients int32 [out][ksize] (fixed point, PRECISION_BITS)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                       # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        wgt = np.clip(1.0 - np.abs((x + xmin - center + 0.5) * ss), 0.0, None)
        tot = wgt.sum()
        if tot != 0.0:
            wgt = wgt / tot
        kk[xx, :xmax] = np.where(wgt < 0, wgt * (1 << PRECISION_BITS) - 0.5, wgt * (1 << PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def transform_boxes(boxes, h, w, newh, neww, flip):
    """ResizeTransform.apply_box + HFlipTransform.apply_box on XYXY boxes (float tensor, any device), clipped as the mapper does
    (detectron2 transform_instance_annotations: clip to the new image size)."""
    b = boxes.clone().float()
    b[:, 0::2] *= neww * 1.0 / w
    b[:, 1::2] *= newh * 1.0 / h
    if flip:
        x1 = neww - b[:, 2]
        x2 = neww - b[:, 0]
        b[:, 0], b[:, 2] = x1, x2
    b[:, 0::2].clamp_(0, neww)
    b[:, 1::2].clamp_(0, newh)
    return b


class DeviceInputPipeline:
    """Train-time augmentation of the reference (INPUT.MIN_SIZE_TRAIN choice, MAX_SIZE_TRAIN, horizontal flip with p = 0.5) + the
    model's preprocess_image, for a list of decoded uint8 HWC images already on the device."""

    def __init__(self, min_sizes=(640, 672, 704, 736, 768, 800), max_size=1333, flip_prob=0.5, pixel_mean=(103.53, 116.28, 123.675),
                 pixel_std=(1.0, 1.0, 1.0), size_divisibility=32, seed=0):
        self.min_sizes, self.max_size, self.flip_prob = tuple(min_sizes), max_size, flip_prob
        self.mean, self.std, self.div = [float(v) for v in pixel_mean], [float(v) for v in pixel_std], size_divisibility
        self.rng = np.random.RandomState(seed)
        self._tables = collections.OrderedDict()

    # COCO has thousands of distinct (source size, target size) pairs: the coefficient tables are an LRU of bounded size (one call
    # uses at most 2 x 64 of them; an evicted table is freed in stream order, behind the launch that read it)
    MAX_TABLES = 1024

    def _coeffs(self, in_size, out_size, device):
        key = (in_size, out_size, device)
        t = self._tables.get(key)
        if t is None:
            b, k = pil_bilinear_coeffs(in_size, out_size)
            t = self._tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
            while len(self._tables) > self.MAX_TABLES:
                self._tables.popitem(last=False)
        else:
            self._tables.move_to_end(key)
        return t

    def draw(self, h, w):
        """The random part of one image: (newh, neww, flip)."""
        size = int(self.rng.choice(self.min_sizes))
        newh, neww = resize_shortest_edge_size(h, w, size, self.max_size)
        return newh, neww, bool(self.rng.rand() < self.flip_prob)

    def __call__(self, images, boxes=None, choices=None):
        """images: list of (H, W, 3) uint8 CUDA tensors; boxes: optional list of (G, 4) XYXY tensors; choices: optional list of
        (newh, neww, flip) to apply instead of drawing them.  Returns (batch (n, Hp, Wp, 8) bf16, image_sizes, boxes, choices)."""
        from .._C import SlenderHipError, call, ptr, stream_ptr

        n = len(images)
        if n == 0 or n > 64:
            raise SlenderHipError("DeviceInputPipeline: 1..64 images per call")
        dev = images[0].device
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_cuda or not im.is_contiguous():
                raise SlenderHipError("DeviceInputPipeline: images must be contiguous (H, W, 3) uint8 CUDA tensors")
        if choices is None:
            choices = [self.draw(int(im.shape[0]), int(im.shape[1])) for im in images]
        Hp = (max(c[0] for c in choices) + self.div - 1) // self.div * self.div
        Wp = (max(c[1] for c in choices) + self.div - 1) // self.div * self.div
        out = torch.empty((n, Hp, Wp, 8), dtype=torch.bfloat16, device=dev)
        tabs = [(self._coeffs(int(im.shape[1]), c[1], dev), self._coeffs(int(im.shape[0]), c[0], dev)) for im, c in zip(images, choices)]
        ia = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
        pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        fa = lambda v: ctypes.cast((ctypes.c_float * 3)(*v), ctypes.c_void_p)
        call("sod_resize_flip_preprocess_batch", n, pa(images), ia(im.shape[0] for im in images), ia(im.shape[1] for im in images),
             ia(c[0] for c in choices), ia(c[1] for c in choices),
             pa([t[0][0] for t in tabs]), pa([t[0][1] for t in tabs]), ia(t[0][2] for t in tabs),
             pa([t[1][0] for t in tabs]), pa([t[1][1] for t in tabs]), ia(t[1][2] for t in tabs), ia(int(c[2]) for c in choices),
             ptr(out), Hp, Wp, 8, fa(self.mean), fa(self.std), stream_ptr())
        new_boxes = None
        if boxes is not None:
            new_boxes = [transform_boxes(b, int(im.shape[0]), int(im.shape[1]), c[0], c[1], c[2]) for b, im, c in zip(boxes, images, choices)]
        return out, [(c[0], c[1]) for c in choices], new_boxes, choices
