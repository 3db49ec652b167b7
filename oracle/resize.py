"""Oracle (test infrastructure, CPU, numpy): the resize + flip + normalise + pad stage in front of the model.

Restates PIL.Image.resize(size, BILINEAR) on a uint8 HWC image - Pillow's ImagingResample (src/libImaging/Resample.c): separable
triangle filter, support max(scale, 1), coefficients normalised in double and quantised to 22 fractional bits, horizontal pass
first, uint8 (rounded, clipped) intermediate, then the vertical pass - which is what detectron2's ResizeTransform.apply_image calls
(reached from slender_det/data/utils.py:42 through the dataset mapper), followed by HFlipTransform and FCOSV2.preprocess_image
(fcosv2.py:268-275).  **Parity unpinned** against Pillow itself (not installed in this image, no vectors in the reference's tests);
pinned properties: identity resize is exact, constant images stay constant, down-scaling by 2 of a 2-periodic pattern averages.
Written independently of slenderobjdet_amd/data/transforms.py (loops instead of tables).
"""
import math

import numpy as np


def _coeffs_1d(in_size, out_size):
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = fscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        ws = []
        for x in range(xmin, xmax):
            t = abs((x - center + 0.5) / fscale)
            ws.append(1.0 - t if t < 1.0 else 0.0)
        tot = sum(ws)
        ws = [w / tot if tot != 0.0 else w for w in ws]
        out.append((xmin, [int(w * (1 << 22) + (0.5 if w >= 0 else -0.5)) for w in ws]))
    return out


def _clip8(v):
    return 0 if v < 0 else (255 if v > 255 else v)


def pil_resize_bilinear(img, new_h, new_w):
    """img (H, W, C) uint8 -> (new_h, new_w, C) uint8."""
    H, W, C = img.shape
    src = img.astype(np.int64)
    cx = _coeffs_1d(W, new_w)
    tmp = np.zeros((H, new_w, C), dtype=np.int64)
    for x, (x0, ks) in enumerate(cx):
        acc = np.full((H, C), 1 << 21, dtype=np.int64)
        for t, k in enumerate(ks):
            acc += src[:, x0 + t, :] * k
        tmp[:, x, :] = np.clip(acc >> 22, 0, 255)
    cy = _coeffs_1d(H, new_h)
    out = np.zeros((new_h, new_w, C), dtype=np.int64)
    for y, (y0, ks) in enumerate(cy):
        acc = np.full((new_w, C), 1 << 21, dtype=np.int64)
        for t, k in enumerate(ks):
            acc += tmp[y0 + t, :, :] * k
        out[y] = np.clip(acc >> 22, 0, 255)
    return out.astype(np.uint8)


def resize_shortest_edge(h, w, size, max_size):
    scale = size / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        s = max_size / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def pipeline(images, choices, mean, std, div=32):
    """images: list of (H, W, 3) uint8 arrays; choices: (newh, neww, flip) -> (n, Hp, Wp, 3) float32 batch (zero padded)."""
    res = []
    for im, (nh, nw, flip) in zip(images, choices):
        r = pil_resize_bilinear(im, nh, nw)
        if flip:
            r = r[:, ::-1, :]
        res.append((r.astype(np.float32) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32))
    Hp = (max(r.shape[0] for r in res) + div - 1) // div * div
    Wp = (max(r.shape[1] for r in res) + div - 1) // div * div
    out = np.zeros((len(res), Hp, Wp, 3), dtype=np.float32)
    for i, r in enumerate(res):
        out[i, : r.shape[0], : r.shape[1]] = r
    return out


def transform_boxes(boxes, h, w, newh, neww, flip):
    b = np.array(boxes, dtype=np.float32).copy()
    b[:, 0::2] *= np.float32(neww / w)
    b[:, 1::2] *= np.float32(newh / h)
    if flip:
        b[:, [0, 2]] = np.float32(neww) - b[:, [2, 0]]
    b[:, 0::2] = np.clip(b[:, 0::2], 0, neww)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, newh)
    return b
