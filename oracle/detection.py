"""Oracle restatements of the detection operators detectron2 / torchvision provide (sources absent from the reference tree:
"parity unpinned" vs upstream; contracts fixed in SURVEY.md Appendix C.5, C.12, C.13, C.14 and exercised with adversarial
fixtures — ties, touching boxes, zero-area — in tests/)."""
import math

import numpy as np
import torch


def nms(boxes, scores, thr):
    """torchvision.ops.nms: stable descending score order, greedy, suppress IoU > thr, IoU = inter/(a+b-inter) in fp32."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    order = torch.sort(scores.detach().cpu(), descending=True, stable=True).indices.numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, dead = [], np.zeros(len(b), dtype=bool)
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        left = np.maximum(b[i, 0], b[rest, 0]); right = np.minimum(b[i, 2], b[rest, 2])
        top = np.maximum(b[i, 1], b[rest, 1]); bottom = np.minimum(b[i, 3], b[rest, 3])
        inter = np.maximum(right - left, np.float32(0)) * np.maximum(bottom - top, np.float32(0))
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + offsets[:, None], scores, thr)


def _bilinear(feat, y, x):
    """feat (C,H,W); detectron2 ROIAlign bilinear_interpolate."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C, dtype=feat.dtype)
    y, x = max(y, 0.0), max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(x, rois, output_size, scale, sampling_ratio=0, rotated=False):
    """x (N,C,H,W) float; rois (R,5) or (R,6 rotated). aligned=True. Returns (R,C,PH,PW). Differentiable w.r.t. x."""
    PH, PW = output_size
    outs = []
    for roi in rois.tolist():
        b = int(roi[0])
        if rotated:
            cw, ch = roi[1] * scale - 0.5, roi[2] * scale - 0.5
            rw, rh = roi[3] * scale, roi[4] * scale
            th = roi[5] * math.pi / 180.0
            ct, st = math.cos(th), math.sin(th)
            sh, sw = -rh / 2.0, -rw / 2.0
        else:
            sw, sh = roi[1] * scale - 0.5, roi[2] * scale - 0.5
            rw, rh = roi[3] * scale - 0.5 - sw, roi[4] * scale - 0.5 - sh
        bh, bw = rh / PH, rw / PW
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / PH))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / PW))
        cnt = max(gh * gw, 1)
        o = torch.zeros(x.shape[1], PH, PW, dtype=x.dtype)
        for ph in range(PH):
            for pw in range(PW):
                acc = torch.zeros(x.shape[1], dtype=x.dtype)
                for iy in range(gh):
                    yy = sh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = sw + pw * bw + (ix + 0.5) * bw / gw
                        if rotated:
                            y, xq = yy * ct - xx * st + ch, yy * st + xx * ct + cw
                        else:
                            y, xq = yy, xx
                        acc = acc + _bilinear(x[b], y, xq)
                o[:, ph, pw] = acc / cnt
        outs.append(o)
    return torch.stack(outs) if outs else torch.zeros(0, x.shape[1], PH, PW)


def pairwise_iou(b1, b2):
    """SURVEY.md C.4."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1))


def matcher(quality, thresholds, labels, allow_low_quality):
    """detectron2 Matcher. quality (G,A). Returns matches (A,) int64, labels (A,) int8."""
    G, A = quality.shape
    if G == 0:
        return torch.zeros(A, dtype=torch.int64), torch.full((A,), labels[0], dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    vals, matches = quality.max(dim=0)
    out = torch.full((A,), 1, dtype=torch.int8)
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best = quality.max(dim=1).values
        idx = torch.nonzero(quality == best[:, None])[:, 1]
        out[idx] = 1
    return matches, out
