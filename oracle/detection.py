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


def roi_align_vec(x, rois, output_size, scale, sampling_ratio=0, rotated=False):
    """roi_align() with the sample points of one ROI evaluated at once (torch ops over a (PH * gh, PW * gw) grid instead of four nested
    Python loops): the same sample positions, validity rules and bilinear weights (formed in float64 like the loop's Python floats, applied
    to the features in their own dtype), summed per bin in a different order.  For the CPU BASELINE of bench.py only (the benchmark's
    map-sized random-init proposals have up to ~1 500 samples per bin: the loop needs minutes per image); the parity tests keep the loop.
    tests/test_oracle_crosscheck.py pins the two to each other."""
    PH, PW = output_size
    N, C, H, W = x.shape
    flat = x.permute(0, 2, 3, 1).reshape(N, H * W, C)          # rows of C values: a corner gather moves contiguous rows
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
        if gh <= 0 or gw <= 0:
            outs.append(torch.zeros(C, PH, PW, dtype=x.dtype))
            continue
        ph = torch.arange(PH, dtype=torch.float64).repeat_interleave(gh)
        iy = torch.arange(gh, dtype=torch.float64).repeat(PH)
        pw = torch.arange(PW, dtype=torch.float64).repeat_interleave(gw)
        ix = torch.arange(gw, dtype=torch.float64).repeat(PW)
        yy = (sh + ph * bh + (iy + 0.5) * bh / gh)[:, None]      # (PH * gh, 1)
        xx = (sw + pw * bw + (ix + 0.5) * bw / gw)[None, :]      # (1, PW * gw)
        if rotated:
            y, xq = yy * ct - xx * st + ch, yy * st + xx * ct + cw
        else:
            y, xq = yy.expand(-1, PW * gw), xx.expand(PH * gh, -1)
        ok = ~((y < -1.0) | (y > H) | (xq < -1.0) | (xq > W))
        y, xq = y.clamp(min=0.0), xq.clamp(min=0.0)
        yl, xl = y.floor().long(), xq.floor().long()
        top, right = yl >= H - 1, xl >= W - 1
        yl, xl = torch.where(top, torch.full_like(yl, H - 1), yl), torch.where(right, torch.full_like(xl, W - 1), xl)
        yh, xh = torch.where(top, yl, yl + 1), torch.where(right, xl, xl + 1)
        y, xq = torch.where(top, yl.double(), y), torch.where(right, xl.double(), xq)
        ly, lx = y - yl, xq - xl
        hy, hx = 1.0 - ly, 1.0 - lx
        okf = ok.double()
        f = flat[b]
        acc = None
        for wgt, yi, xi in ((hy * hx, yl, xl), (hy * lx, yl, xh), (ly * hx, yh, xl), (ly * lx, yh, xh)):
            v = f.index_select(0, (yi * W + xi).reshape(-1)) * (wgt * okf).reshape(-1, 1).to(x.dtype)
            acc = v if acc is None else acc + v
        acc = acc.reshape(PH, gh, PW, gw, C).sum(dim=(1, 3)) / max(gh * gw, 1)
        outs.append(acc.permute(2, 0, 1))
    return torch.stack(outs) if outs else torch.zeros(0, C, PH, PW, dtype=x.dtype)


ROI_ALIGN_IMPL = "loop"      # "vec": roi_pool() of oracle/rcnn.py takes roi_align_vec (bench.py's cpu_baseline)


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


def topk_matcher(quality, thresholds, labels, topk):
    """slender_det/modeling/matchers/topk_matcher.py:38-85 (reference's own code; pinned by tests/golden/topk_matcher.npz)."""
    G, A = quality.shape
    if quality.numel() == 0:
        return torch.zeros(A, dtype=torch.int64), torch.full((A,), labels[0], dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    vals, matches = quality.max(dim=0)
    out = torch.full((A,), 1, dtype=torch.int8)
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    out[quality.topk(k=topk, dim=1)[1]] = 1
    return matches, out


# ------------------------------------------------------------------------------------------------ rotated boxes (C.15)
def _rot_vertices(b):
    cx, cy, w, h, a = [np.float32(v) for v in b]
    th = np.float32(a * np.float32(0.01745329251994329577))
    c2, s2 = np.float32(np.cos(th) * np.float32(0.5)), np.float32(np.sin(th) * np.float32(0.5))
    p0 = (cx + s2 * h + c2 * w, cy + c2 * h - s2 * w)
    p1 = (cx - s2 * h + c2 * w, cy - c2 * h - s2 * w)
    return np.array([p0, p1, (2 * cx - p0[0], 2 * cy - p0[1]), (2 * cx - p1[0], 2 * cy - p1[1])], dtype=np.float32)


def _cross(a, b):
    return np.float32(a[0] * b[1] - b[0] * a[1])


def _dot(a, b):
    return np.float32(a[0] * b[0] + a[1] * b[1])


def box_iou_rotated_single(b1, b2):
    """detectron2 single_box_iou_rotated: polygon clipping by edge intersections + contained vertices, Graham hull, shoelace."""
    a1, a2 = np.float32(b1[2] * b1[3]), np.float32(b2[2] * b2[3])
    if a1 < 1e-14 or a2 < 1e-14:
        return np.float32(0)
    sx, sy = np.float32((b1[0] + b2[0]) / 2), np.float32((b1[1] + b2[1]) / 2)
    p1 = _rot_vertices((b1[0] - sx, b1[1] - sy, b1[2], b1[3], b1[4]))
    p2 = _rot_vertices((b2[0] - sx, b2[1] - sy, b2[2], b2[3], b2[4]))
    v1 = [p1[(i + 1) % 4] - p1[i] for i in range(4)]
    v2 = [p2[(i + 1) % 4] - p2[i] for i in range(4)]
    pts = []
    for i in range(4):
        for j in range(4):
            det = _cross(v2[j], v1[i])
            if abs(det) <= 1e-14:
                continue
            v12 = p2[j] - p1[i]
            t1, t2 = np.float32(_cross(v2[j], v12) / det), np.float32(_cross(v1[i], v12) / det)
            if 0 <= t1 <= 1 and 0 <= t2 <= 1:
                pts.append(p1[i] + v1[i] * t1)
    for (pa, pb, vb) in ((p1, p2, v2), (p2, p1, v1)):
        AB, DA = vb[0], vb[3]
        abab, adad = _dot(AB, AB), _dot(DA, DA)
        for i in range(4):
            AP = pa[i] - pb[0]
            apab, apad = _dot(AP, AB), -_dot(AP, DA)
            if apab >= 0 and apad >= 0 and apab <= abab and apad <= adad:
                pts.append(pa[i])
    if len(pts) <= 2:
        return np.float32(0)
    pts = np.array(pts, dtype=np.float32)
    t = 0
    for i in range(1, len(pts)):
        if pts[i][1] < pts[t][1] or (pts[i][1] == pts[t][1] and pts[i][0] < pts[t][0]):
            t = i
    q = pts - pts[t]
    q[[0, t]] = q[[t, 0]]
    n = len(q)
    for i in range(1, n - 1):
        for j in range(1, n - i):
            c = _cross(q[j], q[j + 1])
            if c < -1e-6 or (abs(c) < 1e-6 and _dot(q[j], q[j]) > _dot(q[j + 1], q[j + 1])):
                q[[j, j + 1]] = q[[j + 1, j]]
    k = 1
    while k < n and _dot(q[k], q[k]) <= 1e-8:
        k += 1
    if k == n:
        return np.float32(0)
    hull = [q[0], q[k]]
    for i in range(k + 1, n):
        while len(hull) > 1 and _cross(q[i] - hull[-2], hull[-1] - hull[-2]) >= 0:
            hull.pop()
        hull.append(q[i])
    if len(hull) <= 2:
        return np.float32(0)
    area = np.float32(0)
    for i in range(1, len(hull) - 1):
        area += abs(_cross(hull[i] - hull[0], hull[i + 1] - hull[0]))
    inter = np.float32(area / 2)
    return np.float32(inter / (a1 + a2 - inter))


def pairwise_iou_rotated(b1, b2):
    b1, b2 = b1.numpy().astype(np.float32), b2.numpy().astype(np.float32)
    out = np.zeros((len(b1), len(b2)), dtype=np.float32)
    with np.errstate(all="ignore"):
        for i in range(len(b1)):
            for j in range(len(b2)):
                out[i, j] = box_iou_rotated_single(b1[i], b2[j])
    return torch.from_numpy(out)


def nms_rotated(boxes, scores, thr):
    iou = pairwise_iou_rotated(boxes, boxes).numpy()
    order = torch.sort(scores, descending=True, stable=True).indices.numpy()
    keep, dead = [], np.zeros(len(boxes), dtype=bool)
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(int(i))
        for j in order[pos + 1:]:
            if iou[i, j] > np.float32(thr):
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64)
