"""Oracle for the reference's own native ops.
BorderAlign restates slender_det/layers/csrc/border_align/BorderAlign_cuda.cu:16-146 (a CUDA-only kernel, so it cannot run here; the
restatement is differentiable through autograd, which also gives the backward of :209-276).  CornerPool is what the reference itself
runs on torch >= 1.5: torch.cummax (layers/corner_pool.py:106-116)."""
import torch


def _bilinear(f, y, x):
    H, W = f.shape
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
    return hy * hx * f[yl, xl] + hy * lx * f[yl, xh] + ly * hx * f[yh, xl] + ly * lx * f[yh, xh]


def border_align(feature, boxes, pool_size):
    """feature (B,4C,H,W), boxes (B,K,4) -> (B,C,K,4)."""
    B, C4, H, W = feature.shape
    C, K = C4 // 4, boxes.shape[1]
    out = []
    for b in range(B):
        for c in range(C):
            for k in range(K):
                x1, y1, x2, y2 = [float(v) for v in boxes[b, k]]
                w, h = x2 - x1, y2 - y1
                for e in range(4):
                    x, y = (x1, y1) if e < 2 else (x2, y2)
                    xs, ys = [(w / pool_size, 0.0), (0.0, h / pool_size), (-w / pool_size, 0.0), (0.0, -h / pool_size)][e]
                    f = feature[b, e * C + c]
                    vals = []
                    for s in range(pool_size + 1):
                        vals.append(_bilinear(f, y, x))
                        x, y = x + xs, y + ys
                    v = torch.stack(vals)
                    out.append(v[int(torch.argmax(v))])     # first maximum, like the strict '>' of the kernel
    return torch.stack(out).view(B, C, K, 4)


def corner_pool(x, mode):
    dim, flip = {"bottom": (2, False), "left": (3, True), "right": (3, False), "top": (2, True)}[mode]
    if flip:
        x = x.flip(dim)
    y, _ = torch.cummax(x, dim=dim)
    return y.flip(dim) if flip else y
