"""Stand-alone timing of sod_anchor_match(_rotated) on the RRPN anchor set of BASELINE configs[4] (1.6 M rotated anchors, 7 boxes)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import rcnn as orc  # noqa: E402  (anchor grid only; timing tool, not the product path)
from slenderobjdet_amd.layers import functional as HF  # noqa: E402


def main():
    dev = torch.device("cuda")
    hw = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    anchors = torch.cat(orc.anchors(hw, [4, 8, 16, 32, 64], [[32], [64], [128], [256], [512]], [[0.5, 1.0, 2.0]], [[-90, -60, -30, 0, 30, 60]])).to(dev)
    g = torch.Generator().manual_seed(0)
    G = 7
    cx, cy = torch.rand(G, generator=g) * 1333, torch.rand(G, generator=g) * 800
    w, h = 40 + torch.rand(G, generator=g) * 300, 20 + torch.rand(G, generator=g) * 200
    ang = torch.rand(G, generator=g) * 180 - 90
    gt = torch.stack((cx, cy, w, h, ang), 1).to(dev).contiguous()
    print("anchors", tuple(anchors.shape))
    for _ in range(3):
        out = HF.anchor_match(gt, anchors, [0.3, 0.7], [0, -1, 1], True)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        out = HF.anchor_match(gt, anchors, [0.3, 0.7], [0, -1, 1], True)
    torch.cuda.synchronize()
    print(f"anchor_match rotated: {(time.time() - t) / 20 * 1e6:.1f} us per call; positives {int((out[2] == 1).sum())}, ignored {int((out[2] == -1).sum())}")
    far = gt.clone()
    far[:, 0] += 1.0e5           # no pair passes the circle test: the floor of the two kernels (loads, loops, stores)
    for _ in range(3):
        HF.anchor_match(far, anchors, [0.3, 0.7], [0, -1, 1], True)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        HF.anchor_match(far, anchors, [0.3, 0.7], [0, -1, 1], True)
    torch.cuda.synchronize()
    print(f"  with all boxes far away (no polygon clipping at all): {(time.time() - t) / 20 * 1e6:.1f} us per call")
    near = int((HF.box_iou_rotated(gt, anchors) > 0).sum())
    print(f"  pairs with IoU > 0: {near} of {G * anchors.shape[0]}")


if __name__ == "__main__":
    main()
