"""A few launches of one weight-gradient kernel on the head-tower shape, for `rocprofv3 --kernel-trace --pmc ...` (counters per dispatch).
    python tools/w9_probe.py [-2 | -1 | fwd | dgrad]        # -2 nine-tap (default), -1 the 256 x 256 weight gradient, fwd / dgrad = conv_igemm256"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "-2"
dev = torch.device("cuda:0")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)
xs = [torch.randn(16, h, w, 256, device=dev).relu().bfloat16() for h, w in hws]
dys = [(torch.randn(16, h, w, 256, device=dev) * 1e-2).bfloat16() for h, w in hws]
dw = torch.zeros(256, 3, 3, 256, device=dev)
wk, wt = HF.weight_prep(torch.randn(256, 3, 3, 256, device=dev) * 0.02)
bias = torch.randn(256, device=dev)
for _ in range(6):
    if mode == "fwd":
        HF.conv2d_fwd_ml(xs, wk, bias, 1, 1, 1, relu=True)
    elif mode == "dgrad":
        HF.conv2d_dgrad_ml(dys, wt, hws, 1, 1, 1)
    else:
        HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=int(mode))
torch.cuda.synchronize()
print("done")
