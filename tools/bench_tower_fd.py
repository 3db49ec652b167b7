"""Forward and data gradient of the head-tower convolution (5 FPN levels, 256 -> 256, 3x3, batch 16) on the library SOD_HIP_LIB selects: us per
launch (best of 5 x 10) and a checksum of the outputs (kernel variants that keep the MFMA order must print the same checksum).
    python tools/bench_tower_fd.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)
xs = [torch.randn(16, h, w, 256, device=dev).relu().bfloat16() for h, w in hws]
dys = [(torch.randn(16, h, w, 256, device=dev) * 1e-2).bfloat16() for h, w in hws]
wk, wt = HF.weight_prep(torch.randn(256, 3, 3, 256, device=dev) * 0.02)
bias = torch.randn(256, device=dev)
fns = {"fwd": lambda: HF.conv2d_fwd_ml(xs, wk, bias, 1, 1, 1, relu=True), "dgrad": lambda: HF.conv2d_dgrad_ml(dys, wt, hws, 1, 1, 1)}
out = []
for name, fn in fns.items():
    m = hashlib.sha1()
    for t in fn():
        m.update(t.float().cpu().numpy().tobytes())
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    out.append("%s %7.1f us %s" % (name, best * 1e3, m.hexdigest()[:8]))
print(" | ".join(out))
