"""conv_ws3.hip (persistent weight-stationary 3x3, 128 -> 128) against the tiled kernel on the res3 conv2 geometry (16 x 100 x 168), forward and
data gradient, operands cycled through a pool larger than the Infinity Cache.    python tools/bench_ws3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd import _C                       # noqa: E402
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W, C = 16, 100, 168, 128
copies = 6
xs = [torch.randn(N, H, W, C, device=dev).relu().bfloat16() for _ in range(copies)]
dys = [(torch.randn(N, H, W, C, device=dev) * 1e-2).bfloat16() for _ in range(copies)]
w = torch.randn(C, 3, 3, C, device=dev) * (9 * C) ** -0.5
wk, wt = HF.weight_prep(w)
bias = torch.randn(C, device=dev)
flops = 2.0 * N * H * W * C * 9 * C


def timeit(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(12):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 12)
    return best * 1e3


for mode, name in ((0, "tiled 128x128"), (1, "weight-stationary"), (0, "tiled 128x128"), (1, "weight-stationary")):
    _C.call("sod_conv_set_ws3", mode)
    f = timeit(lambda i: HF.conv2d_fwd(xs[i % copies], wk, bias, stride=1, pad=1, relu=True))
    vf = int(_C.load().sod_conv_last_variant())
    b = timeit(lambda i: HF.conv2d_dgrad(dys[i % copies], wt, (H, W), 1, 1, 1, relu_mask=xs[i % copies]))
    vb = int(_C.load().sod_conv_last_variant())
    print(f"{name:20s} fwd {f:7.1f} us {flops / f / 1e6:7.1f} TF/s (variant {vf}) | dgrad {b:7.1f} us {flops / b / 1e6:7.1f} TF/s (variant {vb})", flush=True)
_C.call("sod_conv_set_ws3", -1)
