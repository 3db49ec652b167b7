"""The small-grid convolutions of the FPN top (P5 output, P6, P7: 3x3, 256 -> 256, stride 1 / 2 on 25x42 ... 13x21 maps, batch 16):
forward and data gradient, us per launch.   python tools/bench_small_convs.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
N = 16


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best * 1e3


for (H, W, C, K, R, st) in [(25, 42, 256, 256, 3, 1), (25, 42, 256, 256, 3, 2), (13, 21, 256, 256, 3, 2), (25, 42, 2048, 256, 1, 1), (50, 84, 256, 256, 3, 1)]:
    pad = R // 2
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, 1)
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    w = (torch.randn(K, R, R, C, device=dev) * 0.05)
    wk, wt = HF.weight_prep(w)
    b = torch.zeros(K, device=dev)
    dy = torch.randn(N, Ho, Wo, K, device=dev).bfloat16()
    tf = timeit(lambda: HF.conv2d_fwd(x, wk, b, None, st, pad, 1))
    td = timeit(lambda: HF.conv2d_dgrad(dy, wt, (H, W), st, pad, 1))
    fl = 2.0 * N * Ho * Wo * K * R * R * C
    print(f"{H}x{W} C{C} K{K} R{R} s{st}: fwd {tf:6.1f} us ({fl / tf / 1e6:6.1f} TF)  dgrad {td:6.1f} us ({fl / td / 1e6:6.1f} TF)", flush=True)
