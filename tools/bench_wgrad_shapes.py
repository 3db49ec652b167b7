"""wgrad micro-benchmark over the FCOS R50 shapes: python tools/bench_wgrad_shapes.py  (SOD_WGRAD_PLAIN=1: racy plain stores, timing only)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(16, 100, 168, 256, 256, 3), (16, 50, 84, 256, 256, 3), (16, 100, 168, 128, 128, 3), (16, 50, 84, 256, 1024, 1), (16, 50, 84, 1024, 256, 1),
          (16, 100, 168, 128, 512, 1), (16, 100, 168, 512, 128, 1), (16, 25, 42, 512, 512, 3), (16, 25, 42, 512, 2048, 1), (16, 25, 42, 2048, 512, 1)]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (N, H, W, C, K, R) in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    dy = torch.randn(N, H, W, K, device=dev).bfloat16()
    dw = torch.zeros(K, R, R, C, device=dev)
    flops = 2.0 * N * H * W * K * R * R * C
    t = timeit(lambda: HF.conv2d_wgrad(dy, x, dw, R, R, 1, R // 2, 1))
    print(f"wgrad N{N} {H}x{W} C{C} K{K} R{R}: {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TF")
