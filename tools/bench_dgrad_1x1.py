"""Data gradient of the contracting / expanding 1x1 convolutions of res3 / res4 (batch 16) on the 128x128 and the 256x256 kernel, with the
epilogue operands of the step (ReLU mask tensor / accumulate + bit mask).   python tools/bench_dgrad_1x1.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd import _C  # noqa: E402
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
N = 16


def timeit(fn, iters=20):
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


for (H, W, C, K) in [(50, 84, 256, 1024), (100, 168, 128, 512), (25, 42, 512, 2048), (50, 84, 1024, 256), (100, 168, 512, 128)]:
    # conv with C inputs and K outputs: dgrad contracts over K and writes C channels
    dy = torch.randn(N, H, W, K, device=dev).bfloat16()
    w = torch.randn(K, 1, 1, C, device=dev) * 0.05
    _, wt = HF.weight_prep(w)
    mask = torch.randn(N, H, W, C, device=dev).relu().bfloat16()
    row = f"{H}x{W} dY {K} ch -> dX {C} ch:"
    for mode in (0, 1, 2):
        HF.call("sod_conv_set_tile256", mode)
        try:
            t = timeit(lambda: HF.conv2d_dgrad(dy, wt, (H, W), 1, 0, 1, relu_mask=mask))
            v = _C.load().sod_conv_last_variant()
        except Exception as exc:      # noqa: BLE001
            t, v = float("nan"), str(exc)[:20]
        row += f"  tile256={mode}: {t:6.1f} us (variant {v})"
    HF.call("sod_conv_set_tile256", -1)
    print(row, flush=True)
