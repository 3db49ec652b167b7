"""Bias gradient (channel sums of dY) on the tensor sizes of the steps: us per launch and effective read bandwidth.
python tools/bench_bias_grad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
for (N, HW, C) in [(16, 200 * 336, 256), (16, 100 * 168, 256), (16, 22400, 80), (16, 22400, 8), (16, 50 * 84, 256), (16, 201600 // 9, 720)]:
    dy = torch.randn(N, HW, C, device=dev).bfloat16()
    db = torch.zeros(C, device=dev)
    fn = lambda: HF.bias_grad(dy, db, N, HW, C)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    ref = dy.float().sum((0, 1))
    db.zero_(); fn(); torch.cuda.synchronize()
    err = float((db - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    print(f"N{N} HW{HW} C{C}: {best * 1e3:7.1f} us  {dy.numel() * 2 / best / 1e9:6.2f} TB/s  rel err {err:.1e}", flush=True)
