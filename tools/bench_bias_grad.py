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

# the multi-level launch (bias gradient of a conv shared by the five FPN levels)
for C in (256, 80):
    dys = [torch.randn(16, h, w, C, device=dev).bfloat16() for h, w in ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))]
    db = torch.zeros(C, device=dev)
    fn = lambda: HF.bias_grad_ml(dys, db)
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
    nb = sum(t.numel() for t in dys) * 2
    print(f"ml 5 levels C{C}: {best * 1e3:7.1f} us  {nb / best / 1e9:6.2f} TB/s", flush=True)
