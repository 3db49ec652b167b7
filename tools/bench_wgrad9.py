"""conv_wgrad9.hip (nine taps in one workgroup) against the kernels it replaces, per 3x3 shape of the FCOS R50 step at batch 16:
splits = -2 forces the nine-tap kernel, -1 the 256 x 256 kernel (where supported), a positive count the 128 x 128 kernel.  Prints us per launch, algorithmic TFLOP/s and the largest
relative difference of the result to the 128 x 128 kernel's.    python tools/bench_wgrad9.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd import _C                       # noqa: E402
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [("head towers, 5 levels 256->256", 256, 256, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]),
          ("P3 output 100x168 256->256", 256, 256, [(100, 168)]),
          ("res4 conv2 50x84 256->256", 256, 256, [(50, 84)]),
          ("res3 conv2 100x168 128->128", 128, 128, [(100, 168)]),
          ("res5 conv2 25x42 512->512", 512, 512, [(25, 42)]),
          ("P5 output 25x42 256->256", 256, 256, [(25, 42)])]


def timeit(fn):
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
    return best * 1e3


for name, C, K, hws in SHAPES:
    xs = [torch.randn(16, h, w, C, device=dev).relu().bfloat16() for h, w in hws]
    dys = [(torch.randn(16, h, w, K, device=dev) * 1e-2).bfloat16() for h, w in hws]
    flops = sum(2.0 * 16 * h * w * K * 9 * C for h, w in hws)
    res, line = {}, []
    for label, sp in (("128x128", 0), ("256x256", -1), ("nine-tap", -2)):
        dw = torch.zeros(K, 3, 3, C, device=dev)
        if sp == 0:         # the 128 x 128 kernel with the split count its dispatcher path takes: one resident wave of blocks, two per CU
            sp = max(1, 512 // ((K // 128) * (C // 128) * 9))
        fn = lambda: HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=sp)
        try:
            fn()
            torch.cuda.synchronize()
        except _C.SlenderHipError:
            line.append(f"{label}: unsupported")
            continue
        res[label] = dw.clone()
        us = timeit(fn)
        line.append(f"{label}: {us:7.1f} us {flops / us / 1e6:7.1f} TF/s")
    if "nine-tap" in res and "128x128" in res:
        d = (res["nine-tap"] - res["128x128"]).abs().max().item() / res["128x128"].abs().max().item()
        line.append(f"max rel diff {d:.1e}")
    print(f"{name:34s} " + " | ".join(line), flush=True)
