"""wgrad micro-benchmark over the FCOS R50 shapes, 128x128 kernel (atomics) vs 256x256 kernel (slabs), interleaved rounds in one
process: python tools/bench_wgrad_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
# (N, H, W, C, K, R, stride)
SHAPES = [("head ml", None), (16, 100, 168, 256, 256, 3, 1), (16, 50, 84, 256, 256, 3, 1), (16, 50, 84, 256, 1024, 1, 1), (16, 50, 84, 1024, 256, 1, 1),
          (16, 25, 42, 512, 512, 3, 1), (16, 25, 42, 512, 2048, 1, 1), (16, 25, 42, 2048, 512, 1, 1), (16, 100, 168, 512, 256, 1, 1),
          (16, 50, 84, 1024, 2048, 1, 2), (16, 25, 42, 2048, 256, 1, 1), (16, 100, 168, 512, 1024, 1, 2),
          (16, 100, 168, 128, 128, 3, 1), (16, 100, 168, 128, 512, 1, 1), (16, 100, 168, 512, 128, 1, 1)]


def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for shp in SHAPES:
    if shp[0] == "head ml":
        hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
        xs = [torch.randn(16, h, w, 256, device=dev).bfloat16() for h, w in hws]
        dys = [torch.randn(16, h, w, 256, device=dev).bfloat16() for h, w in hws]
        dw = torch.zeros(256, 3, 3, 256, device=dev)
        flops = sum(2.0 * 16 * h * w * 256 * 9 * 256 for h, w in hws)
        # 128x128 kernel: 36 tiles x 3 workgroups per CU -> 21 splits (launch_wgrad's default heuristic)
        fns = {"128": lambda: HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=21),
               "256": lambda: HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=-1)}
        name = "head 5 levels 256->256 3x3"
    else:
        N, H, W, C, K, R, st = shp
        Ho, Wo = HF.conv_out_size(H, W, R, R, st, R // 2, 1)
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        dy = torch.randn(N, Ho, Wo, K, device=dev).bfloat16()
        dw = torch.zeros(K, R, R, C, device=dev)
        flops = 2.0 * N * Ho * Wo * K * R * R * C
        tiles = ((K + 127) // 128) * ((C + 127) // 128) * R * R
        sp = max(1, ((3 if tiles >= 36 else 2) * 256) // tiles)       # launch_wgrad's default heuristic
        fns = {"128": lambda: HF.conv2d_wgrad(dy, x, dw, R, R, st, R // 2, 1, splits=sp)}
        ok256 = K % 256 == 0 and C % 256 == 0
        if ok256:
            fns["256"] = lambda: HF.conv2d_wgrad(dy, x, dw, R, R, st, R // 2, 1, splits=-1)
        name = f"N{N} {H}x{W} C{C} K{K} R{R} s{st}"
    res = {k: [] for k in fns}
    for _ in range(3):
        for k, fn in fns.items():
            res[k].append(timeit(fn))
    line = f"wgrad {name:34s}"
    for k, v in res.items():
        t = min(v)
        line += f" | {k}: {t * 1e3:7.1f} us {flops / t / 1e9:7.1f} TF"
    print(line, flush=True)
