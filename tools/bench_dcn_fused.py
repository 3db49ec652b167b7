"""DeformConv forward: column-buffer path (gather + 1x1 GEMM) vs the fused kernel on the RepPoints shapes: python tools/bench_dcn_fused.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (N, H, W) in [(16, 100, 168), (16, 50, 84), (16, 25, 42), (16, 13, 21)]:
    C = K = 256
    x = torch.randn(N, H, W, C, device=dev).relu().bfloat16()
    off = (torch.rand(N, H, W, 18, device=dev) - 0.5) * 6
    w = (torch.randn(K, 1, 1, 9 * C, device=dev) * 0.02).bfloat16()
    flops = 2.0 * N * H * W * K * 9 * C
    res = {"cols": [], "fused": []}
    for _ in range(3):
        res["cols"].append(timeit(lambda: HF.conv2d_fwd(HF.deform_im2col(x, off, None, (3, 3), 1, 1, 1), w, None, stride=1, pad=0)))
        res["fused"].append(timeit(lambda: HF.deform_conv_fwd_fused(x, off, None, w, None, (3, 3), 1, 1, 1)))
    dy = torch.randn(N, H, W, K, device=dev).bfloat16()
    dw = torch.zeros(K, 1, 1, 9 * C, device=dev)
    cols = HF.deform_im2col(x, off, None, (3, 3), 1, 1, 1)
    wres = {"wgrad on cols": [], "fused": [], "gather+wgrad": []}
    for _ in range(3):
        wres["wgrad on cols"].append(timeit(lambda: HF.conv2d_wgrad(dy, cols, dw, 1, 1, 1, 0, 1)))
        wres["gather+wgrad"].append(timeit(lambda: HF.conv2d_wgrad(dy, HF.deform_im2col(x, off, None, (3, 3), 1, 1, 1), dw, 1, 1, 1, 0, 1)))
        wres["fused"].append(timeit(lambda: HF.deform_conv_wgrad_fused(dy, x, off, None, dw, (3, 3), 1, 1, 1)))
    print(f"dcn wgrad N{N} {H}x{W}: " + " | ".join(f"{k}: {min(v) * 1e3:7.1f} us {flops / min(v) / 1e9:6.1f} TF" for k, v in wres.items()), flush=True)
    print(f"dcn fwd N{N} {H}x{W} 256->256: " + " | ".join(f"{k}: {min(v) * 1e3:7.1f} us {flops / min(v) / 1e9:6.1f} TF" for k, v in res.items()), flush=True)
