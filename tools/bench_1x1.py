"""Stand-alone timing of the bottleneck 1x1 convolutions of FCOS R50 (batch 16, 800x1344) with and without their fused epilogue operands.
Every iteration uses another set of buffers from a pool larger than the Infinity Cache, so that operands come from HBM as in the step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from slenderobjdet_amd.layers import functional as HF

dev = torch.device("cuda:0")
N = 16
SHAPES = [("res3_conv3", 100, 168, 128, 512), ("res4_conv3", 50, 84, 256, 1024), ("res5_conv3", 25, 42, 512, 2048),
          ("res3_conv1", 100, 168, 512, 128), ("res4_conv1", 50, 84, 1024, 256)]
POOL = 6


def timeit(fns, iters=24):
    for f in fns:
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, H, W, C, K in SHAPES:
    xs = [torch.randn(N, H, W, C, device=dev).bfloat16() for _ in range(POOL)]
    rs = [torch.randn(N, H, W, K, device=dev).bfloat16() for _ in range(POOL)]
    w = (torch.randn(K, 1, 1, C, device=dev) * 0.05).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    bits = [torch.empty(N * H * W * K // 8, dtype=torch.uint8, device=dev) for _ in range(POOL)]
    px = N * H * W
    b_plain = px * (C + K) * 2
    rows = {}
    rows["fwd plain"] = (timeit([lambda x=x: HF.conv2d_fwd(x, w, None) for x in xs]), b_plain)
    rows["fwd +res+relu"] = (timeit([lambda x=x, r=r: HF.conv2d_fwd(x, w, None, r, relu=True) for x, r in zip(xs, rs)]), b_plain + px * K * 2)
    rows["fwd +res+relu+bits"] = (timeit([lambda x=x, r=r, b=b: HF.conv2d_fwd(x, w, None, r, relu=True, relu_bits=b) for x, r, b in zip(xs, rs, bits)]),
                                  b_plain + px * K * 2 + px * K // 8)
    # the data gradient of the mirrored conv (K -> C channels): dy (N,H,W,K... ) here: dy has C channels -> dx K channels, accumulate + mask bits
    dys = xs
    rows["dgrad plain (->K ch)"] = (timeit([lambda d=d: HF.conv2d_dgrad(d, w.permute(0, 1, 2, 3).reshape(K, 1, 1, C).permute(3, 1, 2, 0).contiguous().permute(3, 1, 2, 0).contiguous() if False else w.reshape(K, 1, 1, C), (H, W)) for d in dys]) if False else 0.0, 0)
    wd = w.reshape(K, 1, 1, C).contiguous()       # as CRSK of a conv with C_out = C, C_in = K: (C_in=K, 1, 1, K_out=C)
    rows.pop("dgrad plain (->K ch)")
    rows["dgrad ->wide plain"] = (timeit([lambda d=d: HF.conv2d_dgrad(d, wd, (H, W)) for d in dys]), b_plain)
    rows["dgrad ->wide +accum+bits"] = (timeit([lambda d=d, r=r, b=b: HF.conv2d_dgrad(d, wd, (H, W), accum=r, relu_bits=b) for d, r, b in zip(dys, rs, bits)]),
                                        b_plain + px * K * 2 + px * K // 8)
    fl = 2.0 * px * C * K
    for k, (us, by) in rows.items():
        print(f"{name:11s} {k:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:6.2f} TB/s ({by / 1e6:6.0f} MB)", flush=True)
