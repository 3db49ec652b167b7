"""How much of a 256x256 tile's life of the tower convolution is its K loop: the P3-sized 3x3 convolution (16 x 100 x 168 pixels = 1050 tiles,
256 output channels) timed with 128 / 256 / 512 / 1024 input channels (18 / 36 / 72 / 144 K-tiles of 64).  The slope is the K loop, the
intercept the prologue + epilogue + launch; SOD_CONV256=2 forces the 256x256 kernel for every width."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from slenderobjdet_amd.layers import functional as HF

dev = torch.device("cuda:0")
N, H, W, K = 16, 100, 168, 256
POOL = 4


def timeit(fns, iters=16):
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


res = []
for C in (128, 256, 512, 1024):
    xs = [torch.randn(N, H, W, C, device=dev).bfloat16() for _ in range(POOL)]
    w = (torch.randn(K, 3, 3, C, device=dev) * 0.02).bfloat16()
    b = torch.zeros(K, device=dev)
    us = timeit([lambda x=x: HF.conv2d_fwd(x, w, b, None, 1, 1, 1) for x in xs])
    T = 9 * C // 64
    fl = 2.0 * N * H * W * K * 9 * C
    res.append((T, us))
    print(f"C={C:5d}  K-tiles {T:4d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  variant {HF.last_conv_variant() if hasattr(HF, 'last_conv_variant') else '?'}", flush=True)
(t0, u0), (t1, u1) = res[1], res[3]
slope = (u1 - u0) / (t1 - t0)
print(f"slope {slope:.3f} us per K-tile of the whole launch; intercept at 36 K-tiles: {u0 - 36 * slope:.1f} us of {u0:.1f} ({(u0 - 36 * slope) / u0 * 100:.1f} %)")
