"""The expand + contract pair kernel (csrc/bneck_pair.hip) against the two launches it replaces, on the res3 / res4 shapes of FCOS R50 at
batch 16 (operands cycled through a pool larger than the Infinity Cache):  python tools/bench_pair.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
N = 16


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


for name, H, W, CN, CW in (("res3", 100, 168, 128, 512), ("res4", 50, 84, 256, 1024)):
    P = N * H * W
    pool = max(3, (700 << 20) // (P * (2 * CN + 2 * CW) * 2) + 1)
    xs = [torch.randn(N, H, W, CN, device=dev).relu().bfloat16() for _ in range(pool)]
    adds = [torch.randn(N, H, W, CW, device=dev).bfloat16() for _ in range(pool)]
    masks = [torch.randn(N, H, W, CN, device=dev).bfloat16() for _ in range(pool)]
    we, we_t = HF.weight_prep(torch.randn(CW, 1, 1, CN, device=dev) * 0.05)
    wc, wc_t = HF.weight_prep(torch.randn(CN, 1, 1, CW, device=dev) * 0.03)
    be, bc = torch.randn(CW, device=dev), torch.randn(CN, device=dev)
    bits = [torch.empty(P * CW // 8, dtype=torch.uint8, device=dev) for _ in range(pool)]

    def two_fwd(i):
        y = HF.conv2d_fwd(xs[i], we, be, adds[i], relu=True, relu_bits=bits[i])
        return HF.conv2d_fwd(y, wc, bc, relu=True)

    def two_bwd(i):
        g = HF.conv2d_dgrad(xs[i], wc_t, (H, W), accum=adds[i], relu_bits=bits[i])
        return HF.conv2d_dgrad(g, we_t, (H, W), relu_mask=masks[i])

    for i in range(pool):
        two_fwd(i)          # fills the bit arrays
    fl = 4.0 * P * CN * CW
    by = P * (2 * CN + 2 * CW) * 2
    rows = [("fwd  two launches", timeit([lambda i=i: two_fwd(i) for i in range(pool)])),
            ("fwd  pair kernel", timeit([lambda i=i: HF.bottleneck_pair(xs[i], adds[i], we, be, wc, bc, 0) for i in range(pool)])),
            ("bwd  two launches", timeit([lambda i=i: two_bwd(i) for i in range(pool)])),
            ("bwd  pair kernel", timeit([lambda i=i: HF.bottleneck_pair(xs[i], adds[i], wc_t, None, we_t, None, 1, bits_in=bits[i], mask2=masks[i])
                                         for i in range(pool)]))]
    for k, us in rows:
        print(f"{name} {CN}->{CW}->{CN} {k:18s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s (narrow in + add in + wide out + narrow out)", flush=True)
