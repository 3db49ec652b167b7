"""Weight-gradient micro-benchmark over EVERY launch of the FCOS R50-FPN step that the 256x256 kernel does not take (batch 16, 800 x 1344):
the 128x128 kernels of conv_igemm.hip (variant 0) and conv_wgrad_ring.hip (sod_conv_set_wgrad_variant codes), interleaved rounds in one
process, operands cycled through a pool larger than the Infinity Cache.

    python tools/bench_wgrad_backbone.py [variant ...]        # default: 0 2300 2310
Pseudo-variants: -1 = the dispatcher's own choice (what the step runs), 256 = the 256 x 256 kernel forced (splits = -1), 9 = the nine-tap
kernel forced (splits = -2); unsupported shapes print 0.0.
Prints one row per shape (us per launch, best of the rounds) and the count-weighted sum per training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd import _C  # noqa: E402
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
N = 16
# (H, W, C, K, R, stride, launches per step)
SHAPES = [
    (200, 336, 256, 128, 1, 2, 1), (100, 168, 128, 128, 3, 1, 4), (100, 168, 128, 512, 1, 1, 4), (200, 336, 256, 512, 1, 2, 1), (100, 168, 512, 128, 1, 1, 3),
    (100, 168, 512, 256, 1, 2, 1), (50, 84, 256, 256, 3, 1, 7), (50, 84, 256, 1024, 1, 1, 6), (100, 168, 512, 1024, 1, 2, 1), (50, 84, 1024, 256, 1, 1, 6),
    (50, 84, 1024, 512, 1, 2, 1), (25, 42, 512, 512, 3, 1, 3), (25, 42, 512, 2048, 1, 1, 3), (50, 84, 1024, 2048, 1, 2, 1), (25, 42, 2048, 512, 1, 1, 2),
    (100, 168, 512, 256, 1, 1, 1), (25, 42, 2048, 256, 1, 1, 1), (25, 42, 256, 256, 3, 1, 1), (25, 42, 256, 256, 3, 2, 1), (13, 21, 256, 256, 3, 2, 1),
    ("ml", 256, 80), ("ml", 256, 8),
]
POOL_BYTES = 600 << 20
SPLITS = [0]


def timeit(fn, iters):
    fn(0); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0, 2300, 2310]
    total = {v: 0.0 for v in variants}
    flops_total = 0.0
    print("shape".ljust(40) + "".join(f"{v:>10d}" for v in variants) + "   (us per launch; x count per step)")
    for shp in SHAPES:
        if shp[0] == "ml":
            _, C, K = shp
            hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
            nbytes = sum(N * h * w * (C + K) * 2 for h, w in hws)
            copies = max(2, POOL_BYTES // nbytes + 1)
            xs = [[torch.randn(N, h, w, C, device=dev).bfloat16() for h, w in hws] for _ in range(copies)]
            dys = [[torch.randn(N, h, w, K, device=dev).bfloat16() for h, w in hws] for _ in range(copies)]
            dw = torch.zeros(K, 3, 3, C, device=dev)
            flops = sum(2.0 * N * h * w * K * 9 * C for h, w in hws)
            fn = lambda i: HF.conv2d_wgrad_ml(dys[i % copies], xs[i % copies], dw, 3, 3, 1, 1, 1, splits=SPLITS[0])
            name, cnt = f"ml 5 levels C{C} K{K} R3", 1
        else:
            H, W, C, K, R, st, cnt = shp
            Ho, Wo = HF.conv_out_size(H, W, R, R, st, R // 2, 1)
            nbytes = N * (H * W * C + Ho * Wo * K) * 2
            copies = max(2, POOL_BYTES // nbytes + 1)
            x = [torch.randn(N, H, W, C, device=dev).bfloat16() for _ in range(copies)]
            dy = [torch.randn(N, Ho, Wo, K, device=dev).bfloat16() for _ in range(copies)]
            dw = torch.zeros(K, R, R, C, device=dev)
            flops = 2.0 * N * Ho * Wo * K * R * R * C
            fn = lambda i: HF.conv2d_wgrad(dy[i % copies], x[i % copies], dw, R, R, st, R // 2, 1, splits=SPLITS[0])
            name = f"{H}x{W} C{C} K{K} R{R} s{st}"
        best = {v: 1e9 for v in variants}
        for _ in range(3):
            for v in variants:
                _C.call("sod_conv_set_wgrad_variant", v if v not in (-1, 256, 9) else -1)
                SPLITS[0] = {256: -1, 9: -2}.get(v, 0)
                try:
                    best[v] = min(best[v], timeit(fn, 2 * copies if copies < 8 else copies))
                except _C.SlenderHipError:
                    best[v] = 0.0
        SPLITS[0] = 0
        _C.call("sod_conv_set_wgrad_variant", -1)
        for v in variants:
            total[v] += best[v] * cnt
        flops_total += flops * cnt
        print(f"{name:36s} x{cnt:<2d}" + "".join(f"{best[v] * 1e3:10.1f}" for v in variants) + f"   {flops / 1e9:7.1f} GFLOP", flush=True)
        del fn
        torch.cuda.empty_cache()
    print("sum per step (ms)".ljust(40) + "".join(f"{total[v]:10.3f}" for v in variants))
    print("TFLOP/s over the sum".ljust(40) + "".join(f"{flops_total / total[v] / 1e9:10.1f}" for v in variants))


if __name__ == "__main__":
    main()
