#!/usr/bin/env python
"""Stand-alone timing of the fused frozen bottleneck kernel (csrc/bottleneck_fused.hip) against the un-fused launches, res2 of R50 on a
16 x 200x336 batch: per-block time, algorithmic TFLOP/s and effective HBM rate (x in + out out)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_cfg  # noqa: E402
from slenderobjdet_amd.modeling import build_model  # noqa: E402
from slenderobjdet_amd.modeling.backbone import resnet  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    N, H, W = 16, 200, 336
    model = build_model(make_cfg(50))
    scale = 1.0
    if "--damp" in sys.argv[1:]:      # the benchmark's RetinaNet / R-CNN conditioning (bench.damp_residual_branches): conv3 gamma 0.25, stem 1/64,
        from bench import damp_residual_branches      # and activations of the size a damped stem hands on - are small operands slower? (no)
        damp_residual_branches(model)
        scale = 1.0 / 64
        print("damped: last FrozenBN weight of every block 0.25, inputs scaled by 1/64")
    stage = model.backbone.bottom_up.res2
    x64 = (torch.randn(N, H, W, 64, device=dev) * scale).to(torch.bfloat16)
    x256 = (torch.randn(N, H, W, 256, device=dev) * scale).to(torch.bfloat16).relu()
    with torch.no_grad():
        for name, blk, x in (("proj 64->256", stage[0], x64), ("identity 256", stage[1], x256)):
            cin = x.shape[-1]
            macs = cin * 64 + 9 * 64 * 64 + 64 * 256 + (cin * 256 if blk.shortcut is not None else 0)
            fl = 2.0 * N * H * W * macs
            by = N * H * W * (cin + 256) * 2
            t_f = timeit(lambda: resnet._fused_frozen_block(blk, x))
            t_u = timeit(lambda: blk(x))
            print(f"{name:14s} fused {t_f * 1e3:7.1f} us ({fl / t_f / 1e9:6.1f} TFLOP/s, {by / t_f / 1e9:5.2f} TB/s in+out)   un-fused {t_u * 1e3:7.1f} us")
        resnet.BNECK_FUSED = True
        t_f = timeit(lambda: stage(x64))
        resnet.BNECK_FUSED = False
        t_u = timeit(lambda: stage(x64))
        print(f"res2 stage     fused {t_f * 1e3:7.1f} us   un-fused {t_u * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
