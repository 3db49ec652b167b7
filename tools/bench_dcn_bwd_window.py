"""The DeformConv backward's data-dependent cliff (csrc/deform_conv.hip: dcn_bwd_fused_kernel accumulates dX in an LDS window around each
8x8 output tile; samples that land outside it take 32 global float atomics each): time of the fused input / offset gradient and the
share of sample lanes that left the window, against the spread of the offsets - RepPoints' P3 level at batch 16 (rpd.py:637-647).

    python tools/bench_dcn_bwd_window.py            # offsets ~ N(0, s), s = 0.5, 1, 2, 4, 8, 16 px"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    N, H, W, C, K = 16, 100, 168, 256, 256
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device=dev).relu().bfloat16()
    dy = (torch.randn(N, H, W, K, device=dev) * 0.1).bfloat16()
    w = (torch.randn(K, 1, 1, 9 * C, device=dev) * 0.02)
    _, wt = HF.weight_prep(w)
    lanes = N * H * W * 9 * (C // 8)
    print(f"P3 level of RepPoints at batch 16: {N}x{H}x{W}, {C} -> {K}, 3x3; {lanes / 1e6:.1f} M sample lanes; window slack = the library default (2 px) unless a row says otherwise")
    for std in (0.5, 1.0, 2.0, 4.0, 8.0, 16.0):
        off = torch.randn(N, H, W, 18, device=dev) * std
        doff = torch.zeros_like(off)
        run = lambda: HF.deform_conv_bwd_fused(dy, wt, x, off, None, (3, 3), 1, 1, 1, 1, doff, None)
        line = f"offset std {std:5.1f} px:"
        for slack in (2, 4, 6):       # what layers/deform_conv.py::_WindowPolicy chooses among
            HF.call("sod_deform_conv_set_window_slack", slack)
            with HF.DeformWindowCounter(dev) as c:
                run()
                out = c.read()
            t = timeit(run)
            line += f"   slack {slack}: {t * 1e3:8.1f} us, {out / lanes * 100:5.2f} % outside"
        HF.call("sod_deform_conv_set_window_slack", -1)
        print(line, flush=True)


if __name__ == "__main__":
    main()
