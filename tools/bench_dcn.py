"""Micro-benchmark of the deformable-conv gather kernels at the RepPoints level shapes: python tools/bench_dcn.py [--spread S]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spread", type=float, default=0.0, help="std of the learned point offsets (0: all taps sample the centre, as at init)")
    ap.add_argument("--n", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for (h, w) in ((100, 168), (50, 84), (25, 42)):
        x = torch.randn((a.n, h, w, 256), device=dev).bfloat16()
        pts = torch.randn((a.n, h, w, 24), device=dev) * a.spread
        pts[..., 18:] = 0
        off = HF.reppoints_dcn_offset(pts, 9, 1.0, True)
        dcols = torch.randn((a.n, h, w, 9 * 256), device=dev).bfloat16()
        for name, fn in (("im2col", lambda: HF.deform_im2col(x, off, None, (3, 3), 1, 1, 1, 1, 24)),
                         ("col2im", lambda: HF.deform_col2im(dcols, x, off, None, (3, 3), 1, 1, 1, 1, torch.zeros_like(off), None, 24))):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            items = a.n * h * w * 9 * 256
            print(f"{name} {a.n}x{h}x{w}x256 spread {a.spread}: {ms:.3f} ms  ({items * 4 / ms / 1e6:.1f} G corner-updates/s)")


if __name__ == "__main__":
    main()
