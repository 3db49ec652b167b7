"""In-kernel wall-clock stamps of conv_wgrad256_kernel (measurement builds -DSOD_W256_ABL=16|bits, gpurun_abl/lib_w256_abl<n>.so): per
workgroup the time of the prologue, the K loop and the epilogue (s_memrealtime, 10-ns ticks) and the loop's shader cycles, read back from
the slab workspace.   SOD_HIP_LIB=gpurun_abl/lib_w256_abl16.so python tools/bench_wgrad256_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)
xs = [torch.randn(16, h, w, 256, device=dev).relu().bfloat16() for h, w in hws]
dys = [(torch.randn(16, h, w, 256, device=dev) * 1e-2).bfloat16() for h, w in hws]
dw = torch.zeros(256, 3, 3, 256, device=dev)
for _ in range(3):
    HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=-1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=-1)
e.record()
torch.cuda.synchronize()
ws = HF.wgrad_workspace(dev).view(torch.float32)
tiles, nz, SLAB = 9, 28, 65536
o = ws[: tiles * nz * SLAB].view(nz * tiles, SLAB)
pro, loop, epi, T, cyc, t0 = (o[:, i].cpu() for i in (0, 1, 2, 3, 256, 257))
med = lambda t: float(t.median())
print(f"{os.environ.get('SOD_HIP_LIB', 'shipped')}: launch + reduce {s.elapsed_time(e) * 1e3:.1f} us | per workgroup (median / max, us): "
      f"prologue {med(pro) / 100:.2f} / {float(pro.max()) / 100:.2f}, K loop {med(loop) / 100:.1f} / {float(loop.max()) / 100:.1f} "
      f"({med(loop) * 10 / med(T):.0f} ns per K-tile, {med(T):.0f} K-tiles, {med(cyc) / med(T):.0f} shader cycles per K-tile = "
      f"{med(cyc) / (med(loop) * 10):.2f} GHz), epilogue {med(epi) / 100:.2f} / {float(epi.max()) / 100:.2f}; "
      f"start spread {(float(t0.max()) - float(t0.min())) / 100:.1f} us")
