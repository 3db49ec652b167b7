"""Where a launch of conv_wgrad9_kernel spends its time: the head-tower shape (5 FPN levels, 256 -> 256, 3x3, batch 16) on the shipped library
and on measurement builds with parts of the kernel removed (-DSOD_W9_ABL=bits: 1 LDS-DMA requests out of range = no memory traffic, 2 no fragment
reads, 4 no MFMAs; gpurun_abl/lib_w9_abl<bits>.so).    python tools/bench_wgrad9_abl.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)
xs = [torch.randn(16, h, w, 256, device=dev).relu().bfloat16() for h, w in hws]
dys = [(torch.randn(16, h, w, 256, device=dev) * 1e-2).bfloat16() for h, w in hws]
dw = torch.zeros(256, 3, 3, 256, device=dev)
fn = lambda: HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=-2)
for _ in range(3): fn()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 10)
flops = sum(2.0 * 16 * h * w * 256 * 9 * 256 for h, w in hws)
print("%%8.1f us  %%7.1f TFLOP/s (algorithmic)" %% (best * 1e3, flops / best / 1e9))
''' % ROOT
NAMES = {0: "shipped kernel", 1: "LDS-DMA requests out of range (no memory traffic)", 2: "no fragment reads", 4: "no MFMAs",
         6: "no fragment reads, no MFMAs (LDS-DMA + bookkeeping + barriers + slabs)", 7: "loop skeleton only",
         8: "no barrier in the loop (wrong results)", 16: "no vmcnt wait in the loop (wrong results)", 24: "neither barrier nor vmcnt wait", 32: "no s_setprio"}
for abl in (0, 1, 2, 4, 6, 7, 8, 16, 24, 32, 0):
    env = dict(os.environ)
    if abl:
        lib = os.path.join(ROOT, "gpurun_abl", "lib_w9_abl%d.so" % abl)
        if not os.path.exists(lib):
            continue
        env["SOD_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-75s %s" % (NAMES[abl], out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
