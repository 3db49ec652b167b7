"""A/B of the deeper LDS-DMA schedule of the 256x256 kernels (round 6): head-tower shape (5 FPN levels, 256 -> 256, 3x3, batch 16), forward,
data gradient and weight gradient, timed on the in-tree library and on gpurun_abl/lib_{w256,c256,both}_deep.so, with an output checksum per
library (the schedules issue the same MFMAs on the same operands in the same order: results must be EQUAL bit for bit).
    python tools/ab_deep.py [lib ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
hws = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
torch.manual_seed(0)
xs = [torch.randn(16, h, w, 256, device=dev).relu().bfloat16() for h, w in hws]
dys = [(torch.randn(16, h, w, 256, device=dev) * 1e-2).bfloat16() for h, w in hws]
w = (torch.randn(256, 3, 3, 256, device=dev) * 0.02)
wk, wt = HF.weight_prep(w)
bias = torch.randn(256, device=dev)
dw = torch.zeros(256, 3, 3, 256, device=dev)
def h(ts):
    m = hashlib.sha1()
    for t in ts: m.update(t.detach().float().cpu().numpy().tobytes())
    return m.hexdigest()[:10]
fns = {"wgrad": lambda: HF.conv2d_wgrad_ml(dys, xs, dw, 3, 3, 1, 1, 1, splits=-1),
       "fwd": lambda: HF.conv2d_fwd_ml(xs, wk, bias, 1, 1, 1, relu=True),
       "dgrad": lambda: HF.conv2d_dgrad_ml(dys, wt, hws, 1, 1, 1)}
out = []
dw.zero_(); fns["wgrad"](); sums = {"wgrad": h([dw]), "fwd": h(fns["fwd"]()), "dgrad": h(fns["dgrad"]())}
for name, fn in fns.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    flops = sum(2.0 * 16 * hh * ww * 256 * 9 * 256 for hh, ww in hws)
    out.append("%%s %%7.1f us %%6.1f TF/s %%s" %% (name, best * 1e3, flops / best / 1e9, sums[name]))
print(" | ".join(out))
''' % ROOT

libs = sys.argv[1:] or ["", "gpurun_abl/lib_w256_deep.so", "gpurun_abl/lib_c256_deep.so", "gpurun_abl/lib_both_deep.so", ""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["SOD_HIP_LIB"] = os.path.join(ROOT, lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-32s %s" % (lib or "in-tree", out.stdout.strip() or out.stderr.strip()[-400:]), flush=True)
