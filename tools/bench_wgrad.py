"""wgrad / fwd micro-benchmark on the head shape; SPLITS env sweeps the split-K factor."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N, H, W, C, K, R = 16, 100, 168, 256, 256, 3
x = torch.randn(N, H, W, C, device=dev).bfloat16(); w = (torch.randn(K, R, R, C, device=dev) * 0.05).bfloat16()
dy = torch.randn(N, H, W, K, device=dev).bfloat16(); dw = torch.zeros(K, R, R, C, device=dev)
flops = 2.0 * N * H * W * K * R * R * C
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "fwd"):
    t = timeit(lambda: HF.conv2d_fwd(x, w, None, stride=1, pad=1)); print("fwd", round(t, 3), "ms", round(flops / t / 1e9, 1), "TF")
if which in ("all", "wgrad"):
    for sp in [int(v) for v in os.environ.get("SPLITS", "0,1,4,8,14,28,43,64").split(",")]:
        t = timeit(lambda: HF.conv2d_wgrad(dy, x, dw, R, R, 1, 1, 1, splits=sp)); print("wgrad splits", sp, round(t, 3), "ms", round(flops / t / 1e9, 1), "TF")
