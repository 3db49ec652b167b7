"""Times the multi-level head convolution (5 FPN levels, 256->256 3x3, batch 16 @ 800x1344) fwd / dgrad."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N, C, K = 16, 256, int(os.environ.get("BK_OUT", 256))
hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
xs = [torch.randn(N, h, w, C, device=dev).bfloat16() for h, w in hw]
dys = [torch.randn(N, h, w, K, device=dev).bfloat16() for h, w in hw]
w = (torch.randn(K, 3, 3, C, device=dev) * 0.05).bfloat16()
wt = w.permute(3, 1, 2, 0).contiguous()
fl = sum(2.0 * N * h * ww * K * 9 * C for h, ww in hw)
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
t = timeit(lambda: HF.conv2d_fwd_ml(xs, w, None, 1, 1, 1, relu=False))
t2 = timeit(lambda: HF.conv2d_dgrad_ml(dys, wt, hw, 1, 1, 1))
print(json.dumps({"fwd_ms": round(t, 3), "fwd_TF": round(fl / t / 1e9, 1), "dgrad_ms": round(t2, 3), "dgrad_TF": round(fl / t2 / 1e9, 1)}))
