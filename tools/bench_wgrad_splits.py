"""wgrad time vs number of pixel splits for the backbone 1x1 / 3x3 shapes (SOD_WGRAD_PLAIN=1 replaces the atomics by racy plain stores:
the difference is what the atomic epilogue costs)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N = 16
SHAPES = [("res4_conv3", 50, 84, 256, 1024, 1), ("res4_conv1", 50, 84, 1024, 256, 1), ("res3_conv3", 100, 168, 128, 512, 1),
          ("res3_conv1", 100, 168, 512, 128, 1), ("res5_conv3", 25, 42, 512, 2048, 1), ("res2_conv3", 200, 336, 64, 256, 1),
          ("res4_conv2", 50, 84, 256, 256, 3), ("res3_conv2", 100, 168, 128, 128, 3)]
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, H, W, C, K, R in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    dy = torch.randn(N, H, W, K, device=dev).bfloat16()
    dw = torch.zeros(K, R, R, C, device=dev)
    fl = 2.0 * N * H * W * K * R * R * C
    row = {"name": name}
    for sp in (0, 4, 8, 16, 32, 64, 128):
        t = timeit(lambda: HF.conv2d_wgrad(dy, x, dw, R, R, 1, R // 2, 1, splits=sp))
        row[f"s{sp}"] = f"{t*1e3:.0f}us/{fl/t/1e9:.0f}TF"
    print(json.dumps(row), flush=True)
