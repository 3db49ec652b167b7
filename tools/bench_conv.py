"""Micro-benchmark of the HIP implicit-GEMM conv kernels on the FCOS R50-FPN layer shapes (batch 16, 800x1344)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.layers import functional as HF

dev = torch.device("cuda:0")
N = int(os.environ.get("BN", 16))
# (name, H, W, C, K, R, stride, pad)
SHAPES = [
    ("head3x3_p3", 100, 168, 256, 256, 3, 1, 1),
    ("head3x3_p4", 50, 84, 256, 256, 3, 1, 1),
    ("head3x3_p5", 25, 42, 256, 256, 3, 1, 1),
    ("cls_logits_p3", 100, 168, 256, 80, 3, 1, 1),
    ("res3_conv2", 100, 168, 128, 128, 3, 1, 1),
    ("res3_conv3", 100, 168, 128, 512, 1, 1, 0),
    ("res3_conv1", 100, 168, 512, 128, 1, 1, 0),
    ("res4_conv2", 50, 84, 256, 256, 3, 1, 1),
    ("res4_conv3", 50, 84, 256, 1024, 1, 1, 0),
    ("res4_conv1", 50, 84, 1024, 256, 1, 1, 0),
    ("res5_conv2", 25, 42, 512, 512, 3, 1, 1),
    ("res2_conv2", 200, 336, 64, 64, 3, 1, 1),
    ("res2_conv3", 200, 336, 64, 256, 1, 1, 0),
    ("stem", 800, 1344, 8, 64, 7, 2, 3),
    ("fpn_out_p3", 100, 168, 256, 256, 3, 1, 1),
]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
for name, H, W, C, K, R, st, pad in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    w = (torch.randn(K, R, R, C, device=dev) * 0.05).bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, 1)
    dy = torch.randn(N, Ho, Wo, K, device=dev).bfloat16()
    dw = torch.zeros(K, R, R, C, device=dev)
    flops = 2.0 * N * Ho * Wo * K * R * R * C
    row = {"name": name, "gflop": round(flops / 1e9, 1)}
    if "fwd" in which:
        t = timeit(lambda: HF.conv2d_fwd(x, w, None, stride=st, pad=pad)); row["fwd_ms"] = round(t, 3); row["fwd_TF"] = round(flops / t / 1e9, 1)
    if "dgrad" in which and C % 8 == 0 and name != "stem":
        t = timeit(lambda: HF.conv2d_dgrad(dy, wt, (H, W), st, pad, 1)); row["dgrad_ms"] = round(t, 3); row["dgrad_TF"] = round(flops / t / 1e9, 1)
    if "wgrad" in which and name != "stem":
        t = timeit(lambda: HF.conv2d_wgrad(dy, x, dw, R, R, st, pad, 1)); row["wgrad_ms"] = round(t, 3); row["wgrad_TF"] = round(flops / t / 1e9, 1)
    print(json.dumps(row), flush=True)
