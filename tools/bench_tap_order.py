import os, sys, torch
sys.path.insert(0, "/root/repo")
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N = 16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best * 1e3
for (H, W, C, K, R, st) in [(100, 168, 128, 128, 3, 1), (25, 42, 512, 512, 3, 1), (50, 84, 256, 256, 3, 1), (25, 42, 256, 256, 3, 1)]:
    pad = R // 2
    x = torch.randn(N, H, W, C, device=dev).relu().bfloat16()
    w = torch.randn(K, R, R, C, device=dev) * 0.05
    wk, wt = HF.weight_prep(w)
    b = torch.zeros(K, device=dev)
    dy = torch.randn(N, H, W, K, device=dev).bfloat16()
    tf = timeit(lambda: HF.conv2d_fwd(x, wk, b, None, st, pad, 1, relu=True))
    td = timeit(lambda: HF.conv2d_dgrad(dy, wt, (H, W), st, pad, 1, relu_mask=x))
    fl = 2.0 * N * H * W * K * R * R * C
    print(f"{H}x{W} C{C} K{K}: fwd {tf:6.1f} us ({fl/tf/1e6:6.1f} TF)  dgrad {td:6.1f} us ({fl/td/1e6:6.1f} TF)", flush=True)
