"""GroupNorm passes of the head towers (five FPN levels, 256 channels, 32 groups, batch 16): us per call and effective HBM rate of the forward
(statistics + apply) and of the backward (reduce + apply), and a hash of the outputs (two builds of the library can be compared bit for bit:
SOD_HIP_LIB=<other.so> python tools/bench_gn.py)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, C, G = 16, 256, 32
for name, levels in (("fcos", [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]), ("rcnn-p2", [(200, 336)])):
    xs = [torch.randn(N, h, w, C, device=dev).bfloat16() for h, w in levels]
    dys = [torch.randn(N, h, w, C, device=dev).bfloat16() * 0.01 for h, w in levels]
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.1
    dgamma, dbeta, dxsum = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    nbytes = sum(x.numel() * 2 for x in xs)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / reps)
        return best

    ys, stats = HF.groupnorm_fwd_ml(xs, gamma, beta, G, relu=True)
    t_f = timed(lambda: HF.groupnorm_fwd_ml(xs, gamma, beta, G, relu=True))
    t_b = timed(lambda: HF.groupnorm_bwd_ml(dys, xs, gamma, beta, stats, G, dgamma, dbeta, relu=True, dxsum=dxsum))
    dgamma.zero_(); dbeta.zero_(); dxsum.zero_()
    dxs = HF.groupnorm_bwd_ml(dys, xs, gamma, beta, stats, G, dgamma, dbeta, relu=True, dxsum=dxsum)
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in dxs:
        h.update(t.view(torch.int16).cpu().numpy().tobytes())
    print(f"{name}: fwd (stats + apply, 3 passes) {t_f * 1e3:7.1f} us {3 * nbytes / t_f / 1e9:5.2f} TB/s | bwd (reduce + apply, 5 passes) {t_b * 1e3:7.1f} us "
          f"{5 * nbytes / t_b / 1e9:5.2f} TB/s | dx sha1 {h.hexdigest()[:12]} dgamma {float(dgamma.double().sum()):.6f} dbeta {float(dbeta.double().sum()):.6f} "
          f"dxsum {float(dxsum.double().sum()):.6f}", flush=True)
