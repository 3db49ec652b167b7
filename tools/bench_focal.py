"""Times the focal-loss kernels on the FCOS logits shape (16 x 22400 x 80)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.layers import functional as HF
dev = torch.device("cuda:0")
N, L, K = 16, 22400, 80
x = torch.randn(N, L, K, device=dev) * 3 - 4
lab = torch.randint(0, 200, (N, L), device=dev, dtype=torch.int32)
one = torch.ones(1, device=dev)
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out = torch.empty((N, L, K), dtype=torch.bfloat16, device=dev)
print("fwd %.1f us" % timeit(lambda: HF.focal_loss_fwd(x, lab, None, 0.25, 2.0, K=K)))
print("bwd %.1f us" % timeit(lambda: HF.focal_loss_bwd(x, lab, None, 0.25, 2.0, K=K, scale_num=one, scale_den=one, den_mul=1.0, den_min=1.0, ld_out=K, out_bf16=True, out=out)))
