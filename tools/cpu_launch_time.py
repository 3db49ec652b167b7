"""How long does the host need to ENQUEUE one training step?  (If this approaches the GPU time per step the step is launch-bound.)
Runs K steps without synchronising and reports host time per step, then the synchronised GPU time per step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from slenderobjdet_amd.data import SyntheticCocoBatches
from slenderobjdet_amd.modeling import build_model
from slenderobjdet_amd.solver import build_optimizer

torch.cuda.set_device(0)
cfg = bench.make_cfg(50, "fcos")
torch.manual_seed(1)
model = build_model(cfg); model.train()
opt = build_optimizer(cfg, model)
loader = SyntheticCocoBatches(16, 800, 1333, rank=0, device=torch.device("cuda", 0), pool=2)
for _ in range(5):
    bench.train_step(model, opt, next(loader))
torch.cuda.synchronize()
K = 10
host = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter()
    bench.train_step(model, opt, next(loader))
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/step:", [round(h * 1e3, 1) for h in host])
print("host total %.1f ms, +drain %.1f ms  => wall/step %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) / K * 1e3))
