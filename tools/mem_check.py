"""Runs the headline training step for a while and prints the allocator's footprint (stream-related pool growth check)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_cfg, train_step  # noqa: E402
from slenderobjdet_amd.data import SyntheticCocoBatches  # noqa: E402
from slenderobjdet_amd.modeling import build_model  # noqa: E402
from slenderobjdet_amd.solver import build_optimizer  # noqa: E402

cfg = make_cfg(50)
torch.manual_seed(1)
model = build_model(cfg)
model.train()
opt = build_optimizer(cfg, model)
loader = SyntheticCocoBatches(16, 800, 1333, device=torch.device("cuda"), pool=2)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    last = train_step(model, opt, next(loader))
    if i % 20 == 19:
        torch.cuda.synchronize()
        print(i + 1, "steps: allocated %.2f GB, reserved %.2f GB, peak reserved %.2f GB, loss %.4f" % (
            torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_reserved() / 2**30, float(last)), flush=True)
