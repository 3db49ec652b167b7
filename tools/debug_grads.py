import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_model import _build, _cpu
from oracle.model import OracleFCOS
from slenderobjdet_amd.data import synthetic_batch
emu = os.environ.get("EMU", "1") == "1"
cfg, model, opt = _build(int(os.environ.get("DEPTH", 18)))
data = synthetic_batch(2, 320, 384, 3, device="cuda")
oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
ref = oracle.losses(_cpu(data))
got = model(data)
print({k: (float(got[k]), float(ref[k])) for k in ref})
total = sum(got.values())
opt.zero_grad(); model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
names = list(oracle.trainable().keys())
rg = dict(zip(names, torch.autograd.grad(sum(ref.values()), list(oracle.trainable().values()))))
for name, p in model.named_parameters():
    if not p.requires_grad: continue
    g = p.grad.detach().float().cpu(); r = rg[name]
    if g.dim() == 4: g = g.permute(0, 3, 1, 2)
    rel = (g - r).norm() / max(r.norm().item(), 1e-12)
    cos = (g * r).sum() / max((g.norm() * r.norm()).item(), 1e-20)
    print(f"{name:50s} rel {float(rel):8.4f} cos {float(cos):8.5f} |ref| {float(r.norm()):10.4g} |got| {float(g.norm()):10.4g}")
