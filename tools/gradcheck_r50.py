import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_model as T
from oracle.model import OracleFCOS
from slenderobjdet_amd.data import synthetic_batch
from slenderobjdet_amd.layers import functional as HF, nn as HN
cfg, model, opt = T._build(50)
data = synthetic_batch(2, 320, 384, 3, device="cuda")
grads = {}
for emu in (True, False):
    oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
    ref = oracle.losses(T._cpu(data))
    names = list(oracle.trainable().keys())
    grads[emu] = dict(zip(names, torch.autograd.grad(sum(ref.values()), list(oracle.trainable().values()))))
def run(det, fused, stats):
    HF.DETERMINISTIC, HN.GN_BWD_FUSED, HN.GN_EPILOGUE_STATS = det, fused, stats
    got = model(data); total = sum(got.values()); opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    rows = []
    for name, p in model.named_parameters():
        if not p.requires_grad: continue
        g = p.grad.detach().float().cpu()
        if g.dim() == 4: g = g.permute(0, 3, 1, 2)
        r32, remu = grads[False][name], grads[True][name]
        n = max(r32.norm().item(), 1e-12)
        d_hip, d_emu = (g - r32).norm().item() / n, (remu - r32).norm().item() / n
        rows.append((d_hip / (1.5 * d_emu + 0.01), d_hip, d_emu, name))
    rows.sort(reverse=True)
    print("det", det, "fused", fused, "stats", stats, "worst:", [(round(r[0], 3), round(r[1], 3), round(r[2], 3), r[3]) for r in rows[:3]], flush=True)
for _ in range(2):
    run(True, True, True); run(False, False, False); run(False, False, True); run(False, True, True)
