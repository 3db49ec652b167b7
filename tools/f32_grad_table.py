"""Per-tensor gradient distances of the fp32 validation mode (R50 by default): HIP fp32 vs float64 oracle, CPU fp32 oracle vs float64."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(depth=50):
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    HF.set_precision("fp32")
    for k in os.environ.get("F32_OFF", "").split(","):
        if k == "defer":
            from slenderobjdet_amd.layers import nn as nnl
            nnl.DEFER_LATERAL_DGRAD = False
    cfg = make_cfg(depth)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 3, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    refs = {}
    for tag in ("f32", "f64"):
        oracle = OracleFCOS.from_hip_model(model)
        if tag == "f64":
            oracle.double()
        losses = oracle.losses(cpu)
        refs[tag] = dict(zip(oracle.trainable().keys(), torch.autograd.grad(sum(losses.values()), list(oracle.trainable().values()))))
    got = model(data)
    opt.zero_grad()
    model.arena.begin_backward(); sum(got.values()).backward(); model.arena.finish_backward()
    rows = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = p.grad.detach().double().cpu()
        g = g.permute(0, 3, 1, 2) if g.dim() == 4 else g
        r32, r64 = refs["f32"][name].double(), refs["f64"][name]
        n = max(r64.norm().item(), 1e-30)
        rows.append(((g - r64).norm().item() / n, (r32 - r64).norm().item() / n, name, tuple(g.shape)))
    rows.sort(reverse=True)
    for d_hip, d_cpu, name, shp in rows[:40]:
        print(f"{d_hip:.2e}  {d_cpu:.2e}  {d_hip / max(d_cpu, 1e-30):8.1f}x  {name} {shp}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50)
