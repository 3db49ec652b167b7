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
    if os.environ.get("F32_CALIBRATE", "0") == "1":      # a well-conditioned problem: FrozenBN statistics that normalise (oracle/conditioning.py)
        from oracle.conditioning import calibrate_frozen_bn
        log = []
        print("calibrated", calibrate_frozen_bn(model, data, log=log), "norms; raw conv output std range",
              min(r[1] for r in log), max(r[1] for r in log))
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
    pair = sorted(((g_ - refs["f32"][n_].double()).norm().item() / max(refs["f32"][n_].double().norm().item(), 1e-30), n_)
                  for n_, g_ in ((name, (p.grad.detach().double().cpu().permute(0, 3, 1, 2) if p.grad.dim() == 4 else p.grad.detach().double().cpu()))
                                 for name, p in model.named_parameters() if p.requires_grad))
    print(f"hip32 - cpu32: max {pair[-1][0]:.2e} ({pair[-1][1]}), median {pair[len(pair) // 2][0]:.2e}, "
          f"share <= 1e-4: {sum(d <= 1e-4 for d, _ in pair) / len(pair):.3f}, tensors {len(pair)}")
    print("losses hip / f32 / f64:", {k: float(v) for k, v in got.items()})
    for d_hip, d_cpu, name, shp in rows[:12]:
        print(f"{d_hip:.2e}  {d_cpu:.2e}  {d_hip / max(d_cpu, 1e-30):8.1f}x  {name} {shp}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50)
