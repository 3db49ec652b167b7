"""Print the loss trajectory of a few training steps (debug aid): python tools/traj.py --arch retinanet --steps 15 --lr 0.01"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import damp_residual_branches, make_cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="retinanet")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    from slenderobjdet_amd.data import SyntheticCocoBatches
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(a.depth, a.arch)
    cfg.SOLVER.BASE_LR = a.lr
    torch.manual_seed(1)
    model = build_model(cfg)
    model.train()
    if a.arch == "retinanet" and a.depth >= 50:
        damp_residual_branches(model)
    opt = build_optimizer(cfg, model)
    loader = SyntheticCocoBatches(a.batch, 800, 1333, rank=0, device=torch.device("cuda", 0), pool=2)
    for it in range(a.steps):
        losses = model(next(loader))
        total = sum(losses.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
        gn = float(model.arena.grads.norm())
        opt.step()
        print(it, {k: round(float(v), 4) for k, v in losses.items()}, "gradnorm", round(gn, 3), flush=True)


if __name__ == "__main__":
    main()
