"""Scratch end-to-end check on the GPU: build FCOS, run forward/backward/step a few times, print losses + timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slenderobjdet_amd.config import fresh_cfg
from slenderobjdet_amd.modeling import build_model
from slenderobjdet_amd.solver import build_optimizer
from slenderobjdet_amd.data import synthetic_batch

depth = int(os.environ.get("DEPTH", 50)); N = int(os.environ.get("N", 2)); H = int(os.environ.get("H", 800)); W = int(os.environ.get("W", 1333))
cfg = fresh_cfg()
cfg.MODEL.META_ARCHITECTURE = "FCOSV2"
cfg.MODEL.BACKBONE.NAME = "build_retinanet_resnet_fpn_backbone_use_p5"
cfg.MODEL.RESNETS.OUT_FEATURES = ["res3", "res4", "res5"]; cfg.MODEL.FPN.IN_FEATURES = ["res3", "res4", "res5"]
cfg.MODEL.RESNETS.DEPTH = depth
if depth in (18, 34): cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 64
cfg.MODEL.FCOS.CENTER_SAMPLING_RADIUS = 1.5; cfg.MODEL.FCOS.IOU_LOSS_TYPE = "giou"; cfg.MODEL.FCOS.CENTERNESS_ON_REG = True
cfg.SOLVER.BASE_LR = 0.01
torch.manual_seed(0)
model = build_model(cfg); model.train()
opt = build_optimizer(cfg, model)
print("params", sum(p.numel() for p in model.parameters()), "arena", model.arena.total, "head logical", model.head.num_logical_params())
data = synthetic_batch(N, H, W, 1234, device="cuda")
for it in range(int(os.environ.get("ITERS", 6))):
    torch.cuda.synchronize(); t0 = time.time()
    losses = model(data)
    total = sum(losses.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    opt.step()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(it, {k: round(float(v), 5) for k, v in losses.items()}, "total", round(float(total), 5), f"{dt*1e3:.1f} ms", flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
