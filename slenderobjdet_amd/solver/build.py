"""``slender_det.solver`` surface (reference: slender_det/solver/build.py:8-104) + the WarmupMultiStepLR schedule the
configs name (detectron2).

``build_optimizer`` keeps the reference's per-parameter grouping (norm weight decay, bias lr/decay) but, for
``SOLVER.OPTIM == "SGD"`` on a model that owns a flat arena, returns :class:`FusedSGD`: ONE kernel launch updates
every parameter (vs ~160 small per-tensor kernels) and the arena generation counter tells the convolutions to
refresh their bf16 compute copies.
"""
import bisect
from typing import Any, Dict, List, Optional, Set

import torch
import torch.nn as nn

from .. import _C
from .._C import call, ptr, stream_ptr
from ..layers import functional as HF
from ..layers.nn import HipGroupNorm

_NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
               nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm, HipGroupNorm)


def get_default_optimizer_params(model, base_lr, weight_decay, weight_decay_norm, bias_lr_factor=1.0,
                                 weight_decay_bias=None, overrides: Optional[Dict[str, Dict[str, float]]] = None):
    """solver/build.py:36-104: one group per parameter; norm layers use WEIGHT_DECAY_NORM, biases BIAS_LR_FACTOR /
    WEIGHT_DECAY_BIAS."""
    if weight_decay_bias is None:
        weight_decay_bias = weight_decay
    params: List[Dict[str, Any]] = []
    memo: Set[int] = set()
    for module in model.modules():
        for pname, value in module.named_parameters(recurse=False):
            if not value.requires_grad or id(value) in memo:
                continue
            memo.add(id(value))
            hp = {"lr": base_lr, "weight_decay": weight_decay}
            if isinstance(module, _NORM_TYPES):
                hp["weight_decay"] = weight_decay_norm
            elif pname == "bias":
                hp["lr"] = base_lr * bias_lr_factor
                hp["weight_decay"] = weight_decay_bias
            if overrides is not None and pname in overrides:
                hp.update(overrides[pname])
            params.append({"params": [value], "lr": hp["lr"], "weight_decay": hp["weight_decay"]})
    return params


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics (momentum, dampening 0, optional nesterov, L2 weight decay) over the flat arena."""

    def __init__(self, params, lr, momentum=0.0, nesterov=False, arena=None):
        if arena is None:
            raise ValueError("FusedSGD needs the model's ParamArena")
        super().__init__(params, dict(lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=0.0))
        self.arena = arena
        self._base_lr0 = lr
        self._segs, self._nseg = arena.build_segments(self.param_groups, lr)
        # the schedule multiplies every group's lr by the same factor: recover the current base lr from the first group
        # whose multiplier (bias lr factor etc.) is non-zero
        self._ref_group, self._ref_mult = 0, 1.0
        for i, g in enumerate(self.param_groups):
            if lr and g["lr"] != 0:
                self._ref_group, self._ref_mult = i, g["lr"] / lr
                break
        self._steps = 0
        self.grad_scale = 1.0
        if momentum != 0:
            arena.momentum = torch.zeros_like(arena.params)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    # The momentum lives in the arena, not in Optimizer.state: save / restore it explicitly so that ``--resume`` continues the run
    # torch.optim.SGD's momentum_buffer round trip would (slender_det/engine: detectron2's checkpointer stores optimizer.state_dict()).
    def state_dict(self):
        sd = super().state_dict()
        mom = self.arena.momentum
        sd["fused_sgd"] = {"steps": self._steps, "momentum": None if mom is None else mom.detach().cpu().clone(),
                           "arena_names": [(n, o, c) for n, o, c in self.arena.names]}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        fused = sd.pop("fused_sgd", None)
        super().load_state_dict(sd)
        if fused is None:
            raise ValueError("FusedSGD.load_state_dict: the checkpoint holds no 'fused_sgd' entry (momentum buffer); it was not written by FusedSGD")
        if [tuple(x) for x in fused["arena_names"]] != [tuple(x) for x in self.arena.names]:
            raise ValueError("FusedSGD.load_state_dict: the checkpoint's parameter arena layout differs from this model's")
        if fused["momentum"] is not None:
            if self.arena.momentum is None:
                self.arena.momentum = torch.zeros_like(self.arena.params)
            self.arena.momentum.copy_(fused["momentum"])
        self._steps = int(fused["steps"])      # > 0: the next step() must not re-initialise the buffer from the gradient
        # the segment table (per-parameter lr multipliers / weight decay) is structural and stays as built; the loaded param_groups
        # carry the schedule's current lr, which step() reads through the same reference group as before

    @torch.no_grad()
    def step(self, closure=None):
        # every group carries lr = base_lr(t) * its multiplier; read the schedule from the first group
        g0 = self.param_groups[0]
        lr_now = self.param_groups[self._ref_group]["lr"] / self._ref_mult
        a = self.arena
        HF.wgrad_join()
        call("sod_sgd_step", ptr(a.params), ptr(a.grads), ptr(a.momentum), ptr(self._segs), self._nseg, None, float(lr_now),
             float(g0["momentum"]), 1 if g0["nesterov"] else 0, 1 if self._steps == 0 else 0, float(self.grad_scale), stream_ptr())
        self._steps += 1
        a.bump()


class FusedAdaptive(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW / Adagrad semantics (default hyper-parameters of the reference's calls, slender_det/solver/build.py:26-31:
    betas (0.9, 0.999), eps 1e-8 / 1e-10, no amsgrad, lr_decay 0) over the flat arena: ONE launch of ``sod_adaptive_step`` per step
    instead of a chain of ATen kernels per parameter.  The first / second moment buffers are two more flat fp32 arenas."""

    MODES = {"ADAM": 0, "ADAMW": 1, "ADAGRAD": 2}

    def __init__(self, params, lr, kind, arena=None, betas=(0.9, 0.999), eps=None, lr_decay=0.0):
        if arena is None:
            raise ValueError("FusedAdaptive needs the model's ParamArena")
        if kind not in self.MODES:
            raise ValueError(kind)
        eps = (1e-10 if kind == "ADAGRAD" else 1e-8) if eps is None else eps
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0, lr_decay=lr_decay))
        self.arena, self.kind = arena, kind
        self._segs, self._nseg = arena.build_segments(self.param_groups, lr)
        self._ref_group, self._ref_mult = 0, 1.0
        for i, g in enumerate(self.param_groups):
            if lr and g["lr"] != 0:
                self._ref_group, self._ref_mult = i, g["lr"] / lr
                break
        self._steps = 0
        self.grad_scale = 1.0
        self.exp_avg = torch.zeros_like(arena.params) if kind != "ADAGRAD" else None
        self.exp_avg_sq = torch.zeros_like(arena.params)      # Adagrad: the running sum of squared gradients (initial value 0)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def state_dict(self):
        sd = super().state_dict()
        sd["fused_adaptive"] = {"kind": self.kind, "steps": self._steps, "exp_avg": None if self.exp_avg is None else self.exp_avg.detach().cpu().clone(),
                                "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(), "arena_names": [(n, o, c) for n, o, c in self.arena.names]}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        fused = sd.pop("fused_adaptive", None)
        super().load_state_dict(sd)
        if fused is None or fused.get("kind") != self.kind:
            raise ValueError(f"FusedAdaptive.load_state_dict: the checkpoint holds no '{self.kind}' moment buffers")
        if [tuple(x) for x in fused["arena_names"]] != [tuple(x) for x in self.arena.names]:
            raise ValueError("FusedAdaptive.load_state_dict: the checkpoint's parameter arena layout differs from this model's")
        if self.exp_avg is not None:
            self.exp_avg.copy_(fused["exp_avg"])
        self.exp_avg_sq.copy_(fused["exp_avg_sq"])
        self._steps = int(fused["steps"])

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        lr_now = self.param_groups[self._ref_group]["lr"] / self._ref_mult
        self._steps += 1
        t = self._steps
        b1, b2 = g0["betas"]
        if self.kind == "ADAGRAD":
            lr_now = lr_now / (1.0 + (t - 1) * g0["lr_decay"])
            bc1 = bc2s = 1.0
        else:
            bc1, bc2s = 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5
        a = self.arena
        HF.wgrad_join()
        call("sod_adaptive_step", ptr(a.params), ptr(a.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self._segs), self._nseg,
             self.MODES[self.kind], float(lr_now), float(b1), float(b2), float(g0["eps"]), float(bc1), float(bc2s), float(self.grad_scale), stream_ptr())
        a.bump()


def build_optimizer(cfg, model):
    """slender_det/solver/build.py:8-33.  On a model that owns a flat arena every SOLVER.OPTIM value is one fused launch over the arena
    (FusedSGD / FusedAdaptive); without an arena (CPU models) the torch.optim classes the reference constructs."""
    params = get_default_optimizer_params(model, base_lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                                          weight_decay_norm=cfg.SOLVER.WEIGHT_DECAY_NORM, bias_lr_factor=cfg.SOLVER.BIAS_LR_FACTOR,
                                          weight_decay_bias=cfg.SOLVER.WEIGHT_DECAY_BIAS)
    optim = cfg.SOLVER.OPTIM
    arena = getattr(model, "arena", None)
    if optim == "SGD":
        if arena is not None:
            return FusedSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV, arena=arena)
        return torch.optim.SGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV)
    if optim in FusedAdaptive.MODES and arena is not None and arena.device.type == "cuda":
        # (ADAGRAD: the reference passes an undefined name as the default weight_decay, solver/build.py:31; every group carries its own)
        return FusedAdaptive(params, cfg.SOLVER.BASE_LR, optim, arena=arena)
    if optim == "ADAM":
        return torch.optim.Adam(params, cfg.SOLVER.BASE_LR)
    if optim == "ADAMW":
        return torch.optim.AdamW(params, cfg.SOLVER.BASE_LR)
    if optim == "ADAGRAD":   # the reference passes an undefined name here (solver/build.py:31); use the config value
        return torch.optim.Adagrad(params, cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    raise ValueError(optim)


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", last_epoch=-1):
        if not list(milestones) == sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def _warm(self, it):
        if it >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = it / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self):
        w = self._warm(self.last_epoch)
        return [b * w * self.gamma ** bisect.bisect_right(self.milestones, self.last_epoch) for b in self.base_lrs]


def build_lr_scheduler(cfg, optimizer):
    name = cfg.SOLVER.LR_SCHEDULER_NAME
    if name != "WarmupMultiStepLR":
        raise ValueError(f"Unknown LR scheduler: {name}")
    return WarmupMultiStepLR(optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                             warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)
