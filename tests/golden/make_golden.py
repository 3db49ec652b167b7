#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by importing the REFERENCE's own Python (read-only, from
/root/reference) in the build container.  Runs only where /root/reference exists; nothing from the reference is copied:
only inputs/outputs (numpy arrays) are written.

detectron2 / fvcore are absent everywhere, so the reference files are loaded one by one with
importlib under small stub modules written here.  Vectors that pass through a stubbed
third-party op are labelled "reference-Python x restated-op" in meta.json.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz + meta.json
"""
import importlib.util
import json
import math
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    m.__path__ = []
    sys.modules[name] = m
    return m


def focal_restated(inputs, targets, alpha: float = -1, gamma: float = 2, reduction: str = "none"):
    """fvcore.nn.sigmoid_focal_loss (absent) restated from its documented formula."""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


class _Boxes:
    def __init__(self, t):
        self.tensor = t

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


class _Reg:
    def register(self, obj=None):
        return (lambda f: f) if obj is None else obj


def install_stubs():
    _stub("detectron2")
    _stub("detectron2.layers", cat=lambda ts, dim=0: ts[0] if len(ts) == 1 else torch.cat(ts, dim),
          ShapeSpec=SimpleNamespace, batched_nms=None)
    _stub("detectron2.modeling")
    _stub("detectron2.modeling.meta_arch", META_ARCH_REGISTRY=_Reg())
    _stub("detectron2.modeling.postprocessing", detector_postprocess=None)
    _stub("detectron2.structures", ImageList=None, Instances=None, Boxes=_Boxes)
    _stub("fvcore")
    _stub("fvcore.nn", sigmoid_focal_loss_jit=focal_restated)


def main():
    assert os.path.isdir(REF), "make_golden.py only runs in the build container (needs /root/reference)"
    install_stubs()
    iou_mod = _load("ref_iou_loss", "slender_det/layers/iou_loss.py")
    scale_mod = _load("ref_scale", "slender_det/layers/scale.py")
    utils = _load("ref_fcos_utils", "slender_det/modeling/meta_arch/fcos/utils.py")
    _stub("slender_det")
    _stub("slender_det.modeling")
    _stub("slender_det.modeling.backbone", build_backbone=None)
    _stub("slender_det.layers", Scale=scale_mod.Scale, iou_loss=iou_mod.iou_loss, DFConv2d=None)
    # fcosv2.py does `from .utils import ...`: give it a package context
    pkg = _stub("refpkg")
    sys.modules["refpkg.utils"] = utils
    spec = importlib.util.spec_from_file_location("refpkg.fcosv2", os.path.join(REF, "slender_det/modeling/meta_arch/fcos/fcosv2.py"))
    v2 = importlib.util.module_from_spec(spec)
    sys.modules["refpkg.fcosv2"] = v2
    spec.loader.exec_module(v2)
    v2.INF = utils.INF   # the reference forgot this import (fcosv2.py:157)

    meta = {}
    g = torch.Generator().manual_seed(0)

    # ---------------------------------------------------------------- iou_loss (3 types, fwd + grad) : pure reference
    P = 97
    pred = (torch.rand(P, 4, generator=g) * 50 + 1)
    tgt = (torch.rand(P, 4, generator=g) * 50 + 1)
    w = torch.rand(P, generator=g)
    pred[3] = tgt[3]
    out = {"pred": pred.numpy(), "target": tgt.numpy(), "weight": w.numpy()}
    for lt in ("iou", "linear_iou", "giou"):
        p = pred.clone().requires_grad_(True)
        loss = iou_mod.iou_loss(p, tgt, w, loss_type=lt)
        (gp,) = torch.autograd.grad(loss, p)
        out[f"loss_{lt}"] = loss.detach().numpy()
        out[f"grad_{lt}"] = gp.numpy()
        out[f"loss_noweight_{lt}"] = iou_mod.iou_loss(pred, tgt, None, loss_type=lt).numpy()
    np.savez(os.path.join(OUT, "iou_loss.npz"), **out)
    meta["iou_loss.npz"] = "reference: slender_det/layers/iou_loss.py:4-37 (pure reference Python)"

    # ---------------------------------------------------------------- centerness targets : pure reference
    reg = torch.rand(64, 4, generator=g) * 40 + 0.5
    np.savez(os.path.join(OUT, "centerness.npz"), reg=reg.numpy(), ctr=utils.compute_centerness_targets(reg).numpy(),
             slender=utils.compute_slender_centerness_targets(reg).numpy())
    meta["centerness.npz"] = "reference: fcos/utils.py:295-312 (pure reference Python)"

    # ---------------------------------------------------------------- locations : pure reference
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    strides = [8, 16, 32, 64, 128]
    locs = utils.compute_locations(shapes, strides, torch.device("cpu"))
    np.savez(os.path.join(OUT, "locations.npz"), shapes=np.array(shapes), strides=np.array(strides),
             **{f"loc{i}": l.numpy() for i, l in enumerate(locs)})
    meta["locations.npz"] = "reference: fcos/utils.py:82-105 (pure reference Python)"

    # ---------------------------------------------------------------- target assignment : pure reference
    def inst(boxes, classes):
        return SimpleNamespace(gt_boxes=_Boxes(boxes), gt_classes=classes)

    small_shapes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    locs_s = utils.compute_locations(small_shapes, strides, torch.device("cpu"))
    boxes = [
        torch.tensor([[10.0, 20.0, 200.0, 180.0], [30.0, 40.0, 90.0, 220.0], [100.0, 5.0, 310.0, 250.0], [150.0, 100.0, 170.0, 240.0]]),
        torch.tensor([[0.0, 0.0, 319.0, 255.0], [120.0, 60.0, 180.0, 90.0], [120.0, 60.0, 180.0, 90.0]]),   # duplicate box: argmin tie
        torch.tensor([[-40.0, 10.0, 40.0, 200.0], [60.0, 60.0, 260.0, 200.0]]),                             # first centre x == 0 (quirk)
    ]
    classes = [torch.tensor([3, 17, 60, 79]), torch.tensor([5, 8, 9]), torch.tensor([1, 2])]
    soi = []
    sizes = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, utils.INF]]
    for l, pts in enumerate(locs_s):
        soi.append(pts.new_tensor(sizes[l])[None].expand(len(pts), -1))
    soi = torch.cat(soi, 0)
    out = {"shapes": np.array(small_shapes), "strides": np.array(strides)}
    for i, b in enumerate(boxes):
        out[f"boxes{i}"], out[f"classes{i}"] = b.numpy(), classes[i].numpy()
    for radius in (0.0, 1.5):
        lab, rt = utils.compute_targets_for_locations(locs_s, [inst(b, c) for b, c in zip(boxes, classes)], soi, strides, radius, 80)
        out[f"labels_r{radius}"], out[f"reg_r{radius}"] = lab.numpy(), rt.numpy()
    np.savez(os.path.join(OUT, "fcos_targets.npz"), **out)
    meta["fcos_targets.npz"] = "reference: fcos/utils.py:108-212 (pure reference Python; 3 images, radius 0 and 1.5)"

    # ---------------------------------------------------------------- FCOSHead + FCOSV2.losses : reference x restated focal
    def ns(**k):
        return SimpleNamespace(**k)

    for tag, ctr_on_reg, norm_reg, iou_type, radius in (("a", True, False, "giou", 1.5), ("b", False, False, "iou", 0.0), ("c", True, True, "linear_iou", 1.5)):
        fc = ns(NUM_CLASSES=80, FPN_STRIDES=strides, NORM_REG_TARGETS=norm_reg, CENTERNESS_ON_REG=ctr_on_reg, USE_DCN_IN_TOWER=False,
                USE_DCN_V2=True, NUM_CONVS=4, PRIOR_PROB=0.01)
        cfg = ns(MODEL=ns(FCOS=fc))
        torch.manual_seed(11)
        head = v2.FCOSHead(cfg, [SimpleNamespace(channels=32, stride=8)])
        assert sum(p.numel() for p in head.parameters()) > 0
        with torch.no_grad():   # make GN affine and scales non-trivial
            for m in head.modules():
                if isinstance(m, torch.nn.GroupNorm):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.2, 0.2)
            for i, s in enumerate(head.scales):
                s.scale.fill_(0.8 + 0.1 * i)
        fshapes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
        feats = [torch.randn(2, 32, h, w, generator=g) for h, w in fshapes]
        logits, bbox_reg, ctrness = head(feats)
        locs_f = utils.compute_locations(fshapes, strides, torch.device("cpu"))
        gtb = [torch.tensor([[8.0, 8.0, 120.0, 100.0], [40.0, 20.0, 150.0, 120.0]]), torch.tensor([[20.0, 30.0, 90.0, 110.0]])]
        gtc = [torch.tensor([7, 33]), torch.tensor([52])]
        self_ns = ns(num_classes=80, fpn_strides=strides, center_sampling_radius=radius, norm_reg_targets=norm_reg,
                     focal_loss_alpha=0.25, focal_loss_gamma=2.0, iou_loss_type=iou_type)
        labels, reg_t = v2.FCOSV2.get_ground_truth(self_ns, locs_f, [inst(b, c) for b, c in zip(gtb, gtc)])
        losses = v2.FCOSV2.losses(self_ns, labels, reg_t, logits, bbox_reg, ctrness)
        total = sum(losses.values())
        params = dict(head.named_parameters())
        grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
        out = {f"feat{i}": f.numpy() for i, f in enumerate(feats)}
        out.update({f"gtb{i}": b.numpy() for i, b in enumerate(gtb)})
        out.update({f"gtc{i}": c.numpy() for i, c in enumerate(gtc)})
        out.update({"param::" + k: v.detach().numpy() for k, v in params.items()})
        keep_full = ("cls_logits.", "bbox_pred.", "centerness.", "scales.", "cls_tower.0.", "cls_tower.1.", "bbox_tower.9.", "bbox_tower.10.")
        for k, gr in zip(params, grads):
            gr = gr if gr is not None else torch.zeros_like(params[k])
            out["gradnorm::" + k] = gr.norm().numpy()
            if k.startswith(keep_full):
                out["grad::" + k] = gr.numpy()
        out.update({"loss::" + k: v.detach().numpy() for k, v in losses.items()})
        out["labels"], out["reg_targets"] = labels.numpy(), reg_t.numpy()
        for i in range(5):
            out[f"bbox{i}"], out[f"ctr{i}"] = bbox_reg[i].detach().numpy(), ctrness[i].detach().numpy()
            if i >= 1:    # level 0 logits are the bulk of the bytes: keep a strided sample instead
                out[f"logits{i}"] = logits[i].detach().numpy()
            else:
                out["logits0_ch0_7"] = logits[0][:, :8].detach().numpy()
        out["cfg"] = np.array([int(ctr_on_reg), int(norm_reg), {"iou": 0, "linear_iou": 1, "giou": 2}[iou_type], radius])
        np.savez(os.path.join(OUT, f"fcos_head_losses_{tag}.npz"), **out)
        meta[f"fcos_head_losses_{tag}.npz"] = ("reference-Python x restated-op: fcosv2.py FCOSHead (:277-381), get_ground_truth (:150-172), "
                                                "losses (:104-148) with fvcore sigmoid_focal_loss_jit restated (SURVEY C.1)")

    # ---------------------------------------------------------------- DeformConv known-answer test of the reference's tests
    src = open(os.path.join(REF, "tests/test_deformable_conv.py")).read().splitlines()
    grid = _load_grid()
    ns_exec = {"torch": torch, "F": F, "uniform_grid": grid.uniform_grid, "zero_center_grid": grid.zero_center_grid}
    exec("\n".join(src[10:64]), ns_exec)    # the reference's CPU helpers my_dconv / my_conv (tests/test_deformable_conv.py:11-64)
    # inputs exactly as the test builds them (tests/test_deformable_conv.py:70-82), on CPU
    weight = torch.arange(9).float().reshape(1, 1, 3, 3).repeat(1, 2, 1, 1)
    gr = grid.uniform_grid(4).unsqueeze(0).permute(0, 3, 1, 2)
    inp = torch.stack([gr[:, 0], torch.zeros_like(gr[:, 0]) + 0.1], 1)
    offsets_1 = grid.zero_center_grid(3).reshape(1, -1, 1, 1).repeat(1, 1, 4, 4)
    offsets_2 = torch.zeros_like(offsets_1)
    np.savez(os.path.join(OUT, "deform_conv_kat.npz"), input=inp.numpy(), weight=weight.numpy(), offsets_1=offsets_1.numpy(),
             offsets_2=offsets_2.numpy(), y_conv=ns_exec["my_conv"](inp, weight).numpy(),
             y_dconv_zero=ns_exec["my_dconv"](inp, offsets_2, weight).numpy(), y_dconv_1=ns_exec["my_dconv"](inp, offsets_1, weight).numpy())
    meta["deform_conv_kat.npz"] = ("reference: tests/test_deformable_conv.py:11-87 — helpers my_conv/my_dconv evaluated on the test's own inputs "
                                   "(the values its asserts compare detectron2's DeformConv against)")

    json.dump(meta, open(os.path.join(OUT, "meta.json"), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(meta))


def _load_grid():
    _stub("concern")
    _stub("concern.support", make_dual=lambda x: (x, x) if not isinstance(x, (tuple, list)) else tuple(x))
    return _load("ref_grid", "slender_det/modeling/grid_generator.py")


if __name__ == "__main__":
    main()
