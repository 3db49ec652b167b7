#!/usr/bin/env python
"""Golden vectors for the RepPoints path, produced by the REFERENCE's own Python (read-only /root/reference) in the build
container.  Only inputs/outputs (numpy arrays) are written; no reference source is copied.

  * reppoints_matchers.npz — slender_det/modeling/matchers/rep_matcher.py (rep_points_match, nearest_point_match, inside_match)
    with slender_det/structures/points.py: PURE reference Python (only ``Boxes`` is a 10-line container stub).
  * reppoints_losses.npz — RepPointsDetector.points2bbox / get_ground_truth / losses (rpd.py:221-402) called unbound on a
    SimpleNamespace ``self``: reference Python x restated third-party ops (detectron2 pairwise_iou / Matcher, fvcore focal /
    smooth-L1 are absent everywhere; their restatements below follow SURVEY.md Appendix C.1-C.5).

    python tests/golden/make_golden_reppoints.py
"""
import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel, package=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Boxes:
    def __init__(self, t):
        self.tensor = t

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __getitem__(self, i):
        return Boxes(self.tensor[i].view(-1, 4))

    def __len__(self):
        return self.tensor.shape[0]


def pairwise_iou(b1, b2):
    """detectron2.structures.pairwise_iou restated."""
    a1, a2 = b1.area(), b2.area()
    t1, t2 = b1.tensor, b2.tensor
    wh = (torch.min(t1[:, None, 2:], t2[:, 2:]) - torch.max(t1[:, None, :2], t2[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


class Matcher:
    """detectron2.modeling.matcher.Matcher restated."""

    def __init__(self, thresholds, labels, allow_low_quality_matches=False):
        self.thresholds = [-float("inf")] + list(thresholds) + [float("inf")]
        self.labels, self.allow = labels, allow_low_quality_matches

    def __call__(self, q):
        vals, matches = q.max(dim=0)
        lab = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for l, lo, hi in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            lab[(vals >= lo) & (vals < hi)] = l
        if self.allow:
            best, _ = q.max(dim=1)
            lab[(q == best[:, None]).nonzero()[:, 1]] = 1
        return matches, lab


def focal_restated(inputs, targets, alpha=-1, gamma=2, reduction="none"):
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum() if reduction == "sum" else loss


def smooth_l1_restated(inp, target, beta, reduction="none"):
    n = torch.abs(inp - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta) if beta >= 1e-5 else n
    return loss.sum() if reduction == "sum" else loss


class _Reg:
    def register(self, obj=None):
        return (lambda f: f) if obj is None else obj


class _Storage:
    def put_scalar(self, *a, **k):
        pass


def install():
    _stub("cv2")
    _stub("concern")
    _stub("concern.webcv2")
    _stub("concern.support", make_dual=lambda s: (s, s) if isinstance(s, int) else tuple(s))
    _stub("fvcore")
    _stub("fvcore.nn", sigmoid_focal_loss_jit=focal_restated, smooth_l1_loss=smooth_l1_restated)
    _stub("detectron2")
    _stub("detectron2.utils")
    _stub("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    _stub("detectron2.utils.events", get_event_storage=lambda: _Storage())
    _stub("detectron2.structures", Boxes=Boxes, ImageList=None, Instances=None, pairwise_iou=pairwise_iou)
    _stub("detectron2.layers", DeformConv=None, cat=torch.cat, batched_nms=None)
    _stub("detectron2.modeling")
    _stub("detectron2.modeling.meta_arch", META_ARCH_REGISTRY=_Reg(), RetinaNet=None)
    _stub("detectron2.modeling.meta_arch.retinanet", permute_to_N_HWA_K=None)
    _stub("detectron2.modeling.backbone", build_backbone=None)
    _stub("detectron2.modeling.matcher", Matcher=Matcher)
    _stub("detectron2.modeling.postprocessing", detector_postprocess=None)
    # package chain for rep_matcher's ``from ...structures.points import`` relative import
    _stub("refsd")
    _stub("refsd.structures")
    _stub("refsd.modeling")
    _stub("refsd.modeling.matchers")
    _load("refsd.structures.points", "slender_det/structures/points.py", "refsd.structures")
    rm = _load("refsd.modeling.matchers.rep_matcher", "slender_det/modeling/matchers/rep_matcher.py", "refsd.modeling.matchers")
    _stub("slender_det")
    _stub("slender_det.modeling")
    _load("slender_det.modeling.grid_generator", "slender_det/modeling/grid_generator.py")
    rpd = _load("ref_rpd", "slender_det/modeling/meta_arch/reppoints/rpd.py")
    return rm, rpd


def grid(hw, strides):
    cs, ss = [], []
    for (h, w), s in zip(hw, strides):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        cs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), 1) * s)
        ss.append(torch.full((h * w,), float(s)))
    return cs, ss


def random_boxes(g, n, H, W):
    cx, cy = torch.rand(n, generator=g) * W, torch.rand(n, generator=g) * H
    w = torch.exp2(torch.rand(n, generator=g) * 5.0 + 3.0)
    h = torch.exp2(torch.rand(n, generator=g) * 5.0 + 3.0)
    b = torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, W)
    b[:, 1::2] = b[:, 1::2].clamp(0, H)
    keep = ((b[:, 2] - b[:, 0]) > 2) & ((b[:, 3] - b[:, 1]) > 2)
    return b[keep]


class _OracleDeformConv(torch.nn.Module):
    """Stand-in for detectron2.layers.DeformConv (absent): the restated op of oracle/deform_conv.py, itself pinned by the
    reference's own known-answer test (tests/golden/deform_conv_kat.npz)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(cout, cin, k, k))
        torch.nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        self.stride, self.pad = stride, pad

    def forward(self, x, offset):
        sys.path.insert(0, os.path.join(OUT, "..", ".."))
        from oracle.deform_conv import deform_conv2d

        return deform_conv2d(x, offset, self.weight, None, self.stride, self.pad, 1)


def pointset_head_golden(g):
    """PointSetHead (meta/heads/pointset_head.py + meta_head.py + utils.py) built and run by the reference's own Python on CPU:
    towers, init / refine points, feature adaption, point_targets, bbox_targets, the three losses and parameter gradients."""
    from detectron2 import layers as d2l

    class Registry:
        def __init__(self, name):
            self.m = {}

        def register(self, obj=None):
            if obj is None:
                return lambda o: self.register(o)
            self.m[obj.__name__] = obj
            return obj

        def get(self, n):
            return self.m[n]

    _stub("detectron2.utils.registry", Registry=Registry)
    d2l.ShapeSpec = SimpleNamespace
    d2l.get_norm = lambda norm, c: torch.nn.GroupNorm(32, c) if norm == "GN" else None
    d2l.DeformConv = _OracleDeformConv
    d2l.cat = torch.cat
    _stub("refheads")
    _load("refheads.meta_head", "slender_det/modeling/meta_arch/meta/heads/meta_head.py", "refheads")
    _load("refheads.utils", "slender_det/modeling/meta_arch/meta/heads/utils.py", "refheads")
    psh = _load("refheads.pointset_head", "slender_det/modeling/meta_arch/meta/heads/pointset_head.py", "refheads")
    meta = {}
    C = 32
    hw = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    strides = [8, 16, 32, 64, 128]
    cases = [("empty", "Empty", True, "minmax", g), ("sup", "Supervised Offset", True, "minmax", g), ("unsup", "Unsupervised Offset", False, "minmax", g)]
    # the other two TRANSFORM_METHODs (pointset_head.py:322-343), on their own generators so that the fixtures above keep their bytes
    cases += [("partial", "Empty", True, "partial_minmax", torch.Generator().manual_seed(2101)),
              ("moment", "Empty", False, "moment", torch.Generator().manual_seed(2102))]
    for tag, fa, res, method, g in cases:
        head_params = SimpleNamespace(
            IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], FPN_STRIDES=strides, NUM_CLASSES=80, FEAT_CHANNELS=C, STACK_CONVS=3, NORM="GN",
            FEAT_ADAPTION=fa, RES_REFINE=res, LOC_FEAT_CHANNELS=C, GRADIENT_MUL=0.1, PRIOR_PROB=0.01, FOCAL_LOSS_GAMMA=2.0,
            FOCAL_LOSS_ALPHA=0.25, LOSS_CLS_WEIGHT=1.0, LOSS_LOC_INIT_WEIGHT=0.5, LOSS_LOC_REFINE_WEIGHT=1.0, SCORE_THRESH_TEST=0.05,
            TOPK_CANDIDATES_TEST=1000, NMS_THRESH_TEST=0.5, NUM_POINTS=9, POINT_BASE_SCALE=4, TRANSFORM_METHOD=method, MOMENT_MUL=0.01)
        cfg = SimpleNamespace(MODEL=SimpleNamespace(META_ARCH=head_params), TEST=SimpleNamespace(DETECTIONS_PER_IMAGE=100))
        torch.manual_seed(7)
        head = psh.PointSetHead(cfg, [SimpleNamespace(channels=C, stride=s) for s in strides])
        with torch.no_grad():      # larger-than-init weights so that boxes / IoUs are not degenerate; fp16-representable values on disk
            for n, p in head.named_parameters():
                if n.endswith("weight") and p.dim() == 4 and "subnet" not in n:
                    p.mul_(6.0)
                p.copy_(p.half().float())
            if method == "moment":      # a non-trivial, fp16-representable moment_transfer
                head.moment_transfer.copy_(torch.tensor([0.375, -0.25]))
        head.train()
        feats = [(torch.randn(2, C, h, w, generator=g) * 1.5).half().float() for h, w in hw]
        gtb = [random_boxes(g, 5, 128, 160), random_boxes(g, 8, 128, 160)]
        gtc = [torch.randint(0, 80, (len(b),), generator=g) for b in gtb]
        inst = [SimpleNamespace(gt_boxes=Boxes(b), gt_classes=c) for b, c in zip(gtb, gtc)]
        Boxes.to = lambda self, *a, **k: self
        images = SimpleNamespace(image_sizes=[(128, 160), (128, 160)])
        losses = head(images, feats, inst)
        tot = sum(losses.values())
        params = dict(head.named_parameters())
        grads = torch.autograd.grad(tot, list(params.values()), allow_unused=True)
        out = {"hw": np.array(hw), "strides": np.array(strides), "channels": np.array(C), "res_refine": np.array(res),
               "losses": np.array([float(losses[k]) for k in ("loss_cls", "loss_pts_init", "loss_pts_refine")])}
        for l, f in enumerate(feats):
            out[f"feat{l}"] = f.numpy().astype(np.float16)
        for i in range(2):
            out[f"gt_boxes{i}"], out[f"gt_classes{i}"] = gtb[i].numpy(), gtc[i].numpy()
        for (n, p), gr in zip(params.items(), grads):
            out["param:" + n] = p.detach().numpy().astype(np.float16)
            out["gradnorm:" + n] = np.array(0.0 if gr is None else float(gr.norm()))
        for n in ("cls_out.weight", "loc_refine_out.weight", "loc_init_out.weight") + (("moment_transfer",) if method == "moment" else ()):
            out["grad:" + n] = grads[list(params).index(n)].numpy()
        np.savez_compressed(os.path.join(OUT, f"pointset_head_{tag}.npz"), **out)
        meta[f"pointset_head_{tag}.npz"] = ("reference: meta/heads/pointset_head.py:19-470 + meta_head.py:21-104 + utils.py (reference Python"
                                            + (" x restated DeformConv" if fa != "Empty" else "") + " x restated focal / smooth-L1 / pairwise_iou)")
        print(tag, {k: float(v) for k, v in losses.items()})
    return meta


def anchor_head_golden(g, rm):
    """AnchorHead (meta/heads/anchor_head.py) built and run by the reference's own Python on CPU; detectron2 / fvcore pieces restated."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from oracle import losses as ol
    from oracle import rcnn as orc

    class B2B:
        def __init__(self, weights):
            self.weights = weights

        def get_deltas(self, src, tgt):
            return orc.get_deltas(src, tgt, self.weights)

        def apply_deltas(self, deltas, boxes):
            return orc.apply_deltas(deltas, boxes, self.weights)

    sizes = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
    ratios = [[0.5, 1.0, 2.0]]

    class AG:
        num_cell_anchors = [9] * 5

        def __call__(self, features):
            hw = [tuple(f.shape[-2:]) for f in features]
            return [Boxes(a) for a in orc.anchors(hw, [8, 16, 32, 64, 128], sizes, ratios)]

    Boxes.cat = classmethod(lambda cls, bl: cls(torch.cat([b.tensor for b in bl])))
    _stub("detectron2.modeling.box_regression", Box2BoxTransform=B2B)
    _stub("detectron2.modeling.anchor_generator", build_anchor_generator=lambda cfg, shp: AG())
    sys.modules["fvcore.nn"].giou_loss = lambda a, b, reduction="none": ol.giou_loss_xyxy(a, b, reduction)
    _stub("slender_det.modeling.matchers", nearest_point_match=rm.nearest_point_match)
    for n in ("refah", "refah.heads"):
        _stub(n)
    _load("refah.heads.meta_head", "slender_det/modeling/meta_arch/meta/heads/meta_head.py", "refah.heads")
    _load("refah.heads.utils", "slender_det/modeling/meta_arch/meta/heads/utils.py", "refah.heads")
    ah = _load("refah.heads.anchor_head", "slender_det/modeling/meta_arch/meta/heads/anchor_head.py", "refah.heads")
    meta = {}
    C = 32
    hw = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    strides = [8, 16, 32, 64, 128]
    for tag, fa, box_loss in (("none", "none", "giou"), ("sup", "supervised", "smooth_l1"), ("unsup", "unsupervised", "giou")):
        hp = SimpleNamespace(
            IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], FPN_STRIDES=strides, NUM_CLASSES=80, FEAT_CHANNELS=C, STACK_CONVS=3, NORM="GN",
            FEAT_ADAPTION=fa, RES_REFINE=False, LOC_FEAT_CHANNELS=C, GRADIENT_MUL=0.1, PRIOR_PROB=0.01, FOCAL_LOSS_GAMMA=2.0, FOCAL_LOSS_ALPHA=0.25,
            LOSS_CLS_WEIGHT=1.0, LOSS_LOC_INIT_WEIGHT=0.5, LOSS_LOC_REFINE_WEIGHT=1.0, SCORE_THRESH_TEST=0.05, TOPK_CANDIDATES_TEST=1000,
            NMS_THRESH_TEST=0.5, BBOX_REG_LOSS_TYPE=box_loss, BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), IOU_THRESHOLDS=[0.4, 0.5], IOU_LABELS=[0, -1, 1])
        cfg = SimpleNamespace(MODEL=SimpleNamespace(META_ARCH=hp), TEST=SimpleNamespace(DETECTIONS_PER_IMAGE=100))
        torch.manual_seed(13)
        head = ah.AnchorHead(cfg, [SimpleNamespace(channels=C, stride=s) for s in strides])
        with torch.no_grad():
            for n, p in head.named_parameters():
                if n.endswith("weight") and p.dim() == 4 and "subnet" not in n:
                    p.mul_(5.0)
                p.copy_(p.half().float())
        head.train()
        feats = [(torch.randn(2, C, h, w, generator=g) * 1.5).half().float() for h, w in hw]
        gtb = [random_boxes(g, 5, 128, 160), random_boxes(g, 8, 128, 160)]
        gtc = [torch.randint(0, 80, (len(b),), generator=g) for b in gtb]

        class _I(SimpleNamespace):
            def __len__(self):
                return len(self.gt_classes)

        sizes_img = [(120, 150), (128, 160)]
        inst = [_I(gt_boxes=Boxes(b), gt_classes=c_, image_size=s) for b, c_, s in zip(gtb, gtc, sizes_img)]
        images = SimpleNamespace(image_sizes=sizes_img)
        losses = head(images, feats, inst)
        params = dict(head.named_parameters())
        grads = torch.autograd.grad(sum(losses.values()), list(params.values()), allow_unused=True)
        keys = ("loss_cls", "loss_loc_init", "loss_loc_refine")
        out = {"hw": np.array(hw), "strides": np.array(strides), "channels": np.array(C), "losses": np.array([float(losses[k]) for k in keys]),
               "normalizer": np.array(float(head.loss_normalizer)), "image_sizes": np.array(sizes_img),
               "cfg": np.array(json.dumps(dict(fa=fa, box_loss=box_loss)))}
        for l, f in enumerate(feats):
            out[f"feat{l}"] = f.numpy().astype(np.float16)
        for i in range(2):
            out[f"gt_boxes{i}"], out[f"gt_classes{i}"] = gtb[i].numpy(), gtc[i].numpy()
        for (n, p), gr in zip(params.items(), grads):
            out["param:" + n] = p.detach().numpy().astype(np.float16)
            out["gradnorm:" + n] = np.array(0.0 if gr is None else float(gr.norm()))
        for n in ("loc_refine_out.weight", "loc_init_out.weight"):
            out["grad:" + n] = grads[list(params).index(n)].numpy()
        np.savez_compressed(os.path.join(OUT, f"anchor_head_{tag}.npz"), **out)
        meta[f"anchor_head_{tag}.npz"] = ("reference: meta/heads/anchor_head.py:25-434 + meta_head.py + matchers/rep_matcher.py (reference Python x "
                                          "restated anchor generator / Box2BoxTransform / Matcher / focal / giou / smooth-L1"
                                          + (" / DeformConv" if fa != "none" else "") + ")")
        print("anchor", tag, {k: float(v) for k, v in losses.items()}, float(head.loss_normalizer))
    return meta


def lrtb_head_golden(g):
    """LRTBHead (meta/heads/lrtb_head.py) built and run by the reference's own Python on CPU, three configurations."""
    iou_mod = _load("ref_iou_loss2", "slender_det/layers/iou_loss.py")
    scale_mod = _load("ref_scale2", "slender_det/layers/scale.py")
    sys.modules["slender_det.layers"] = types.ModuleType("slender_det.layers")
    sys.modules["slender_det.layers"].iou_loss, sys.modules["slender_det.layers"].Scale = iou_mod.iou_loss, scale_mod.Scale
    for n in ("refma", "refma.meta", "refma.meta.heads", "refma.fcos"):
        _stub(n)
    _load("refma.fcos.utils", "slender_det/modeling/meta_arch/fcos/utils.py", "refma.fcos")
    _load("refma.meta.heads.meta_head", "slender_det/modeling/meta_arch/meta/heads/meta_head.py", "refma.meta.heads")
    _load("refma.meta.heads.utils", "slender_det/modeling/meta_arch/meta/heads/utils.py", "refma.meta.heads")
    lh = _load("refma.meta.heads.lrtb_head", "slender_det/modeling/meta_arch/meta/heads/lrtb_head.py", "refma.meta.heads")
    meta = {}
    C = 32
    hw = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    strides = [8, 16, 32, 64, 128]
    cases = (("empty", dict(fa="Empty", res=False, norm_reg=True, ctr_on_loc=True, iou="giou", slender=False, radius=1.5)),
             ("sup", dict(fa="Supervised Offset", res=False, norm_reg=True, ctr_on_loc=True, iou="giou", slender=True, radius=1.5)),
             ("unsup", dict(fa="Unsupervised Offset", res=True, norm_reg=False, ctr_on_loc=False, iou="iou", slender=False, radius=0.0)),
             ("topk", dict(fa="Empty", res=False, norm_reg=True, ctr_on_loc=True, iou="giou", slender=False, radius=0.0, topk=True)))
    lt = _load("refma.meta.heads.lrtb_topk_head", "slender_det/modeling/meta_arch/meta/heads/lrtb_topk_head.py", "refma.meta.heads")
    for tag, c in cases:
        hp = SimpleNamespace(
            NAME="LRTBHead", IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], FPN_STRIDES=strides, NUM_CLASSES=80, FEAT_CHANNELS=C, STACK_CONVS=3, NORM="GN",
            FEAT_ADAPTION=c["fa"], RES_REFINE=c["res"], LOC_FEAT_CHANNELS=C, GRADIENT_MUL=0.1, PRIOR_PROB=0.01, FOCAL_LOSS_GAMMA=2.0,
            FOCAL_LOSS_ALPHA=0.25, LOSS_CLS_WEIGHT=1.0, LOSS_LOC_INIT_WEIGHT=0.5, LOSS_LOC_REFINE_WEIGHT=1.0, SCORE_THRESH_TEST=0.05,
            TOPK_CANDIDATES_TEST=1000, NMS_THRESH_TEST=0.5, NUM_POINTS=2, CENTER_SAMPLING_RADIUS=c["radius"], NORM_REG_TARGETS=c["norm_reg"],
            CENTERNESS_ON_LOC=c["ctr_on_loc"], IOU_LOSS_TYPE=c["iou"], PRE_NMS_THRESH=0.05, PRE_NMS_TOP_N=1000, SLENDER_CENTERNESS=c["slender"])
        cfg = SimpleNamespace(MODEL=SimpleNamespace(META_ARCH=hp), TEST=SimpleNamespace(DETECTIONS_PER_IMAGE=100))
        torch.manual_seed(11)
        head = (lt.LRTBTopkHead if c.get("topk") else lh.LRTBHead)(cfg, [SimpleNamespace(channels=C, stride=s) for s in strides])
        with torch.no_grad():
            for n, p in head.named_parameters():
                if n.endswith("weight") and p.dim() == 4 and "subnet" not in n:
                    p.mul_(4.0)
                if "scales" in n:
                    p.add_(torch.rand(p.shape, generator=g) * 0.4 - 0.2)
                if n in ("loc_init_out.bias", "loc_refine_out.bias"):
                    p.fill_(0.75)      # keeps relu(z) * stride (NORM_REG_TARGETS) away from the all-zero box
                p.copy_(p.half().float())
        head.train()
        feats = [(torch.randn(2, C, h, w, generator=g) * 1.5).half().float() for h, w in hw]
        gtb = [random_boxes(g, 5, 128, 160), random_boxes(g, 8, 128, 160)]
        gtc = [torch.randint(0, 80, (len(b),), generator=g) for b in gtb]
        inst = [SimpleNamespace(gt_boxes=Boxes(b), gt_classes=c_, __len__=None) for b, c_ in zip(gtb, gtc)]

        class _I(SimpleNamespace):
            def __len__(self):
                return len(self.gt_classes)

        inst = [_I(gt_boxes=Boxes(b), gt_classes=c_) for b, c_ in zip(gtb, gtc)]
        images = SimpleNamespace(image_sizes=[(128, 160), (128, 160)])
        losses = head(images, feats, inst)
        params = dict(head.named_parameters())
        grads = torch.autograd.grad(sum(losses.values()), list(params.values()), allow_unused=True)
        keys = ("loss_cls", "centerness_loss", "loss_loc_init", "loss_loc_refine")
        out = {"hw": np.array(hw), "strides": np.array(strides), "channels": np.array(C), "losses": np.array([float(losses[k]) for k in keys]),
               "cfg": np.array(json.dumps(c))}
        for l, f in enumerate(feats):
            out[f"feat{l}"] = f.numpy().astype(np.float16)
        for i in range(2):
            out[f"gt_boxes{i}"], out[f"gt_classes{i}"] = gtb[i].numpy(), gtc[i].numpy()
        for (n, p), gr in zip(params.items(), grads):
            out["param:" + n] = p.detach().numpy().astype(np.float16)
            out["gradnorm:" + n] = np.array(0.0 if gr is None else float(gr.norm()))
        for n in ("cls_out.weight", "loc_refine_out.weight", "loc_init_out.weight", "ctn_out.weight"):
            out["grad:" + n] = grads[list(params).index(n)].numpy()
        np.savez_compressed(os.path.join(OUT, f"lrtb_head_{tag}.npz"), **out)
        meta[f"lrtb_head_{tag}.npz"] = ("reference: meta/heads/lrtb_head.py:24-258 + meta_head.py + heads/utils.py + fcos/utils.py + layers/iou_loss.py"
                                        " (reference Python" + (" x restated DeformConv" if c["fa"] != "Empty" else "") + " x restated focal)")
        print("lrtb", tag, {k: float(v) for k, v in losses.items()})
    return meta


def main():
    assert os.path.isdir(REF), "runs only in the build container (needs /root/reference)"
    rm, rpd = install()
    g = torch.Generator().manual_seed(20240)
    meta = {}
    hw = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    strides = [8, 16, 32, 64, 128]
    cs, ss = grid(hw, strides)
    centers, st = torch.cat(cs), torch.cat(ss)

    # ------------------------------------------------------------------ matchers: pure reference
    out = {"hw": np.array(hw), "strides": np.array(strides)}
    cases = [random_boxes(g, n, 256, 320) for n in (1, 3, 7, 15, 30)]
    cases.append(torch.tensor([[100.0, 100.0, 103.0, 104.0]]))            # tiny box: nothing "inside" -> nearest fallback
    cases.append(torch.tensor([[10.0, 10.0, 74.0, 74.0], [12.0, 12.0, 76.0, 76.0], [40.0, 40.0, 45.0, 300.0]]))   # competing + slender
    out["num_cases"] = np.array(len(cases))
    for i, b in enumerate(cases):
        out[f"boxes{i}"] = b.numpy()
        for name, fn in (("points", rm.rep_points_match), ("nearest_points", rm.nearest_point_match), ("inside", rm.inside_match)):
            o, l = fn(centers, st, Boxes(b))
            out[f"{name}_obj{i}"] = o.numpy().astype(np.int8)
            out[f"{name}_box{i}"] = l.numpy()
    np.savez_compressed(os.path.join(OUT, "reppoints_matchers.npz"), **out)
    meta["reppoints_matchers.npz"] = "reference: matchers/rep_matcher.py:9-101,199-248 + structures/points.py:6-45 (pure reference Python)"

    # ------------------------------------------------------------------ points2bbox / get_ground_truth / losses
    R = rpd.RepPointsDetector
    N, K = 2, 80
    X = centers.shape[0]
    self = SimpleNamespace(transform_method="minmax", num_classes=K, matcher=rm.rep_points_match,
                           bbox_matcher=Matcher([0.4, 0.5], [0, -1, 1], allow_low_quality_matches=True),
                           loss_normalizer=20, loss_normalizer_momentum=0.9, focal_loss_alpha=0.25, focal_loss_gamma=2.0)
    oi = [torch.randn(N, 18, h, w, generator=g) * 1.5 for h, w in hw]
    orf = [o + torch.randn(o.shape, generator=g) * 0.5 for o in oi]
    pc = [c.clone() for c in cs]
    init_boxes = rpd.flat_and_concate_levels(R.points2bbox(self, pc, oi, [1, 2, 4, 8, 16]))
    refine_boxes = rpd.flat_and_concate_levels(R.points2bbox(self, pc, orf, [1, 2, 4, 8, 16]))
    gtb = [random_boxes(g, 6, 250, 300), random_boxes(g, 11, 256, 320)]
    gtc = [torch.randint(0, K, (len(b),), generator=g) for b in gtb]
    sizes = [(250, 300), (256, 320)]
    inst = [SimpleNamespace(image_size=s, gt_boxes=Boxes(b), gt_classes=c) for s, b, c in zip(sizes, gtb, gtc)]
    res = {"hw": np.array(hw), "strides": np.array(strides), "image_sizes": np.array(sizes),
           "init_boxes": init_boxes.numpy(), "refine_boxes": refine_boxes.numpy()}
    for l in range(len(hw)):
        res[f"oi{l}"], res[f"or{l}"] = oi[l].numpy(), orf[l].numpy()
    for i in range(N):
        res[f"gt_boxes{i}"], res[f"gt_classes{i}"] = gtb[i].numpy(), gtc[i].numpy()
    logits = torch.randn(N, X, K, generator=g) * 2 - 3
    res["logits"] = logits.numpy().astype(np.float16)         # inputs are the fp16-rounded values (kept small on disk)
    logits = torch.from_numpy(res["logits"]).float()
    for mode, fn in (("points", rm.rep_points_match), ("nearest_points", rm.nearest_point_match), ("inside", rm.inside_match)):
        self.matcher = fn
        self.loss_normalizer = 20
        tg = R.get_ground_truth.__wrapped__(self, centers, st, init_boxes, inst) if hasattr(R.get_ground_truth, "__wrapped__") \
            else R.get_ground_truth(self, centers, st, init_boxes, inst)
        obj, ib, cl, rb = tg
        res[f"{mode}_obj"], res[f"{mode}_init"] = obj.numpy().astype(np.int8), ib.numpy()
        res[f"{mode}_cls"], res[f"{mode}_refine"] = cl.numpy().astype(np.int16), rb.numpy()
        lg = logits.clone().requires_grad_(True)
        b1 = init_boxes.clone().requires_grad_(True)
        b2 = refine_boxes.clone().requires_grad_(True)
        d = R.losses(self, lg, b1, b2, obj, ib, cl, rb, st)
        tot = d["loss_cls"] + d["loss_localization_init"] + d["loss_localization_refine"]
        gl, g1, g2 = torch.autograd.grad(tot, (lg, b1, b2))
        res[f"{mode}_losses"] = np.array([float(d[k]) for k in ("loss_cls", "loss_localization_init", "loss_localization_refine")])
        res[f"{mode}_normalizer"] = np.array(float(self.loss_normalizer))
        res[f"{mode}_grad_init"], res[f"{mode}_grad_refine"] = g1.numpy(), g2.numpy()
        res[f"{mode}_grad_logits_sum"] = gl.sum(-1).numpy()
    np.savez_compressed(os.path.join(OUT, "reppoints_losses.npz"), **res)
    meta["reppoints_losses.npz"] = ("reference Python (rpd.py:221-402, rep_matcher.py) x restated third-party ops "
                                    "(pairwise_iou, Matcher, sigmoid_focal_loss_jit, smooth_l1_loss)")
    meta.update(pointset_head_golden(g))
    meta.update(lrtb_head_golden(g))
    meta.update(anchor_head_golden(g, rm))
    # ------------------------------------------------------------------ TopKMatcher: pure reference
    sys.modules["detectron2.layers"].nonzero_tuple = lambda x: x.nonzero(as_tuple=True)
    tk = _load("ref_topk_matcher", "slender_det/modeling/matchers/topk_matcher.py")
    gtb = random_boxes(g, 9, 256, 320)
    anc = random_boxes(g, 900, 256, 320)
    q = pairwise_iou(Boxes(gtb), Boxes(anc))
    m, lab = tk.TopKMatcher([0.3, 0.7], [0, -1, 1], 10)(q)
    np.savez_compressed(os.path.join(OUT, "topk_matcher.npz"), gt=gtb.numpy(), anchors=anc.numpy(), quality=q.numpy(), matches=m.numpy(),
                        labels=lab.numpy())
    meta["topk_matcher.npz"] = "reference: matchers/topk_matcher.py:7-85 (pure reference Python on a restated pairwise_iou matrix)"
    mp = os.path.join(OUT, "meta.json")
    old = json.load(open(mp)) if os.path.exists(mp) else {}
    old.update(meta)
    json.dump(old, open(mp, "w"), indent=1, sort_keys=True)
    for k in meta:
        print(k, os.path.getsize(os.path.join(OUT, k)))


if __name__ == "__main__":
    main()
