"""Oracle RepPointsDetector (CPU, fp32, NCHW, plain torch) — TEST INFRASTRUCTURE, never imported by the product path.

Restates slender_det/modeling/meta_arch/reppoints/rpd.py (forward :589-681, points2bbox :221-249, get_ground_truth :276-333,
losses :335-402, inference :701-789), the init-box matchers of slender_det/modeling/matchers/rep_matcher.py (:9-101, :199-223,
:226-248) and structures/points.py:6-45.  Pinned against tests/golden/reppoints_*.npz, generated from the reference's own
Python by tests/golden/make_golden_reppoints.py (matchers: pure reference; get_ground_truth / losses: reference Python x
restated detectron2 ``pairwise_iou`` / ``Matcher`` / fvcore losses, which are absent everywhere: "parity unpinned" for those).
"""
import math

import torch
import torch.nn.functional as F

from . import detection as od
from . import losses as ol
from .deform_conv import deform_conv2d
from .model import OracleFCOS, _RoundSTE
from .nn import relu as _band_relu      # torch.relu unless a ReluBand context is active (oracle/nn.py)
from .nn import relu_at as _relu_at       # ... or teacher-forced decisions for this position (ForcedMasks)


# ------------------------------------------------------------------------------------------------ structures/points.py
def pairwise_dist(points, boxes):
    """points.py:6-28: centre distance normalised by the box size, (P, M)."""
    centers = (boxes[:, :2] + boxes[:, 2:]) / 2
    wh = boxes[:, 2:] - boxes[:, :2]
    return ((points[:, None] - centers[None]) / wh[None]).norm(dim=2)


def stride_match(strides, boxes):
    """points.py:31-45."""
    wh = boxes[:, 2:] - boxes[:, :2]
    e = ((torch.log2(wh[:, 0]) + torch.log2(wh[:, 1])) / 2).int()
    box_strides = torch.pow(2.0, e.float()).clamp(strides.min(), strides.max())
    return torch.eq(strides[:, None], box_strides[None])


# ------------------------------------------------------------------------------------------------ matchers/rep_matcher.py
def rep_points_match(centers, strides, boxes, scale=4, pos_num=1):
    """rep_matcher.py:9-101."""
    if centers.shape[0] == 0 or boxes.shape[0] == 0:
        raise ValueError("No gt or bboxes")
    P = centers.shape[0]
    lvl = torch.log2(strides).int()
    lmin, lmax = int(lvl.min()), int(lvl.max())
    cxy = (boxes[:, :2] + boxes[:, 2:]) / 2
    wh = (boxes[:, 2:] - boxes[:, :2]).clamp(min=1e-6)
    glvl = ((torch.log2(wh[:, 0] / scale) + torch.log2(wh[:, 1] / scale)) / 2).int().clamp(lmin, lmax)
    assigned = torch.zeros(P, dtype=torch.long)
    adist = torch.full((P,), float("inf"))
    arange = torch.arange(P)
    for g in range(boxes.shape[0]):
        sel = lvl == glvl[g]
        index = arange[sel]
        d = ((centers[sel] - cxy[g:g + 1]) / wh[g:g + 1]).norm(dim=1)
        md, mi = torch.topk(d, pos_num, largest=False)
        pts = index[mi]
        better = md < adist[pts]
        pts = pts[better]
        assigned[pts] = g + 1
        adist[pts] = md[better]
    labels = (assigned > 0).long()
    out = torch.zeros(P, 4)
    pos = assigned > 0
    out[pos] = boxes[assigned[pos] - 1]
    return labels, out


def nearest_point_match(centers, strides, boxes):
    """rep_matcher.py:199-223."""
    obj = torch.zeros(centers.shape[0])
    lab = torch.zeros(centers.shape[0], 4)
    D = pairwise_dist(centers, boxes) + (~stride_match(strides, boxes)) * 1e5
    gmin, gidx = D.min(0)
    pmin, _ = D.min(1)
    lost = pmin.gather(0, gidx) < gmin
    for g in range(boxes.shape[0]):
        if lost[g]:
            continue
        obj[gidx[g]] = 1
        lab[gidx[g]] = boxes[g]
    return obj, lab


def inside_match(centers, strides, boxes):
    """rep_matcher.py:226-248."""
    upper = centers + strides[:, None]
    inside = ((upper[:, None, 0] >= boxes[None, :, 0]) & (upper[:, None, 1] >= boxes[None, :, 1])
              & (centers[:, None, 0] <= boxes[None, :, 2]) & (centers[:, None, 1] <= boxes[None, :, 3]))
    inside = (inside & stride_match(strides, boxes)).any(1)
    if not bool(inside.any()):
        return nearest_point_match(centers, strides, boxes)
    obj = inside.float()
    return obj, boxes[pairwise_dist(centers, boxes).argmin(1)]


MATCHERS = {"points": rep_points_match, "nearest_points": nearest_point_match, "inside": inside_match}


# ------------------------------------------------------------------------------------------------ rpd.py pieces
def center_grid(hw, strides):
    """rpd.py:206-219 concatenated: centers (X,2) = (j, i)*stride, strides (X,)."""
    cs, ss = [], []
    for (h, w), s in zip(hw, strides):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        cs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), 1) * s)
        ss.append(torch.full((h * w,), float(s)))
    return torch.cat(cs), torch.cat(ss)


def points2bbox(deltas, hw, strides, point_strides):
    """rpd.py:221-249 ("minmax"): deltas list of (N, 2P, H, W) with channel 2k = x, 2k+1 = y -> (N, X, 4)."""
    out = []
    for d, (h, w), s, ps in zip(deltas, hw, strides, point_strides):
        N = d.shape[0]
        gy, gx = torch.meshgrid(torch.arange(h, dtype=d.dtype), torch.arange(w, dtype=d.dtype), indexing="ij")
        pts = d.view(N, -1, 2, h, w) * ps
        x = pts[:, :, 0] + gx * s
        y = pts[:, :, 1] + gy * s
        box = torch.stack((x.min(1)[0], y.min(1)[0], x.max(1)[0], y.max(1)[0]), 1)
        out.append(box.view(N, 4, -1).permute(0, 2, 1))
    return torch.cat(out, 1)


def get_ground_truth(centers, strides, init_boxes, gt_boxes, gt_classes, image_sizes, num_classes, mode="points",
                     thresholds=(0.4, 0.5), labels=(0, -1, 1)):
    """rpd.py:276-333.  Note (:316-318): only matcher label 0 is rewritten to background; label -1 keeps the gt class."""
    objs, init_labs, clss, refs = [], [], [], []
    for i, (b, c) in enumerate(zip(gt_boxes, gt_classes)):
        h, w = image_sizes[i]
        invalid = (centers[:, 0] >= w) | (centers[:, 1] >= h)
        obj, lab = MATCHERS[mode](centers, strides, b)
        obj = obj.clone().float()
        obj[invalid] = 0
        q = od.pairwise_iou(b, init_boxes[i].detach())
        midx, mlab = od.matcher(q, list(thresholds), list(labels), True)
        cl = c[midx].clone().long()
        cl[mlab == 0] = num_classes
        cl[invalid] = -1
        objs.append(obj); init_labs.append(lab); clss.append(cl); refs.append(b[midx])
    return torch.stack(objs), torch.stack(init_labs), torch.stack(clss), torch.stack(refs)


def losses(logits, init_boxes, refine_boxes, gt_obj, gt_init, gt_cls, gt_refine, strides, num_classes, alpha, gamma, normalizer,
           momentum=0.9, beta=0.11):
    """rpd.py:335-402.  Returns (dict, new_normalizer)."""
    valid = gt_cls >= 0
    fg = valid & (gt_cls != num_classes)
    num_fg = float(fg.sum()) / gt_init.shape[0]
    target = torch.zeros_like(logits)
    target[fg, gt_cls[fg]] = 1
    normalizer = momentum * normalizer + (1 - momentum) * num_fg
    loss_cls = ol.sigmoid_focal_loss(logits[valid], target[valid], alpha, gamma, "sum") / max(1, normalizer)
    ifg = gt_obj > 0
    st = strides[None].repeat(logits.shape[0], 1)
    n1 = st[ifg].unsqueeze(-1) * 4
    loss_init = ol.smooth_l1_loss(init_boxes[ifg] / n1, gt_init[ifg] / n1, beta, "sum") / max(1, float(gt_obj.sum())) * 0.5
    n2 = st[fg].unsqueeze(-1) * 4
    loss_refine = ol.smooth_l1_loss(refine_boxes[fg] / n2, gt_refine[fg] / n2, beta, "sum") / max(1, normalizer)
    return {"loss_cls": loss_cls, "loss_localization_init": loss_init, "loss_localization_refine": loss_refine}, normalizer


def inference_single_image(logits, init_boxes, refine_boxes, bounds, topk, score_thr, nms_thr, max_det):
    """rpd.py:717-789 for one image: logits (X,K), boxes (X,4), bounds = level starts + [X]."""
    B, I, C, S = [], [], [], []
    for l in range(len(bounds) - 1):
        sl = slice(bounds[l], bounds[l + 1])
        scores, cls = logits[sl].sigmoid().max(1)
        prob, idx = scores.sort(descending=True)
        k = min(topk, cls.shape[0])
        prob, idx = prob[:k], idx[:k]
        idx = idx[prob > score_thr]
        B.append(refine_boxes[sl][idx]); I.append(init_boxes[sl][idx]); C.append(cls[idx]); S.append(scores[idx])
    B, I, C, S = torch.cat(B), torch.cat(I), torch.cat(C), torch.cat(S)
    keep = od.batched_nms(B, S, C, nms_thr)[:max_det]
    return B[keep], S[keep], C[keep], I[keep]


# ------------------------------------------------------------------------------------------------ whole model
class OracleRepPoints(OracleFCOS):
    """Functional RepPointsDetector over torch-layout fp32 tensors; reuses the ResNet bottom-up of OracleFCOS."""

    @classmethod
    def from_hip_model(cls, model, emulate_bf16=False):
        from slenderobjdet_amd.layers.deform_conv import DeformConv
        from slenderobjdet_amd.layers.nn import HipConv2d, HipGroupNorm

        params, buffers = {}, {}
        for name, m in model.named_modules():
            if isinstance(m, (HipConv2d, DeformConv)):
                params[name + ".weight"] = m.weight.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(m.weight.requires_grad)
                if m.bias is not None:
                    params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(m.bias.requires_grad)
                if getattr(m, "frozen_bn", False):
                    scale = m.bn_weight.float().cpu() * torch.rsqrt(m.bn_running_var.float().cpu() + 1e-5)
                    buffers[name + ".scale"], buffers[name + ".shift"] = scale, m.bn_bias.float().cpu() - m.bn_running_mean.float().cpu() * scale
            elif isinstance(m, HipGroupNorm):
                params[name + ".weight"] = m.weight.detach().float().cpu().clone().requires_grad_(True)
                params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(True)
        bu = model.backbone.bottom_up
        res_names = [name for _, name in bu.stages_and_names]
        cfg_like = dict(
            blocks={n: len(getattr(bu, n)) for n in res_names}, bottleneck=any(k.endswith("conv3.weight") for k in params),
            num_classes=model.num_classes, strides=list(model.strides), point_scales=list(model.point_scales), alpha=model.focal_loss_alpha,
            gamma=model.focal_loss_gamma, mean=[float(v) for v in model.pixel_mean.flatten()], std=[float(v) for v in model.pixel_std.flatten()],
            size_div=model.backbone.size_divisibility, mode=model.sample_mode, npts=model.num_points, gmul=model.gradient_mul,
            thresholds=list(model.iou_thresholds), labels=list(model.iou_labels), fpn_in=list(model.backbone.in_features),
            fpn_norm=model.backbone.norm, in_features=list(model.in_features), top_in=model.backbone.top_block.in_feature,
            stride_in_1x1={n: [blk.conv1.stride for blk in getattr(bu, n)] for n in res_names},
            block_stride={n: [blk.stride for blk in getattr(bu, n)] for n in res_names},
            topk=model.topk_candidates, score_thr=model.score_threshold, nms_thr=model.nms_threshold, max_det=model.max_detections_per_image,
        )
        o = cls(params, buffers, cfg_like, emulate_bf16)
        o.normalizer = float(model.loss_normalizer.item())
        return o

    def _gn(self, name, x):
        return self._act(F.group_norm(x, 32, self.p[name + ".weight"], self.p[name + ".bias"], 1e-5))

    def _fpn_any(self, feats):
        """d2 FPN with optional GroupNorm after every lateral / output conv, P6/P7 from res5 or p5."""
        c = self.c
        gn = c["fpn_norm"] == "GN"
        prev, outs = None, {}
        for n in c["fpn_in"][::-1]:
            s = int(n[-1])
            lat = self._conv(f"backbone.fpn_lateral{s}", feats[n], 1, 0)
            if gn:
                lat = self._gn(f"backbone.fpn_lateral{s}.norm", lat)
            if prev is not None:
                lat = self._act(lat + F.interpolate(prev, scale_factor=2, mode="nearest"))
            prev = lat
            o = self._conv(f"backbone.fpn_output{s}", prev, 1, 1)
            outs[f"p{s}"] = self._gn(f"backbone.fpn_output{s}.norm", o) if gn else o
        src = feats[c["top_in"]] if c["top_in"] in feats else outs[c["top_in"]]
        p6 = self._conv("backbone.top_block.p6", src, 2, 1)
        outs["p6"] = p6
        outs["p7"] = self._conv("backbone.top_block.p7", self._act(_relu_at(p6, "backbone.top_block.p7:in")), 2, 1)
        return [outs[k] for k in c["in_features"]]

    def _tower3(self, prefix, x):
        for i in range(3):
            y = self._conv(f"{prefix}.{i}.conv", x, 1, 1)
            x = self._act(_relu_at(F.group_norm(y, 32, self.p[f"{prefix}.{i}.gn.weight"], self.p[f"{prefix}.{i}.gn.bias"], 1e-5), f"{prefix}.{i}.gn"))
        return x

    def _w(self, name):
        w = self.p[name]
        return _RoundSTE.apply(w) if self.emu else w

    def head(self, feats):
        """rpd.py:606-647 -> logits (N,X,K), init / refine point offsets per level (N,18,H,W)."""
        c = self.c
        n2 = 2 * c["npts"]
        ks = int(math.sqrt(c["npts"]))
        pad = (ks - 1) // 2
        base = torch.arange(-pad, pad + 1, dtype=torch.float32)
        base_off = torch.stack((base.repeat_interleave(ks), base.repeat(ks)), 1).reshape(1, -1, 1, 1)   # y-major (rpd.py:105-110)
        logits, oi_all, or_all = [], [], []
        for f in feats:
            N = f.shape[0]
            cf, rf = self._tower3("cls_conv", f), self._tower3("reg_conv", f)
            t = self._conv("offsets_init.0.conv", rf, 1, 1, relu=True)
            oi = F.conv2d(t, self._w("offsets_init.1.conv.weight")[:n2], self.p["offsets_init.1.conv.bias"][:n2])
            gm = (1 - c["gmul"]) * oi.detach() + c["gmul"] * oi
            off = gm.reshape(N, c["npts"], 2, *gm.shape[-2:]).flip(2).reshape(N, n2, *gm.shape[-2:]) - base_off
            hook = (lambda s: _RoundSTE.apply(s)) if self.emu else None
            dc = self._act(_relu_at(deform_conv2d(cf, off, self._w("deform_cls_conv.weight"), None, 1, pad, 1, sample_hook=hook), "deform_cls_conv"))
            dr = self._act(_relu_at(deform_conv2d(rf, off, self._w("deform_reg_conv.weight"), None, 1, pad, 1, sample_hook=hook), "deform_reg_conv"))
            lg = F.conv2d(dc, self._w("logits.weight"), self.p["logits.bias"])
            orf = F.conv2d(dr, self._w("offsets_refine.weight")[:n2], self.p["offsets_refine.bias"][:n2]) + oi.detach()
            logits.append(lg.permute(0, 2, 3, 1).reshape(N, -1, lg.shape[1]))
            oi_all.append(oi)
            or_all.append(orf)
        return torch.cat(logits, 1), oi_all, or_all

    def forward_boxes(self, batched_inputs):
        c = self.c
        x = self.preprocess(batched_inputs)
        feats = self._fpn_any(self._bottom_up(x))
        hw = [tuple(f.shape[2:]) for f in feats]
        logits, oi, orf = self.head(feats)
        init_boxes = points2bbox(oi, hw, c["strides"], c["point_scales"])
        refine_boxes = points2bbox(orf, hw, c["strides"], c["point_scales"])
        return logits, init_boxes, refine_boxes, hw

    def losses(self, batched_inputs, update_normalizer=True, targets=None):
        """``targets``: optional (objectness, init boxes, cls, refine boxes) override — the labels depend on the PREDICTED init
        boxes through an IoU arg-max, so gradient comparisons pin them to the labels of the run under test."""
        c = self.c
        logits, init_boxes, refine_boxes, hw = self.forward_boxes(batched_inputs)
        centers, strides = center_grid(hw, c["strides"])
        boxes = [b["instances"].gt_boxes.tensor.float().cpu() for b in batched_inputs]
        classes = [b["instances"].gt_classes.cpu() for b in batched_inputs]
        sizes = [tuple(b["image"].shape[-2:]) for b in batched_inputs]
        tg = targets if targets is not None else get_ground_truth(centers, strides, init_boxes, boxes, classes, sizes, c["num_classes"],
                                                                  c["mode"], c["thresholds"], c["labels"])
        self.last_targets = tg
        out, nrm = losses(logits, init_boxes, refine_boxes, *tg, strides, c["num_classes"], c["alpha"], c["gamma"], self.normalizer)
        if update_normalizer:
            self.normalizer = nrm
        return out

    def inference(self, batched_inputs):
        c = self.c
        with torch.no_grad():
            logits, init_boxes, refine_boxes, hw = self.forward_boxes(batched_inputs)
        bounds = [0]
        for h, w in hw:
            bounds.append(bounds[-1] + h * w)
        return [inference_single_image(logits[i], init_boxes[i], refine_boxes[i], bounds, c["topk"], c["score_thr"], c["nms_thr"], c["max_det"])
                for i in range(logits.shape[0])]
