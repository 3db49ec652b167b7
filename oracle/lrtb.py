"""Oracle LRTBHead (CPU, fp32, NCHW, plain torch) — TEST INFRASTRUCTURE, never imported by the product path.

Restates slender_det/modeling/meta_arch/meta/heads/lrtb_head.py (``_forward`` :123-188, ``losses`` :190-258, inference :283-375)
with meta_head.py:21-104, heads/utils.py:14-23 (``grad_mul``, ``lrtb_to_points``) and the FCOS target / centerness helpers of
fcos/utils.py that oracle/fcos_targets.py and oracle/losses.py already pin.  Pinned against tests/golden/lrtb_head_*.npz, produced by
the reference's own head built and run on CPU (tests/golden/make_golden_reppoints.py).
"""
import torch
import torch.nn.functional as F

from . import detection as od
from . import fcos_targets as ot
from . import losses as ol
from .deform_conv import deform_conv2d
from .model import _RoundSTE
from .pointset import OraclePointSetHead


def slender_centerness_targets(reg):
    """fcos/utils.py:302-312."""
    lr, tb = reg[:, [0, 2]], reg[:, [1, 3]]
    r1 = (reg[:, 0] + reg[:, 2]) / (reg[:, 1] + reg[:, 3])
    ratio = torch.stack((r1, 1 / r1), 1).min(1)[0]
    c = (lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0])
    return torch.pow(c, 0.5 * ratio)


def topk_locations(level_hw, strides, gt_boxes, gt_classes, radius, num_classes, topk=5):
    """Selection part of compute_topk_targets_for_locations (fcos/utils.py:215-292): per gt box the top-k positive locations by
    centerness (all of them when there are at most k).  Returns labels (N,L), reg (N,L,4), mask (N,L) bool."""
    locs = ot.locations(level_hw, strides)
    pts = [len(l) for l in locs]
    allp = torch.cat(locs)
    labs, regs, masks = [], [], []
    for b, c in zip(gt_boxes, gt_classes):
        lab, reg, idx = ot.targets_for_image(allp, pts, strides, b.float(), c, radius, num_classes, return_inds=True)
        fg = (lab >= 0) & (lab != num_classes)
        m = torch.zeros(len(lab), dtype=torch.bool)
        for g in range(len(b)):
            sel = (idx == g) & fg
            n = int(sel.sum())
            if n > topk:
                score = ol.centerness_targets(reg[sel])
                _, inds = torch.topk(score, topk, sorted=False)
                m[sel.nonzero()[inds]] = True
            elif n > 0:
                m[sel.nonzero()] = True
        labs.append(lab); regs.append(reg); masks.append(m)
    return torch.stack(labs), torch.stack(regs), torch.stack(masks)


def losses(labels, reg_t, cls, ctr, init, refine, num_classes, alpha=0.25, gamma=2.0, iou_type="iou", slender=False, w=(1.0, 0.5, 1.0), world=1,
           topk_mask=None):
    """lrtb_head.py:190-258 on flattened predictions (M rows); ``topk_mask`` (M,) bool = lrtb_topk_head.py:234-243: the init loss runs
    over those rows with the STANDARD centerness as weight and its sum as normaliser."""
    fg = (labels >= 0) & (labels != num_classes)
    pos_avg = max(int(fg.sum()) / float(world), 1.0)
    onehot = ol.one_hot_from_labels(labels, num_classes).to(cls.dtype)
    loss_cls = ol.sigmoid_focal_loss(cls, onehot, alpha, gamma, "sum") / pos_avg
    if int(fg.sum()) > 0:
        ct = slender_centerness_targets(reg_t[fg]) if slender else ol.centerness_targets(reg_t[fg])
        s = float(ct.sum()) / float(world)
        if topk_mask is None:
            l_init = ol.iou_loss_ltrb(init[fg], reg_t[fg], ct, iou_type) / s
        else:
            ct_k = ol.centerness_targets(reg_t[topk_mask])
            l_init = ol.iou_loss_ltrb(init[topk_mask], reg_t[topk_mask], ct_k, iou_type) / (float(ct_k.sum()) / float(world))
        l_ref = ol.iou_loss_ltrb(refine[fg], reg_t[fg], ct, iou_type) / s
        l_ctr = F.binary_cross_entropy_with_logits(ctr[fg], ct, reduction="sum") / pos_avg
    else:
        l_init, l_ref, l_ctr = init[fg].sum(), refine[fg].sum(), ctr[fg].sum()
    return {"loss_cls": loss_cls * w[0], "centerness_loss": l_ctr * w[0], "loss_loc_init": l_init * w[1], "loss_loc_refine": l_ref * w[2]}


def inference_single_image(locs_l, cls, ctr, refine, bounds, score_thr, topk, nms_thr, max_det):
    """lrtb_head.py:318-375: cls (L,K), ctr (L,), refine (L,4), locs_l per level (HW,2)."""
    B, S, C = [], [], []
    for l in range(len(bounds) - 1):
        sl = slice(bounds[l], bounds[l + 1])
        p = cls[sl].sigmoid()
        keep = p > score_thr
        p = p * ctr[sl].sigmoid()[:, None]
        sc = p[keep]
        idx = keep.nonzero()
        li, ci = idx[:, 0], idx[:, 1]
        reg, loc = refine[sl][li], locs_l[l][li]
        n = int(keep.sum())
        if n > min(n, topk):
            sc, ti = sc.topk(min(n, topk), sorted=False)
            ci, reg, loc = ci[ti], reg[ti], loc[ti]
        B.append(torch.stack([loc[:, 0] - reg[:, 0], loc[:, 1] - reg[:, 1], loc[:, 0] + reg[:, 2], loc[:, 1] + reg[:, 3]], 1))
        S.append(torch.sqrt(sc)); C.append(ci)
    B, S, C = torch.cat(B), torch.cat(S), torch.cat(C)
    keep = od.batched_nms(B, S, C, nms_thr)[:max_det]
    return B[keep], S[keep], C[keep]


class OracleLRTBHead(OraclePointSetHead):
    """Functional LRTBHead; parameter names follow the product module (cls_pred / box_pred hold the fused cls+ctn / refine+ctn rows)."""

    @classmethod
    def from_reference_arrays(cls, arrays, cfg):
        ref = {k[len("param:"):]: torch.tensor(v.astype("float32")).requires_grad_(True) for k, v in arrays.items() if k.startswith("param:")}
        p = {}
        for tower in ("cls_subnet", "loc_subnet"):
            for i in range(3):
                p[f"{tower}.{i}.conv.weight"], p[f"{tower}.{i}.conv.bias"] = ref[f"{tower}.{3 * i}.weight"], ref[f"{tower}.{3 * i}.bias"]
                p[f"{tower}.{i}.gn.weight"], p[f"{tower}.{i}.gn.bias"] = ref[f"{tower}.{3 * i + 1}.weight"], ref[f"{tower}.{3 * i + 1}.bias"]
        fa = cfg["fa"]
        ren = {"loc_init_conv": "loc_init_conv.conv", "loc_init_out": "loc_init_out.conv", "offset_conv": "offset_conv.conv",
               "offset_conv_cls": "offset_conv_cls.conv", "offset_conv_loc": "offset_conv_loc.conv", "offset_conv_extend": "offset_conv_extend.conv",
               "cls_conv": "cls_conv.conv" if fa == "Empty" else "cls_conv", "loc_refine_conv": "loc_refine_conv.conv" if fa == "Empty" else "loc_refine_conv"}
        for k, v in ref.items():
            base, leaf = k.rsplit(".", 1)
            if base in ren:
                p[f"{ren[base]}.{leaf}"] = v
        # fused prediction convs of the product path: rows [cls_out; ctn_out] and [loc_refine_out; ctn_out]
        ctn_w, ctn_b = ref["ctn_out.weight"], ref["ctn_out.bias"]
        if cfg["ctr_on_loc"]:
            p["cls_pred.conv.weight"], p["cls_pred.conv.bias"] = ref["cls_out.weight"], ref["cls_out.bias"]
            p["box_pred.conv.weight"] = torch.cat((ref["loc_refine_out.weight"], ctn_w))
            p["box_pred.conv.bias"] = torch.cat((ref["loc_refine_out.bias"], ctn_b))
        else:
            p["cls_pred.conv.weight"], p["cls_pred.conv.bias"] = torch.cat((ref["cls_out.weight"], ctn_w)), torch.cat((ref["cls_out.bias"], ctn_b))
            p["box_pred.conv.weight"], p["box_pred.conv.bias"] = ref["loc_refine_out.weight"], ref["loc_refine_out.bias"]
        p["scales_init"] = torch.stack([ref[f"scales_init.{i}.scale"].reshape(()) for i in range(5)])
        p["scales_refine"] = torch.stack([ref[f"scales_refine.{i}.scale"].reshape(()) for i in range(5)])
        return cls(p, dict(cfg))

    @classmethod
    def from_hip_head(cls, head, emulate_bf16=False):
        from slenderobjdet_amd.layers.deform_conv import DeformConv
        from slenderobjdet_amd.layers.nn import HipConv2d, HipGroupNorm

        p = {}
        for name, m in head.named_modules():
            if isinstance(m, (HipConv2d, DeformConv)):
                p[name + ".weight"] = m.weight.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
                if m.bias is not None:
                    p[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(True)
            elif isinstance(m, HipGroupNorm):
                p[name + ".weight"] = m.weight.detach().float().cpu().clone().requires_grad_(True)
                p[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(True)
        p["scales_init"] = head.scales_init.detach().float().cpu().clone().requires_grad_(True)
        p["scales_refine"] = head.scales_refine.detach().float().cpu().clone().requires_grad_(True)
        cfg = dict(fa=head.feat_adaption, res=head.res_refine, K=head.num_classes, gmul=head.gradient_mul, strides=list(head.fpn_strides),
                   norm_reg=head.norm_reg_targets, ctr_on_loc=head.centerness_on_loc, iou_type=head.iou_loss_type, slender=head.slender_centerness,
                   radius=head.center_sampling_radius, w=(head.loss_cls_weight, head.loss_loc_init_weight, head.loss_loc_refine_weight),
                   alpha=head.focal_loss_alpha, gamma=head.focal_loss_gamma)
        return cls(p, cfg, emulate_bf16)

    def _decode(self, raw, scale, stride):
        z = raw * scale
        return torch.relu(z) * stride if self.c["norm_reg"] else torch.exp(z)

    def forward(self, feats):
        """-> cls (N,L,K), ctr (N,L), init (N,L,4), refine (N,L,4), hw."""
        c = self.c
        K = c["K"]
        hook = (lambda s: _RoundSTE.apply(s)) if self.emu else None
        C, T, I, R = [], [], [], []
        for l, f in enumerate(feats):
            N = f.shape[0]
            cf, lf = self._tower("cls_subnet", f), self._tower("loc_subnet", f)
            raw = self._conv("loc_init_out.conv", self._conv("loc_init_conv.conv", lf, 1, relu=True), 0, rows=4, f32_out=True)
            init = self._decode(raw, self.p["scales_init"][l], c["strides"][l])
            if c["fa"] == "Empty":
                cfa, lfa = self._conv("cls_conv.conv", cf, 1, relu=True), self._conv("loc_refine_conv.conv", lf, 1, relu=True)
            else:
                if c["fa"] == "Unsupervised Offset":
                    oc = ol_ = self._conv("offset_conv.conv", lf, 0, rows=18, f32_out=True)
                elif c["fa"] == "Split Unsup Offset":
                    oc = self._conv("offset_conv_cls.conv", lf, 0, rows=18, f32_out=True)
                    ol_ = self._conv("offset_conv_loc.conv", lf, 0, rows=18, f32_out=True)
                else:
                    gm = (1 - c["gmul"]) * init.detach() + c["gmul"] * init
                    l_, r_, t_, b_ = torch.split(gm, 1, dim=1)                      # lrtb_to_points (heads/utils.py:20-23)
                    d = torch.cat([-l_, -t_, r_, b_], 1) / c["strides"][l] - torch.tensor([-1.0, -1.0, 1.0, 1.0]).view(1, 4, 1, 1)
                    ext = self._conv("offset_conv_extend.conv", lf, 0, rows=14, f32_out=True)
                    oc = ol_ = torch.cat([d[:, 0:2], ext, d[:, 2:4]], 1)
                cfa = self._r(torch.relu(deform_conv2d(cf, oc, self._r(self.p["cls_conv.weight"]), None, 1, 1, 1, sample_hook=hook)))
                lfa = self._r(torch.relu(deform_conv2d(lf, ol_, self._r(self.p["loc_refine_conv.weight"]), None, 1, 1, 1, sample_hook=hook)))
            kc = K + (0 if c["ctr_on_loc"] else 1)
            cp = self._conv("cls_pred.conv", cfa, 0, rows=kc, f32_out=True)
            bp = self._conv("box_pred.conv", lfa, 0, rows=5 if c["ctr_on_loc"] else 4, f32_out=True)
            ctr = bp[:, 4] if c["ctr_on_loc"] else cp[:, K]
            ref = self._decode(bp[:, :4], self.p["scales_refine"][l], c["strides"][l])
            if c["res"]:
                ref = ref + init.detach()
            C.append(cp[:, :K].permute(0, 2, 3, 1).reshape(N, -1, K)); T.append(ctr.reshape(N, -1))
            I.append(init.permute(0, 2, 3, 1).reshape(N, -1, 4)); R.append(ref.permute(0, 2, 3, 1).reshape(N, -1, 4))
        return torch.cat(C, 1), torch.cat(T, 1), torch.cat(I, 1), torch.cat(R, 1), [tuple(f.shape[2:]) for f in feats]

    def losses(self, feats, gt_boxes, gt_classes, topk=False):
        c = self.c
        cls, ctr, init, ref, hw = self.forward(feats)
        K = c["K"]
        mask = None
        if topk:
            labels, reg_t, mask = topk_locations(hw, c["strides"], gt_boxes, gt_classes, c["radius"], K)
            mask = mask.reshape(-1)
        else:
            labels, reg_t = ot.targets_for_batch(hw, c["strides"], gt_boxes, gt_classes, c["radius"], K)
        return losses(labels.reshape(-1), reg_t.reshape(-1, 4), cls.reshape(-1, K), ctr.reshape(-1), init.reshape(-1, 4), ref.reshape(-1, 4), K,
                      c["alpha"], c["gamma"], c["iou_type"], c["slender"], c["w"], topk_mask=mask)
