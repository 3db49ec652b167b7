"""Oracle FCOS detector (CPU, fp32, NCHW, plain torch): ResNet-FPN -> FCOSHead -> targets -> losses -> SGD.

Restates the training step of slender_det/modeling/meta_arch/fcos/fcosv2.py:63-148 on top of the detectron2
ResNet/FPN semantics of SURVEY.md Appendix C.9/C.10 (third-party source absent: "parity unpinned" for those parts;
FCOSHead / targets / losses are pinned by tests/golden).  Also the ``cpu_baseline`` of bench.py.

``emulate_bf16=True`` rounds weights and every stored activation to bf16 at the same points where the HIP path stores
bf16, so that end-to-end comparisons isolate kernel errors from the precision the product path computes in.  A SET of storage points
instead of ``True`` rounds only those (tests/test_oracle_storage_ablation.py): ``"w"`` the compute copies of the weights, ``"act_bb"`` /
``"act_head"`` the stored activations of backbone + FPN / of the head towers, ``"grad"`` the stored activation gradients, ``"input"`` the
normalised image.
"""
import torch
import torch.nn.functional as F

from . import fcos_targets as ot
from . import losses as ol
from .nn import relu as _band_relu      # torch.relu unless a ReluBand context is active (oracle/nn.py)
from .nn import relu_at as _relu_at       # ... or teacher-forced decisions for this position (ForcedMasks)


def _rb(t, on):
    return t.to(torch.bfloat16).to(torch.float32) if on else t


class _RoundSTE(torch.autograd.Function):
    """bf16 rounding with a rounded straight-through gradient (activation gradients are stored as bf16 too)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    """bf16 rounding of the forward value only (a stored activation, or the bf16 compute copy of an fp32 master weight)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bf16: fp32 prediction rows whose GRADIENT rows the product path stores in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class OracleFCOS:
    """Functional model over a dict of fp32 tensors in torch layouts (conv weights KCRS)."""

    def __init__(self, params, buffers, cfg_like, emulate_bf16=False):
        self.p = params          # name -> tensor (requires_grad for trainable)
        self.b = buffers         # frozen BN scale/shift per conv name
        self.c = cfg_like        # dict of hyper-parameters
        self.emu = emulate_bf16

    # ------------------------------------------------------------------ construction from the HIP model
    @classmethod
    def from_hip_model(cls, model, emulate_bf16=False):
        """Copy weights out of a slenderobjdet_amd FCOSV2 (any device) into torch-layout CPU tensors."""
        params, buffers, dcn, groups = cls._collect(model)
        params["head.scales"] = model.head.scales.detach().float().cpu().clone().requires_grad_(True)
        cfg_like = cls._backbone_cfg(model, params, dcn, groups)
        cfg_like.update(
            num_classes=model.num_classes, strides=list(model.fpn_strides),
            radius=model.center_sampling_radius, alpha=model.focal_loss_alpha, gamma=model.focal_loss_gamma,
            iou_type=model.iou_loss_type, norm_reg=model.head.norm_reg_targets, ctr_on_reg=model.head.centerness_on_reg,
            kc=model.head.kc, num_convs=len(model.head.cls_tower))
        return cls(params, buffers, cfg_like, emulate_bf16)

    @staticmethod
    def _collect(model):
        from slenderobjdet_amd.layers.deform_conv import DeformConv
        from slenderobjdet_amd.layers.nn import HipConv2d, HipGroupNorm

        params, buffers, dcn, groups = {}, {}, {}, {}
        for name, m in model.named_modules():
            if isinstance(m, DeformConv):      # DeformConv / ModulatedDeformConv (detectron2): KRSC -> KCRS, optional FrozenBN
                params[name + ".weight"] = m.weight.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(m.weight.requires_grad)
                if m.bias is not None:
                    params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(m.bias.requires_grad)
                if m.frozen_bn:
                    scale = m.bn_weight.float().cpu() * torch.rsqrt(m.bn_running_var.float().cpu() + 1e-5)
                    buffers[name + ".scale"], buffers[name + ".shift"] = scale, m.bn_bias.float().cpu() - m.bn_running_mean.float().cpu() * scale
                dcn[name] = dict(modulated=m.modulated, dg=m.deformable_groups, stride=m.stride, pad=m.padding, dil=m.dilation, groups=getattr(m, "groups", 1))
            elif isinstance(m, HipConv2d):
                if getattr(m, "groups", 1) > 1:
                    groups[name] = m.groups
                w = m.weight.detach().float().cpu().permute(0, 3, 1, 2).contiguous()      # (NO copy for a CPU model's 1x1 weights: oracles built from one CPU model share them - copy the model first)
                params[name + ".weight"] = w.requires_grad_(m.weight.requires_grad)
                if m.bias is not None:
                    params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(m.bias.requires_grad)
                if m.frozen_bn:
                    scale = m.bn_weight.float().cpu() * torch.rsqrt(m.bn_running_var.float().cpu() + 1e-5)
                    shift = m.bn_bias.float().cpu() - m.bn_running_mean.float().cpu() * scale
                    buffers[name + ".scale"], buffers[name + ".shift"] = scale, shift
            elif isinstance(m, HipGroupNorm):
                params[name + ".weight"] = m.weight.detach().float().cpu().clone().requires_grad_(True)
                params[name + ".bias"] = m.bias.detach().float().cpu().clone().requires_grad_(True)
        return params, buffers, dcn, groups

    @staticmethod
    def _backbone_cfg(model, params, dcn, groups):
        res_names = [name for _, name in model.backbone.bottom_up.stages_and_names]
        blocks = {n: len(getattr(model.backbone.bottom_up, n)) for n in res_names}
        bottleneck = any(k.endswith("conv3.weight") for k in params)
        top = getattr(model.backbone, "top_block", None)
        return dict(
            blocks=blocks, bottleneck=bottleneck,
            mean=[float(v) for v in model.pixel_mean.flatten()], std=[float(v) for v in model.pixel_std.flatten()],
            size_div=model.backbone.size_divisibility,
            stride_in_1x1={n: [blk.conv1.stride for blk in getattr(model.backbone.bottom_up, n)] for n in res_names},
            block_stride={n: [blk.stride for blk in getattr(model.backbone.bottom_up, n)] for n in res_names},
            dcn=dcn, groups=groups, p6_from=getattr(top, "in_feature", "p5"),
        )

    def double(self):
        """The same model in float64 (an arbiter between two fp32 implementations: tests/test_gpu_f32_mode.py).  In place; returns self."""
        self.p = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in self.p.items()}
        self.b = {k: v.double() for k, v in self.b.items()}
        return self

    # ------------------------------------------------------------------ layers
    def _act(self, x, where="bb"):
        e = self.emu
        if e is True:
            return _RoundSTE.apply(x)
        if not e:
            return x
        if ("act_" + where) in e:
            x = _RoundFwd.apply(x)
        return _RoundGrad.apply(x) if "grad" in e else x

    def _wt(self, w):
        e = self.emu
        if e is True:
            return _RoundSTE.apply(w)
        return _RoundFwd.apply(w) if (e and "w" in e) else w

    @staticmethod
    def _where(name):
        return "head" if name.startswith("head.") else "bb"

    def _dcn(self, name, x, om, relu=False):
        """DeformConv / ModulatedDeformConv ``name`` on x with the offset conv's output ``om`` (first 18*G channels offsets, next 9*G mask
        logits; df_conv.py:67-78, detectron2 DeformBottleneckBlock), optional folded FrozenBN, ReLU."""
        from . import deform_conv as odc

        d = self.c["dcn"][name]
        w, bias = self.p[name + ".weight"], self.p.get(name + ".bias")
        if name + ".scale" in self.b:
            w = w * self.b[name + ".scale"].view(-1, 1, 1, 1)
            bias = self.b[name + ".shift"] + (bias * self.b[name + ".scale"] if bias is not None else 0)
        w = self._wt(w)
        g = d["dg"]
        mask = om[:, 18 * g:27 * g].sigmoid() if d["modulated"] else None
        wh = self._where(name)
        y = odc.deform_conv2d(x, om[:, :18 * g], w, bias, d["stride"], d["pad"], d["dil"], mask, g,
                              sample_hook=(lambda t: self._act(t, wh)) if self.emu else None, groups=d.get("groups", 1))
        if relu:
            y = _relu_at(y, name)
        return self._act(y, wh)

    def _conv(self, name, x, stride=1, pad=0, relu=False, res=None, out_f32=False):
        w = self.p[name + ".weight"]
        bias = self.p.get(name + ".bias")
        if name + ".scale" in self.b:      # FrozenBN folded the way the product path folds it
            w = w * self.b[name + ".scale"].view(-1, 1, 1, 1)
            bias = self.b[name + ".shift"] + (bias * self.b[name + ".scale"] if bias is not None else 0)
        w = self._wt(w)
        y = F.conv2d(x, w, bias, stride=stride, padding=pad, groups=self.c.get("groups", {}).get(name, 1))      # ResNeXt 3x3: grouped
        if res is not None:
            y = y + res
        if relu:
            y = _relu_at(y, name)
        return y if out_f32 else self._act(y, self._where(name))      # offset / prediction convs keep fp32 rows on the product path

    def _bottom_up(self, x):
        c = self.c
        x = self._conv("backbone.bottom_up.stem.conv1", x, 2, 3, relu=True)
        x = F.max_pool2d(x, 3, 2, 1)
        outs = {}
        for stage, nblk in c["blocks"].items():
            for b in range(nblk):
                pre = f"backbone.bottom_up.{stage}.{b}"
                bs = c["block_stride"][stage][b]
                sc = self._conv(pre + ".shortcut", x, bs, 0) if (pre + ".shortcut.weight") in self.p else x
                if c["bottleneck"]:
                    s1 = c["stride_in_1x1"][stage][b]
                    y = self._conv(pre + ".conv1", x, s1, 0, relu=True)
                    if (pre + ".conv2_offset.weight") in self.p:       # DeformBottleneckBlock (MODEL.RESNETS.DEFORM_ON_PER_STAGE)
                        om = self._conv(pre + ".conv2_offset", y, bs // s1, 1, out_f32=True)
                        y = self._dcn(pre + ".conv2", y, om, relu=True)
                    else:
                        y = self._conv(pre + ".conv2", y, bs // s1, 1, relu=True)
                    x = self._conv(pre + ".conv3", y, 1, 0, relu=True, res=sc)
                else:
                    y = self._conv(pre + ".conv1", x, bs, 1, relu=True)
                    x = self._conv(pre + ".conv2", y, 1, 1, relu=True, res=sc)
            outs[stage] = x
        return outs

    def _fpn(self, feats):
        names = ["res5", "res4", "res3"]
        stage_id = {"res3": 3, "res4": 4, "res5": 5}
        prev, outs = None, {}
        for n in names:
            s = stage_id[n]
            up = F.interpolate(prev, scale_factor=2, mode="nearest") if prev is not None else None
            prev = self._conv(f"backbone.fpn_lateral{s}", feats[n], 1, 0, res=up)
            outs[f"p{s}"] = self._conv(f"backbone.fpn_output{s}", prev, 1, 1)
        # LastLevelP6P7 reads P5 (fpn.py:94-115, the FCOS builder) or res5 (detectron2's RetinaNet builder)
        p6 = self._conv("backbone.top_block.p6", feats["res5"] if self.c.get("p6_from", "p5") == "res5" else outs["p5"], 2, 1)
        p7 = self._conv("backbone.top_block.p7", self._act(_relu_at(p6, "backbone.top_block.p7:in")), 2, 1)
        outs["p6"], outs["p7"] = p6, p7
        return [outs[k] for k in ("p3", "p4", "p5", "p6", "p7")]

    def _tower(self, prefix, x):
        for i in range(self.c["num_convs"]):
            if f"{prefix}.{i}.conv.offset.weight" in self.p:       # DFConv2d as the last tower conv (USE_DCN_IN_TOWER, fcosv2.py:300-336)
                om = self._conv(f"{prefix}.{i}.conv.offset", x, 1, 1, out_f32=True)
                y = self._dcn(f"{prefix}.{i}.conv.conv", x, om)
            else:
                y = self._conv(f"{prefix}.{i}.conv", x, 1, 1)
            y = F.group_norm(y, 32, self.p[f"{prefix}.{i}.gn.weight"], self.p[f"{prefix}.{i}.gn.bias"], 1e-5)
            x = self._act(_relu_at(y, f"{prefix}.{i}.gn"), "head")
        return x

    def _head(self, feats):
        """Returns flattened (N*L, K) logits, (N*L, 4) box predictions, (N*L,) centerness logits (fcosv2.py:358-380 +
        permute_and_concat)."""
        c = self.c
        K, kc = c["num_classes"], c["kc"]
        cls_all, box_all, ctr_all = [], [], []
        for lvl, f in enumerate(feats):
            ct, bt = self._tower("head.cls_tower", f), self._tower("head.bbox_tower", f)
            wc, bc = self.p["head.cls_pred.weight"], self.p["head.cls_pred.bias"]
            wb, bb = self.p["head.box_pred.weight"], self.p["head.box_pred.bias"]
            wc, wb = self._wt(wc), self._wt(wb)
            co = F.conv2d(ct, wc[:kc], bc[:kc], padding=1)
            bo = F.conv2d(bt, wb[:5 if c["ctr_on_reg"] else 4], bb[:5 if c["ctr_on_reg"] else 4], padding=1)
            logits = co[:, :K]
            ctr = bo[:, 4:5] if c["ctr_on_reg"] else co[:, K:K + 1]
            z = bo[:, :4] * self.p["head.scales"][lvl]
            box = _band_relu(z) * c["strides"][lvl] if c["norm_reg"] else torch.exp(z)
            N = f.shape[0]
            cls_all.append(logits.permute(0, 2, 3, 1).reshape(N, -1, K))
            box_all.append(box.permute(0, 2, 3, 1).reshape(N, -1, 4))
            ctr_all.append(ctr.permute(0, 2, 3, 1).reshape(N, -1))
        return torch.cat(cls_all, 1).reshape(-1, K), torch.cat(box_all, 1).reshape(-1, 4), torch.cat(ctr_all, 1).reshape(-1)

    # ------------------------------------------------------------------ step
    def preprocess(self, batched_inputs):
        c = self.c
        imgs = [(x["image"].float().cpu() - torch.tensor(c["mean"]).view(-1, 1, 1)) / torch.tensor(c["std"]).view(-1, 1, 1) for x in batched_inputs]
        mh, mw = max(i.shape[1] for i in imgs), max(i.shape[2] for i in imgs)
        d = c["size_div"]
        mh, mw = (mh + d - 1) // d * d, (mw + d - 1) // d * d
        batch = torch.zeros(len(imgs), 3, mh, mw)
        for i, im in enumerate(imgs):
            batch[i, :, : im.shape[1], : im.shape[2]] = im
        return _rb(batch, self.emu is True or (bool(self.emu) and "input" in self.emu)).to(next(iter(self.p.values())).dtype)

    def losses(self, batched_inputs, world=1):
        c = self.c
        x = self.preprocess(batched_inputs)
        feats = self._fpn(self._bottom_up(x))
        level_hw = [tuple(f.shape[2:]) for f in feats]
        boxes = [b["instances"].gt_boxes.tensor.float().cpu() for b in batched_inputs]
        classes = [b["instances"].gt_classes.cpu() for b in batched_inputs]
        labels, reg_t = ot.targets_for_batch(level_hw, c["strides"], boxes, classes, c["radius"], c["num_classes"])
        cls, box, ctr = self._head(feats)
        return ol.fcos_losses(labels.reshape(-1), reg_t.reshape(-1, 4), cls, box, ctr, c["num_classes"], c["alpha"], c["gamma"],
                              c["iou_type"], world)

    def trainable(self):
        return {k: v for k, v in self.p.items() if v.requires_grad}

    def sgd_step(self, grads, state, lr, momentum=0.9, wd=1e-4, wd_norm=0.0):
        """torch.optim.SGD with the reference's per-parameter weight decay (solver/build.py:36-104)."""
        with torch.no_grad():
            for k, p in self.trainable().items():
                g = grads[k]
                decay = wd_norm if ".gn." in k else wd
                d = g + decay * p
                if momentum:
                    buf = state.get(k)
                    buf = d.clone() if buf is None else momentum * buf + d
                    state[k] = buf
                    d = buf
                p -= lr * d


class OracleRetinaNet(OracleFCOS):
    """The RetinaNet training step (BASELINE configs[2]; retina_rotated.py:129-295 for axis-aligned boxes, head :390-474) over the same
    backbone restatement: conv+ReLU subnets without a norm, A*K logits / A*4 deltas per location, anchor labels by IoU matching, focal +
    smooth-L1 / GIoU loss over the EMA loss normaliser."""

    @classmethod
    def from_hip_model(cls, model, emulate_bf16=False):
        params, buffers, dcn, groups = cls._collect(model)
        cfg_like = cls._backbone_cfg(model, params, dcn, groups)
        cfg_like.update(
            num_classes=model.num_classes, strides=list(model.strides), alpha=model.focal_loss_alpha, gamma=model.focal_loss_gamma,
            beta=model.smooth_l1_loss_beta, box_reg=model.box_reg_loss_type, weights=tuple(model.bbox_reg_weights),
            thresholds=list(model.iou_thresholds), labels=list(model.iou_labels), sizes=model.anchor_sizes, ratios=model.anchor_ratios,
            offset=model.anchor_offset, num_anchors=model.head.num_anchors, num_convs=len(model.head.cls_subnet),
            normalizer=float(model.loss_normalizer), momentum=model.loss_normalizer_momentum)
        return cls(params, buffers, cfg_like, emulate_bf16)

    def _subnet(self, prefix, x):
        for i in range(self.c["num_convs"]):
            x = self._conv(f"{prefix}.{i}.conv", x, 1, 1, relu=True)
        return x

    def predictions(self, feats):
        """(N, R, K) logits and (N, R, 4) deltas in detectron2's (level, h, w, anchor) row order (permute_to_N_HWA_K, :150-153)."""
        c = self.c
        A, K = c["num_anchors"], c["num_classes"]
        wc, bc = self.p["head.cls_score.weight"][:A * K], self.p["head.cls_score.bias"][:A * K]
        wb, bb = self.p["head.bbox_pred.weight"][:A * 4], self.p["head.bbox_pred.bias"][:A * 4]       # the product path pads 36 -> 40 rows
        if self.emu:
            wc, wb = _RoundSTE.apply(wc), _RoundSTE.apply(wb)
        logits, deltas = [], []
        for f in feats:
            N = f.shape[0]
            co = F.conv2d(self._subnet("head.cls_subnet", f), wc, bc, padding=1)
            bo = F.conv2d(self._subnet("head.bbox_subnet", f), wb, bb, padding=1)
            if self.emu:        # the focal / box gradient rows are bf16 on the product path (retinanet.py _RetinaLossFn.backward)
                co, bo = _RoundGrad.apply(co), _RoundGrad.apply(bo)
            logits.append(co.permute(0, 2, 3, 1).reshape(N, -1, K))
            deltas.append(bo.permute(0, 2, 3, 1).reshape(N, -1, 4))
        return torch.cat(logits, 1), torch.cat(deltas, 1)

    def losses(self, batched_inputs, world=1):
        from . import rcnn as orc
        from . import retinanet as orn

        c = self.c
        feats = self._fpn(self._bottom_up(self.preprocess(batched_inputs)))
        level_hw = [tuple(f.shape[2:]) for f in feats]
        anchors = torch.cat(orc.anchors(level_hw, c["strides"], c["sizes"], c["ratios"], None, c["offset"]))
        boxes = [b["instances"].gt_boxes.tensor.float().cpu() for b in batched_inputs]
        classes = [b["instances"].gt_classes.cpu() for b in batched_inputs]
        gt_labels, matched = orn.label_anchors(anchors, boxes, classes, c["thresholds"], c["labels"], c["num_classes"])
        logits, deltas = self.predictions(feats)
        dt = logits.dtype
        out, norm = orn.losses(anchors.to(dt), logits, deltas, gt_labels, matched.to(dt), c["num_classes"], c["alpha"], c["gamma"], c["beta"],
                               c["weights"], c["normalizer"], c["momentum"], box_reg_loss_type=c["box_reg"])
        self.new_normalizer = norm
        return out


def random_batch_cpu(n, h, w, seed):
    from slenderobjdet_amd.data import synthetic_batch

    return synthetic_batch(n, h, w, seed, device="cpu")
