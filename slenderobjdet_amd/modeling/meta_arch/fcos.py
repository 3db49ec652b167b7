"""FCOS / FCOSV2 meta-architectures on the HIP kernels.

Mirror of slender_det/modeling/meta_arch/fcos/fcosv2.py:22-381 (``FCOSV2``, what configs/fcos/fcos_R_50_FPN_1x.yaml:3
selects) and fcos.py:174-582 (``FCOS``): same constructor contract (``cls(cfg)``), same ``forward(batched_inputs)``
contract and loss-dict keys (``cls_loss``, ``reg_loss``, ``centerness_loss``).

MI355X-first differences in HOW (not WHAT):
  * activations are NHWC bf16; the five per-level prediction convs write straight into concatenated
    (N, sum Hi*Wi, K) fp32 buffers, so ``permute_and_concat`` (fcos/utils.py:32-52) disappears;
  * target assignment (fcos/utils.py:160-212) is one kernel for the whole batch; the positives are never gathered
    (``nonzero`` at fcosv2.py:112) — the loss kernels run over all locations with the label as mask;
  * the two scalar all-reduces (num_pos, sum of centerness targets; fcosv2.py:116,132) depend only on the targets,
    so they are issued as ONE 2-element all-reduce before the backbone runs and stay on the device: no ``.item()``;
  * ``Scale`` and ``exp`` of FCOSHead.forward (fcosv2.py:372-378) are fused into the regression-loss kernel.
"""
import math
import os
from typing import List

import torch
import torch.distributed as dist
from torch import nn
from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers.deform_conv import DFConv2d
from ...layers.nn import ConvGnRelu, HipConv2d, HipGroupNorm, _arena_of, group_norm_relu
from ...structures import Boxes, ImageList, Instances
from ...utils import comm
from ..backbone import build_backbone
from ..shape_spec import ShapeSpec
from .build import META_ARCH_REGISTRY

INF = 100000000   # fcos/utils.py:7
SIZES_OF_INTEREST = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]   # fcosv2.py:152-158


def _ceil8(v):
    return (v + 7) // 8 * 8


# SOD_TOWER_STREAMS=1: box tower on a second stream (experiment; measured neutral on the FCOS R50 step, 463.1 vs 463.5 img/s, so off)
TOWER_STREAMS = os.environ.get("SOD_TOWER_STREAMS", "0") == "1"
_tower_streams = {}


class DcnGnRelu(nn.Module):
    """[DFConv2d (no bias) -> GroupNorm(32) -> ReLU]: the last tower unit under MODEL.FCOS.USE_DCN_IN_TOWER."""

    def __init__(self, channels, v2):
        super().__init__()
        self.conv = DFConv2d(channels, channels, with_modulated_dcn=v2, kernel_size=3, stride=1, padding=1, bias=False)
        self.gn = HipGroupNorm(32, channels)

    def forward(self, xs):
        return [group_norm_relu(self.conv(x), self.gn, True) for x in xs]


class FCOSHead(nn.Module):
    """fcosv2.py:277-381. ``cls_logits`` (+ ``centerness`` when not CENTERNESS_ON_REG) live in one fused conv
    ``cls_pred`` whose output channels are padded to a multiple of 8; ``bbox_pred`` (+ ``centerness`` when
    CENTERNESS_ON_REG) live in ``box_pred`` (8 output channels: l, t, r, b, ctr, 0, 0, 0)."""

    def __init__(self, cfg, input_shape: List[ShapeSpec]):
        super().__init__()
        in_channels = input_shape[0].channels
        self.num_classes = cfg.MODEL.FCOS.NUM_CLASSES
        self.fpn_strides = list(cfg.MODEL.FCOS.FPN_STRIDES)
        self.norm_reg_targets = cfg.MODEL.FCOS.NORM_REG_TARGETS
        self.centerness_on_reg = cfg.MODEL.FCOS.CENTERNESS_ON_REG
        n = cfg.MODEL.FCOS.NUM_CONVS
        self.use_dcn_in_tower = cfg.MODEL.FCOS.USE_DCN_IN_TOWER
        self.use_dcn_v2 = cfg.MODEL.FCOS.USE_DCN_V2

        def unit(i):   # fcosv2.py:296-336: the LAST tower conv becomes DFConv2d (no bias) when USE_DCN_IN_TOWER
            if self.use_dcn_in_tower and i == n - 1:
                return DcnGnRelu(in_channels, self.use_dcn_v2)
            return ConvGnRelu(in_channels)

        self.cls_tower = nn.ModuleList([unit(i) for i in range(n)])
        self.bbox_tower = nn.ModuleList([unit(i) for i in range(n)])
        self.kc = self.num_classes + (0 if self.centerness_on_reg else 1)
        self.kc_pad = _ceil8(self.kc)
        self.cls_pred = HipConv2d(in_channels, self.kc_pad, 3, 1, 1, bias=True)
        self.box_pred = HipConv2d(in_channels, 8, 3, 1, 1, bias=True)
        for unit in list(self.cls_tower) + list(self.bbox_tower):
            if isinstance(unit, ConvGnRelu):   # the reference's init loop only touches nn.Conv2d, not DFConv2d (fcos.py:549-550)
                unit.conv.init_normal(0.01, 0.0)
        bias_value = -math.log((1 - cfg.MODEL.FCOS.PRIOR_PROB) / cfg.MODEL.FCOS.PRIOR_PROB)
        with torch.no_grad():
            self.cls_pred.init_normal(0.01, 0.0)
            self.cls_pred.bias[: self.num_classes].fill_(bias_value)
            self.cls_pred.weight[self.kc:].zero_()
            self.box_pred.init_normal(0.01, 0.0)
            nb = 5 if self.centerness_on_reg else 4
            self.box_pred.weight[nb:].zero_()
        self.scales = nn.Parameter(torch.ones(len(self.fpn_strides)))   # five Scale(init_value=1.0) modules

    def num_logical_params(self):
        """Parameter count of the reference head (padding rows excluded): 4 920 666 for the default config."""
        c = self.cls_pred.in_channels
        pad_rows = (self.kc_pad - self.kc) + (8 - (5 if self.centerness_on_reg else 4))
        return sum(p.numel() for p in self.parameters()) - pad_rows * (9 * c + 1)

    def run_towers(self, feats):
        """Every tower unit runs over all FPN levels in one multi-level launch (the levels share the weights)."""
        cls_t, box_t = list(feats), list(feats)
        if not (TOWER_STREAMS and feats[0].is_cuda):
            for unit in self.cls_tower:
                cls_t = unit(cls_t)
            for unit in self.bbox_tower:
                box_t = unit(box_t)
            return cls_t, box_t
        # The two towers are independent chains of (MFMA-bound conv, HBM-bound GroupNorm) launches: the box tower runs on a second
        # stream, enqueued unit by unit alongside the classification tower, so that one tower's GroupNorm passes overlap the other's
        # convolution.  autograd replays each node's backward on the stream of its forward, which gives the same overlap in backward.
        dev = feats[0].device
        main = torch.cuda.current_stream(dev)
        s2 = _tower_streams.get(dev.index)
        if s2 is None:
            s2 = _tower_streams[dev.index] = torch.cuda.Stream(device=dev)
        s2.wait_stream(main)
        for f in feats:
            f.record_stream(s2)
        for cu, bu in zip(self.cls_tower, self.bbox_tower):
            with torch.cuda.stream(s2):
                box_t = bu(box_t)
            cls_t = cu(cls_t)
        main.wait_stream(s2)
        for t in box_t:
            t.record_stream(main)
        return cls_t, box_t

    def predict(self, cls_t, box_t):
        """Prediction convs of all levels into concatenated fp32 buffers (N, L, kc_pad) and (N, L, 8)."""
        self.cls_pred.prepare()
        self.box_pred.prepare()
        N = cls_t[0].shape[0]
        hw = [(t.shape[1], t.shape[2]) for t in cls_t]
        L = sum(h * w for h, w in hw)
        dev = cls_t[0].device
        cls_buf = torch.empty((N, L, self.kc_pad), dtype=torch.float32, device=dev)
        box_buf = torch.empty((N, L, 8), dtype=torch.float32, device=dev)
        offs, off = [], 0
        for h, w in hw:
            offs.append(off)
            off += h * w
        HF.conv2d_fwd_ml(list(cls_t), self.cls_pred.w_bf16, self.cls_pred.bias_eff, 1, 1, 1, out_f32=True,
                         outs=[cls_buf.view(-1)[o * self.kc_pad:] for o in offs], y_img_stride=L * self.kc_pad, k_real=self.kc)
        HF.conv2d_fwd_ml(list(box_t), self.box_pred.w_bf16, self.box_pred.bias_eff, 1, 1, 1, out_f32=True,
                         outs=[box_buf.view(-1)[o * 8:] for o in offs], y_img_stride=L * 8, k_real=5 if self.centerness_on_reg else 4)
        return cls_buf, box_buf, hw


class _FcosHeadLossFn(torch.autograd.Function):
    """Prediction convs + FCOSV2.losses (fcosv2.py:104-148) as one autograd node over the ten tower outputs."""

    @staticmethod
    def forward(ctx, model, scales, labels, reg_t, ctr_t, stats, inv_world, *towers):
        head = model.head
        nl = len(towers) // 2
        cls_t, box_t = list(towers[:nl]), list(towers[nl:])
        cls_buf, box_buf, hw = head.predict(cls_t, box_t)
        N = cls_buf.shape[0]
        K = head.num_classes
        focal_sum, _ = HF.focal_loss_fwd(cls_buf, labels, None, model.focal_loss_alpha, model.focal_loss_gamma, K=K)
        if head.centerness_on_reg:
            ctr_ptr, ld_ctr = box_buf.view(-1)[4:], 8
        else:
            ctr_ptr, ld_ctr = cls_buf.view(-1)[K:], head.kc_pad
        sums = HF.fcos_regctr_loss_fwd(box_buf, 8, ctr_ptr, ld_ctr, labels, reg_t, ctr_t, head.scales.detach(), N, hw,
                                       head.fpn_strides, K, model.iou_loss_type, head.norm_reg_targets)
        out3 = HF.fcos_finalize_losses(focal_sum, sums, stats, inv_world)
        ctx.model, ctx.hw, ctx.inv_world = model, hw, inv_world
        ctx.save_for_backward(cls_buf, box_buf, labels, reg_t, ctr_t, stats, *towers)
        arena = _arena_of(head)
        if arena is not None:
            for p in (head.cls_pred.weight, head.cls_pred.bias, head.box_pred.weight, head.box_pred.bias, head.scales):
                arena.note_use(p)
        return out3

    @staticmethod
    @once_differentiable
    def backward(ctx, g3):
        model, hw, inv_world = ctx.model, ctx.hw, ctx.inv_world
        head = model.head
        cls_buf, box_buf, labels, reg_t, ctr_t, stats = ctx.saved_tensors[:6]
        towers = ctx.saved_tensors[6:]
        nl = len(towers) // 2
        cls_t, box_t = towers[:nl], towers[nl:]
        g3 = g3.contiguous().float()
        N, L, K, kcp = cls_buf.shape[0], cls_buf.shape[1], head.num_classes, head.kc_pad
        arena = _arena_of(head)
        dev = cls_buf.device
        # d(cls logits): focal gradient * g[0] / max(num_pos/world, 1), bf16 rows padded to kc_pad
        dcls = torch.empty((N, L, kcp), dtype=torch.bfloat16, device=dev)
        HF.focal_loss_bwd(cls_buf, labels, None, model.focal_loss_alpha, model.focal_loss_gamma, K=K, scale_num=g3[0:1],
                          scale_den=stats[0:1], den_mul=inv_world, den_min=1.0, ld_out=kcp, out_bf16=True, out=dcls)
        dbox = torch.empty((N, L, 8), dtype=torch.bfloat16, device=dev)
        if head.centerness_on_reg:
            ctr_ptr, ld_ctr = box_buf.view(-1)[4:], 8
            dctr, ld_dctr, dctr_col, ctr_col = dbox, 8, 4, 4
        else:
            ctr_ptr, ld_ctr = cls_buf.view(-1)[K:], kcp
            dctr, ld_dctr, dctr_col, ctr_col = dcls, kcp, K, 4
        HF.fcos_regctr_loss_bwd(box_buf, 8, ctr_ptr, ld_ctr, labels, reg_t, ctr_t, head.scales.detach(), N, hw, head.fpn_strides, K,
                                model.iou_loss_type, head.norm_reg_targets, g3[1:2], g3[2:3], stats, inv_world,
                                dbox, 8, ctr_col, dctr, ld_dctr, dctr_col, arena.grad_view(head.scales))
        arena.mark_ready(head.scales)
        offs, off = [], 0
        for h, w in hw:
            offs.append(off)
            off += h * w
        grads = []
        for pred, dbuf, kk, tower, kr in ((head.cls_pred, dcls, kcp, cls_t, head.kc), (head.box_pred, dbox, 8, box_t, 5 if head.centerness_on_reg else 4)):
            dys = [dbuf.view(-1)[o * kk:] for o in offs]
            HF.conv2d_wgrad_ml(dys, list(tower), arena.grad_view(pred.weight), 3, 3, 1, 1, 1, dy_img_stride=L * kk, K=kk, k_real=kr)
            arena.mark_ready(pred.weight)
            HF.bias_grad(dbuf, arena.grad_view(pred.bias), N, L, kk)
            arena.mark_ready(pred.bias)
            grads.append(HF.conv2d_dgrad_ml(dys, pred.wt_bf16, hw, 1, 1, 1, dy_img_stride=L * kk, N=N, k_real=kr))
        grads_cls, grads_box = grads
        return (None, None, None, None, None, None, None, *grads_cls, *grads_box)


@META_ARCH_REGISTRY.register()
class FCOSV2(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.in_features = cfg.MODEL.FCOS.IN_FEATURES
        self.fpn_strides = list(cfg.MODEL.FCOS.FPN_STRIDES)
        self.center_sampling_radius = cfg.MODEL.FCOS.CENTER_SAMPLING_RADIUS
        self.norm_reg_targets = cfg.MODEL.FCOS.NORM_REG_TARGETS
        self.focal_loss_alpha = cfg.MODEL.FCOS.FOCAL_LOSS_ALPHA
        self.focal_loss_gamma = cfg.MODEL.FCOS.FOCAL_LOSS_GAMMA
        self.iou_loss_type = cfg.MODEL.FCOS.IOU_LOSS_TYPE
        self.score_thresh = 0.3
        self.pre_nms_thresh = cfg.MODEL.FCOS.INFERENCE_TH
        self.pre_nms_top_n = cfg.MODEL.FCOS.PRE_NMS_TOP_N
        self.nms_thresh = cfg.MODEL.FCOS.NMS_TH
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.min_size = 0
        self.num_classes = cfg.MODEL.FCOS.NUM_CLASSES

        self.backbone = build_backbone(cfg)
        backbone_shape = self.backbone.output_shape()
        feature_shapes = [backbone_shape[f] for f in self.in_features]
        self.head = FCOSHead(cfg, feature_shapes)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------ forward
    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        if "instances" in batched_inputs[0]:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        elif "targets" in batched_inputs[0]:
            gt_instances = [x["targets"].to(self.device) for x in batched_inputs]
        else:
            gt_instances = None

        N, Hp, Wp = images.tensor.shape[:3]
        level_hw = [((Hp + s - 1) // s, (Wp + s - 1) // s) for s in self.fpn_strides]
        if self.training:
            # targets first: they depend only on the ground truth, so the normaliser all-reduce overlaps the backbone
            labels, reg_t, ctr_t, stats = self.get_ground_truth(level_hw, gt_instances)
            world = comm.get_world_size()
            stats_work = None
            if world > 1:       # asynchronous: the compute stream only waits for it in front of the loss node, a whole forward pass later
                stats_work = dist.all_reduce(stats, op=dist.ReduceOp.SUM, async_op=True)

        features = self.backbone(images.tensor)
        features = [features[f] for f in self.in_features]
        assert [tuple(f.shape[1:3]) for f in features] == level_hw, "feature map sizes differ from the location grid"
        cls_t, box_t = self.head.run_towers(features)

        if self.training:
            if stats_work is not None:
                stats_work.wait()
            out3 = _FcosHeadLossFn.apply(self, self.head.scales, labels, reg_t, ctr_t, stats, 1.0 / float(world), *cls_t, *box_t)
            return dict(cls_loss=out3[0], reg_loss=out3[1], centerness_loss=out3[2])
        results = self.inference(level_hw, cls_t, box_t, images.image_sizes)
        return self.postprocess(results, batched_inputs, images.image_sizes)

    def losses_from_outputs(self, *a, **k):   # kept for symmetry with the reference's method name
        raise NotImplementedError("losses are fused with the prediction convs in _FcosHeadLossFn")

    @torch.no_grad()
    def get_ground_truth(self, level_hw, gt_instances):
        """fcosv2.py:150-172 + fcos/utils.py:160-212 for the whole batch in one kernel."""
        dev = self.device
        counts = [len(g) for g in gt_instances]
        offs = torch.tensor([0] + counts, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev, non_blocking=True)
        if sum(counts) > 0:
            boxes = torch.cat([g.gt_boxes.tensor for g in gt_instances]).float().contiguous()
            classes = torch.cat([g.gt_classes for g in gt_instances]).to(torch.int32).contiguous()
        else:
            boxes = torch.zeros((1, 4), dtype=torch.float32, device=dev)
            classes = torch.zeros((1,), dtype=torch.int32, device=dev)
        return HF.fcos_assign(boxes, classes, offs, len(gt_instances), level_hw, self.fpn_strides, SIZES_OF_INTEREST,
                              self.center_sampling_radius, self.num_classes)

    # ------------------------------------------------------------------ inference (fcosv2.py:174-266)
    @torch.no_grad()
    def inference(self, level_hw, cls_t, box_t, image_sizes):
        head = self.head
        cls_buf, box_buf, hw = head.predict(cls_t, box_t)
        K = self.num_classes
        N, L = cls_buf.shape[:2]
        dev = cls_buf.device
        lvl = torch.cat([torch.full((h * w,), i, dtype=torch.long) for i, (h, w) in enumerate(hw)]).to(dev)
        scale = head.scales.detach()[lvl][:, None]
        strides = torch.tensor(self.fpn_strides, dtype=torch.float32, device=dev)[lvl]
        z = box_buf[..., :4] * scale
        box_reg = torch.relu(z) * strides[:, None] if head.norm_reg_targets else torch.exp(z)
        ctr = box_buf[..., 4] if head.centerness_on_reg else cls_buf[..., K]
        locs = []
        for (h, w), s in zip(hw, self.fpn_strides):
            ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=dev)
            xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=dev)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            locs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + s // 2)
        results = []
        bounds = [0]
        for h, w in hw:
            bounds.append(bounds[-1] + h * w)
        for i, image_size in enumerate(image_sizes):
            boxes_all, scores_all, cls_all = [], [], []
            for l in range(len(hw)):
                sl = slice(bounds[l], bounds[l + 1])
                p = cls_buf[i, sl, :K].sigmoid()
                keep = p > self.pre_nms_thresh
                p = p * ctr[i, sl].sigmoid()[:, None]
                sc = p[keep]
                idx = keep.nonzero()
                loc_i, class_i = idx[:, 0], idx[:, 1]
                reg_i, locs_i = box_reg[i, sl][loc_i], locs[l][loc_i]
                n_keep = int(keep.sum())
                top_n = min(n_keep, self.pre_nms_top_n)
                if n_keep > top_n:
                    sc, ti = sc.topk(top_n, sorted=False)
                    class_i, reg_i, locs_i = class_i[ti], reg_i[ti], locs_i[ti]
                boxes_all.append(torch.stack([locs_i[:, 0] - reg_i[:, 0], locs_i[:, 1] - reg_i[:, 1],
                                              locs_i[:, 0] + reg_i[:, 2], locs_i[:, 1] + reg_i[:, 3]], dim=1))
                scores_all.append(torch.sqrt(sc))
                cls_all.append(class_i)
            boxes_all, scores_all, cls_all = torch.cat(boxes_all), torch.cat(scores_all), torch.cat(cls_all)
            from ...layers.nms import batched_nms

            keep = batched_nms(boxes_all, scores_all, cls_all, self.nms_thresh)[: self.max_detections_per_image]
            r = Instances(tuple(image_size))
            r.pred_boxes = Boxes(boxes_all[keep])
            r.scores = scores_all[keep]
            r.pred_classes = cls_all[keep]
            results.append(r)
        return results

    def postprocess(self, instances, batched_inputs, image_sizes):
        from ..postprocessing import detector_postprocess

        out = []
        for res, inp, size in zip(instances, batched_inputs, image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            out.append({"instances": detector_postprocess(res, h, w)})
        return out

    def preprocess_image(self, batched_inputs):
        """fcosv2.py:268-275: normalise, pad to size_divisibility, batch — one kernel per image straight into the
        NHWC(8) bf16 batch buffer (the H2D copy of the uint8 image is the only other traffic)."""
        imgs = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in imgs]
        Hp, Wp = ImageList.padded_size(sizes, self.backbone.size_divisibility)
        batch = torch.empty((len(imgs), Hp, Wp, 8), dtype=torch.bfloat16, device=self.device)
        imgs = [im if im.dtype == torch.uint8 else im.float() for im in imgs]
        HF.preprocess_batch(imgs, batch, self._mean, self._std)       # one launch for the batch
        return ImageList(batch, sizes)


@META_ARCH_REGISTRY.register()
class FCOS(FCOSV2):
    """slender_det/modeling/meta_arch/fcos/fcos.py:174-473 — same head, targets and losses as FCOSV2 (it differs only in
    how the reference transposes the targets, fcos.py:343-372, and in ``inference_single_image``'s threshold order)."""
