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


# The box tower runs on a second stream beside the classification tower (FCOSHead.run_towers).  Round 1 measured this neutral (463.1
# vs 463.5 img/s); with the round-2 kernels the convs are short enough that the GroupNorm passes of one tower hide behind the other's
# convs: 573.3-574.6 -> 583.1-584.1 img/s in a 4-round A/B, allocator pool 11.1 -> 14.9 GB and flat over 160 steps.
# SOD_TOWER_STREAMS=0 keeps both towers on one stream.
TOWER_STREAMS = os.environ.get("SOD_TOWER_STREAMS", "1") != "0"
# SOD_TOWER_FOLD (default 1): the second tower's first data-gradient launch adds the first one's in its epilogue (layers/nn.py SiblingFold)
# instead of autograd adding the two towers' input gradients itself (five elementwise launches).  Its effect is below the noise of 60-step
# runs (round 3: 622.2 / 623.3 vs 622.2 / 623.5 img/s; round 4: 631.0 / 631.1 vs 632.4 / 633.2 with the accumulating tower on the side
# stream, 625.8 / 624.8 vs 623.8 / 625.2 with it on the main stream, as built now); nine alternating pairs of 100-step runs on two boxes put
# it at +0.3 % (634.4 vs 632.7 together with a second weight-gradient stream, which alone is -0.3 %; 644.6 vs 642.6 alone, every pair in
# favour): on.
TOWER_FOLD = os.environ.get("SOD_TOWER_FOLD", "1") != "0"
_tower_streams = {}
_prefetch_streams = {}


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

    def run_towers(self, feats, fold_input_grads=False):
        """Every tower unit runs over all FPN levels in one multi-level launch (the levels share the weights)."""
        cls_t, box_t = list(feats), list(feats)
        from ...layers import nn as _nn
        # the first units of the two towers read the same tensors: the second one to run backward adds the first one's data gradient in
        # its epilogue (layers/nn.py SiblingFold).  Only under forward()'s fused loss node, which always feeds both towers.
        fold = None
        if (fold_input_grads and TOWER_FOLD and torch.is_grad_enabled() and not HF.is_f32() and feats[0].is_cuda
                and all(f.requires_grad for f in feats) and isinstance(self.cls_tower[0], ConvGnRelu) and isinstance(self.bbox_tower[0], ConvGnRelu)):
            fold = _nn.SiblingFold()

        def run(unit, xs, prev):     # the first units of the two towers share the SiblingFold
            if isinstance(unit, ConvGnRelu):
                return unit(xs, fold=fold if prev is None else None)
            return unit(xs)

        if not (TOWER_STREAMS and feats[0].is_cuda):
            prev = None
            for unit in self.cls_tower:
                cls_t, prev = run(unit, cls_t, prev), unit
            prev = None
            for unit in self.bbox_tower:
                box_t, prev = run(unit, box_t, prev), unit
            return cls_t, box_t
        # The two towers are independent chains of (MFMA-bound conv, HBM-bound GroupNorm) launches: the box tower runs on a second
        # stream, enqueued unit by unit alongside the classification tower, so that one tower's GroupNorm passes overlap the other's
        # convolution.  autograd replays each node's backward on the stream of its forward, which gives the same overlap in backward.
        dev = feats[0].device
        main = torch.cuda.current_stream(dev)
        s2 = _tower_streams.get(dev.index)
        if s2 is None:
            # the box tower's stream at the HIGH HIP priority (-1): in backward it is the stream the main stream ends up waiting for (it
            # shares the CUs with the weight-gradient stream from its first kernel on).  Five + four alternating pairs of 100-step runs:
            # 642.9 vs 640.5 img/s (+0.4 %), every pair in favour; with the main stream high as well: 635.5 vs 641.2.
            s2 = _tower_streams[dev.index] = HF.make_stream(dev, -1, "TOWER")
            HF.register_compute_stream(dev, s2)
        s2.wait_stream(main)
        for f in feats:
            f.record_stream(s2)
        pc = pb = None
        for i, (cu, bu) in enumerate(zip(self.cls_tower, self.bbox_tower)):
            if i == 0 and fold is not None:
                # autograd runs the later-created node first: the box tower (whose stream gets ahead of the main stream in
                # backward - the main stream also carries the loss node) parks its gradient, the classification tower on the main
                # stream adds it without waiting.  The other way round the main stream waits for a serialised second launch.
                cls_t, pc = run(cu, cls_t, pc), cu
            with torch.cuda.stream(s2):
                box_t, pb = run(bu, box_t, pb), bu
            if not (i == 0 and fold is not None):
                cls_t, pc = run(cu, cls_t, pc), cu
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
        # three OUTPUTS, not one vector the caller indexes: ``out3[i]`` outside the node costs a zeros + copy launch per loss in backward
        # (SelectBackward) plus two adds to merge them - eight tiny launches at the forward / backward junction (no measurable effect on the
        # step, 644.2 vs 644.2 img/s over four alternating pairs: the look-ahead stream fills that window; kept for the shorter graph)
        return out3[0], out3[1], out3[2]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_cls, g_reg, g_ctr):
        model, hw, inv_world = ctx.model, ctx.hw, ctx.inv_world
        head = model.head
        cls_buf, box_buf, labels, reg_t, ctr_t, stats = ctx.saved_tensors[:6]
        towers = ctx.saved_tensors[6:]
        nl = len(towers) // 2
        cls_t, box_t = towers[:nl], towers[nl:]
        g3 = [g.reshape(1).float() if g is not None else torch.zeros(1, dtype=torch.float32, device=cls_buf.device) for g in (g_cls, g_reg, g_ctr)]
        N, L, K, kcp = cls_buf.shape[0], cls_buf.shape[1], head.num_classes, head.kc_pad
        arena = _arena_of(head)
        dev = cls_buf.device
        # d(cls logits): focal gradient * g[0] / max(num_pos/world, 1), bf16 rows padded to kc_pad
        dcls = torch.empty((N, L, kcp), dtype=HF.ACT_DTYPE, device=dev)
        HF.focal_loss_bwd(cls_buf, labels, None, model.focal_loss_alpha, model.focal_loss_gamma, K=K, scale_num=g3[0],
                          scale_den=stats[0:1], den_mul=inv_world, den_min=1.0, ld_out=kcp, out_bf16=not HF.is_f32(), out=dcls)
        dbox = torch.empty((N, L, 8), dtype=HF.ACT_DTYPE, device=dev)
        if head.centerness_on_reg:
            ctr_ptr, ld_ctr = box_buf.view(-1)[4:], 8
            dctr, ld_dctr, dctr_col, ctr_col = dbox, 8, 4, 4
        else:
            ctr_ptr, ld_ctr = cls_buf.view(-1)[K:], kcp
            dctr, ld_dctr, dctr_col, ctr_col = dcls, kcp, K, 4
        HF.fcos_regctr_loss_bwd(box_buf, 8, ctr_ptr, ld_ctr, labels, reg_t, ctr_t, head.scales.detach(), N, hw, head.fpn_strides, K,
                                model.iou_loss_type, head.norm_reg_targets, g3[1], g3[2], stats, inv_world,
                                dbox, 8, ctr_col, dctr, ld_dctr, dctr_col, arena.grad_view(head.scales))
        arena.mark_ready(head.scales)
        offs, off = [], 0
        for h, w in hw:
            offs.append(off)
            off += h * w
        grads = []
        preds = ((head.cls_pred, dcls, kcp, cls_t, head.kc), (head.box_pred, dbox, 8, box_t, 5 if head.centerness_on_reg else 4))
        with HF.wgrad_batch():      # the four weight / bias gradient launches of the two prediction convs: one hand-over to the side stream
            for pred, dbuf, kk, tower, kr in preds:
                dys = [dbuf.view(-1)[o * kk:] for o in offs]
                HF.conv2d_wgrad_ml(dys, list(tower), arena.grad_view(pred.weight), 3, 3, 1, 1, 1, dy_img_stride=L * kk, K=kk, k_real=kr)
                arena.mark_ready(pred.weight)
                HF.bias_grad(dbuf, arena.grad_view(pred.bias), N, L, kk)
                arena.mark_ready(pred.bias)
        for pred, dbuf, kk, tower, kr in preds:
            dys = [dbuf.view(-1)[o * kk:] for o in offs]
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
    def prefetch(self, batched_inputs):
        """Software pipelining for training loops: run ``preprocess_image`` and the FROZEN bottom of the backbone (stem + the
        FREEZE_AT stages: no gradient, weights that never change) for the NEXT batch on a side stream.  Called between forward and
        backward of the current batch, the side stream starts when that forward has finished on the GPU, so the HBM-bound frozen
        convolutions run beside the MFMA-bound head backward.  ``forward`` picks the result up when it is handed the same list
        object; a batch that was not prefetched (or another list) takes the normal path.  Results are identical either way."""
        bottom = getattr(self.backbone, "bottom_up", self.backbone)
        if not (self.device.type == "cuda" and hasattr(bottom, "forward_frozen_prefix")) or bottom.frozen_prefix_len() < 0:
            return False
        dev = self.device
        main = torch.cuda.current_stream(dev)
        # (re)build the frozen prefix's folded weights on the MAIN stream first: if a checkpoint load changed a buffer, the refolded
        # copies must not be allocated and written on the side stream, where a forward that does not come through _take_prefetched
        # (an evaluation hook, another batch) would read them without any ordering
        if hasattr(bottom, "prepare_frozen_prefix"):
            bottom.prepare_frozen_prefix()
        side = _prefetch_streams.get(dev.index)
        if side is None:
            side = _prefetch_streams[dev.index] = HF.make_stream(dev, 0, "PREFETCH")
        side.wait_stream(main)
        self._prefetched = None
        with torch.cuda.stream(side):
            images = self.preprocess_image(batched_inputs)
            prefix = bottom.forward_frozen_prefix(images.tensor)
        self._prefetched = (batched_inputs, images, prefix, side)
        return True

    def _take_prefetched(self, batched_inputs):
        pre, self._prefetched = getattr(self, "_prefetched", None), None
        if pre is None or pre[0] is not batched_inputs:
            return None
        _, images, prefix, side = pre
        self.prefetch_hits = getattr(self, "prefetch_hits", 0) + 1
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(side)
        for t in [prefix.tensor] + list(prefix.outputs.values()):
            t.record_stream(main)           # allocated on the side stream's pool, consumed (and kept for backward) on this one
        return ImageList(prefix, images.image_sizes)

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
            if comm.collectives_active():       # asynchronous: the compute stream only waits for it in front of the loss node, a whole forward pass later
                stats_work = dist.all_reduce(stats, op=dist.ReduceOp.SUM, async_op=True)

        features = self.backbone(images.tensor)
        features = [features[f] for f in self.in_features]
        assert [tuple(f.shape[1:3]) for f in features] == level_hw, "feature map sizes differ from the location grid"
        cls_t, box_t = self.head.run_towers(features, fold_input_grads=self.training)

        if self.training:
            if stats_work is not None:
                stats_work.wait()
            l_cls, l_reg, l_ctr = _FcosHeadLossFn.apply(self, self.head.scales, labels, reg_t, ctr_t, stats, 1.0 / float(world), *cls_t, *box_t)
            return dict(cls_loss=l_cls, reg_loss=l_reg, centerness_loss=l_ctr)
        results = self.inference(level_hw, cls_t, box_t, images.image_sizes)
        return self.postprocess(results, batched_inputs, images.image_sizes)

    def losses(self, gt_classes, reg_targets, pred_class_logits, pred_box_reg, pred_center_score):
        """FCOSV2.losses with the reference's argument contract (fcosv2.py:104-148): per-level NCHW prediction lists, labels
        (N, L) / (N*L,), regression targets (N, L, 4).  The training step does NOT come through here - forward() fuses the three
        losses with the prediction convs (_FcosHeadLossFn) - but code written against the reference's method keeps working, on the
        same HIP loss kernels and with the normalisers kept on the device (the reference reads them back with .item())."""
        from ...layers.losses import bce_with_logits_fg_sum, iou_loss, sigmoid_focal_loss_jit
        from ...utils import comm

        K = self.num_classes

        def cat(ts, c):       # permute_and_concat (fcos/utils.py:32-52): NCHW -> (N * sum(HW), c)
            return torch.cat([t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, c) for t in ts], 1).reshape(-1, c).float()

        logits, reg, ctr = cat(pred_class_logits, K), cat(pred_box_reg, 4), cat(pred_center_score, 1).reshape(-1)
        labels = gt_classes.flatten().to(torch.int32)
        reg_t = reg_targets.reshape(-1, 4).float()
        fg = (labels >= 0) & (labels != K)
        world = float(comm.get_num_gpus())
        num_pos_avg = (comm.reduce_sum(fg.sum().float().reshape(1)) / world).clamp(min=1.0)
        cls_loss = sigmoid_focal_loss_jit(logits, labels, alpha=self.focal_loss_alpha, gamma=self.focal_loss_gamma, reduction="sum") / num_pos_avg[0]
        lr, tb = reg_t[:, [0, 2]], reg_t[:, [1, 3]]           # compute_centerness_targets (fcos/utils.py:295-300) on every row
        ctr_t = torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]).clamp(min=0) * (tb.min(-1)[0] / tb.max(-1)[0]).clamp(min=0))
        ctr_t = torch.where(fg, ctr_t, torch.zeros_like(ctr_t))
        if bool(fg.any()):
            sum_ctr_avg = comm.reduce_sum(ctr_t.sum().reshape(1)) / world
            reg_loss = iou_loss(reg[fg], reg_t[fg], ctr_t[fg], loss_type=self.iou_loss_type) / sum_ctr_avg[0]
            centerness_loss = bce_with_logits_fg_sum(ctr, ctr_t, labels, K) / num_pos_avg[0]
        else:                 # fcosv2.py:143-146: keep the graph and the collective alive
            reg_loss = reg[fg].sum()
            comm.reduce_sum(ctr.new_zeros(1))
            centerness_loss = ctr[fg].sum()
        return dict(cls_loss=cls_loss, reg_loss=reg_loss, centerness_loss=centerness_loss)

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
    def decode_candidates(self, cls_t, box_t):
        """Per-level threshold -> top-k -> decode for the whole batch: ONE kernel, no host sync (replaces the per-image / per-level
        boolean-indexing loop of inference_single_image, fcosv2.py:194-238).  -> boxes (N,M,4), scores (N,M), classes (N,M), counts."""
        head = self.head
        cls_buf, box_buf, hw = head.predict(cls_t, box_t)
        return HF.fcos_decode(cls_buf, box_buf, head.scales.detach().float().contiguous(), hw, self.fpn_strides, self.num_classes,
                              head.centerness_on_reg, head.norm_reg_targets, self.pre_nms_thresh, self.pre_nms_top_n)

    @torch.no_grad()
    def nms_candidates(self, cand, image_sizes):
        """batched_nms + top-``max_detections_per_image`` (fcosv2.py:240-249) for all images on the device; the only host
        synchronisation of the whole post-processing is the final read of the per-image detection counts."""
        boxes, scores, classes, _counts = cand
        keep, nkeep = HF.batched_nms_topk(boxes, scores, classes, self.nms_thresh, self.max_detections_per_image)
        kb = torch.gather(boxes, 1, keep[:, :, None].expand(-1, -1, 4))
        ks = torch.gather(scores, 1, keep)
        kc = torch.gather(classes, 1, keep)
        nk = nkeep.cpu().tolist()
        results = []
        for i, image_size in enumerate(image_sizes):
            r = Instances(tuple(image_size))
            r.pred_boxes = Boxes(kb[i, : nk[i]])
            r.scores = ks[i, : nk[i]]
            r.pred_classes = kc[i, : nk[i]].long()
            results.append(r)
        return results

    @torch.no_grad()
    def inference(self, level_hw, cls_t, box_t, image_sizes):
        return self.nms_candidates(self.decode_candidates(cls_t, box_t), image_sizes)

    def postprocess(self, instances, batched_inputs, image_sizes):
        from ..postprocessing import detector_postprocess

        out = []
        for res, inp, size in zip(instances, batched_inputs, image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            out.append({"instances": detector_postprocess(res, h, w)})
        return out

    def preprocess_image(self, batched_inputs):
        """fcosv2.py:268-275: normalise, pad to size_divisibility, batch — one kernel per image straight into the
        NHWC(8) bf16 batch buffer (the H2D copy of the uint8 image is the only other traffic).  In training mode a batch that
        ``prefetch`` has already been given comes back as an ImageList around its FrozenPrefix (which ``backbone`` resumes from)."""
        if self.training and getattr(self, "_prefetched", None) is not None:
            images = self._take_prefetched(batched_inputs)
            if images is not None:
                return images
        imgs = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in imgs]
        Hp, Wp = ImageList.padded_size(sizes, self.backbone.size_divisibility)
        if (not HF.is_f32() and all(im.dtype == torch.uint8 and im.dim() == 3 and im.shape[0] == 3 for im in imgs) and len(imgs) <= 64
                and Hp % 4 == 0 and Wp % 4 == 0):
            # decoded uint8 images: hand the raw pixels to the backbone - a frozen stem normalises, convolves and pools them in one
            # kernel (csrc/stem_fused.hip); any other consumer materialises the NHWC(8) tensor below on demand
            return ImageList(HF.RawImageBatch([im.contiguous() for im in imgs], sizes, (Hp, Wp), self._mean, self._std), sizes)
        batch = torch.empty((len(imgs), Hp, Wp, 8), dtype=HF.ACT_DTYPE, device=self.device)
        imgs = [im if im.dtype == torch.uint8 else im.float() for im in imgs]
        HF.preprocess_batch(imgs, batch, self._mean, self._std)       # one launch for the batch
        return ImageList(batch, sizes)


@META_ARCH_REGISTRY.register()
class FCOS(FCOSV2):
    """slender_det/modeling/meta_arch/fcos/fcos.py:174-473.  Same head, target assignment and losses as FCOSV2; the reference
    class differs in two places, both of which are kept:

    * ``get_ground_truth`` (fcos.py:326-372) returns the targets LEVEL-first (one tensor per level holding all images) where
      FCOSV2 stacks them image-first; ``losses`` flattens either into the same multiset of (label, target) rows, so the three loss
      sums are equal up to fp32 summation order.  Here both run on the one batched assignment kernel; ``targets_level_first``
      exposes the reference's level-first view of it.
    * inference (fcos.py:374-464): ``inference`` only selects and decodes the candidates per feature map
      (``inference_single_feature_map``) and concatenates them per image; the class-aware NMS and the top-100 cut happen in
      ``postprocess``.  FCOSV2 does both inside ``inference``.  The detections are the same."""

    def targets_level_first(self, level_hw, gt_instances):
        """fcos.py:343-372: ``labels_level_first`` / ``reg_targets_level_first`` - per level, (N * Hl * Wl,) and (N * Hl * Wl, 4)."""
        labels, reg, _ctr, _stats = self.get_ground_truth(level_hw, gt_instances)
        sizes = [h * w for h, w in level_hw]
        return ([t.reshape(-1) for t in labels.split(sizes, dim=1)], [t.reshape(-1, 4) for t in reg.split(sizes, dim=1)])

    @torch.no_grad()
    def inference(self, level_hw, cls_t, box_t, image_sizes):
        """fcos.py:374-383: candidates of every feature map, concatenated per image, NO suppression yet."""
        cand = self.decode_candidates(cls_t, box_t)
        boxes, scores, classes, counts = cand
        cnt = counts.cpu()
        top_n = self.pre_nms_top_n
        results = []
        for i, image_size in enumerate(image_sizes):
            idx = torch.cat([torch.arange(l * top_n, l * top_n + int(cnt[i, l]), device=boxes.device) for l in range(cnt.shape[1])])
            r = Instances(tuple(image_size))
            r.pred_boxes = Boxes(boxes[i, idx])
            r.scores = scores[i, idx]
            r.pred_classes = classes[i, idx].long()
            results.append(r)
        # padded device tensors for postprocess (one NMS launch for the batch), valid only for THIS list of THESE Instances objects
        self._pending = (cand, results, [id(r) for r in results])
        return results

    def postprocess(self, instances, batched_inputs, image_sizes):
        """fcos.py:438-464: per-class NMS, top ``max_detections_per_image``, then rescale to the requested output size."""
        pend, self._pending = getattr(self, "_pending", None), None
        cand = None
        # the batched path only when the caller hands back exactly what inference() returned; a filtered / augmented / re-ordered
        # list (or other Instances) goes through the per-image path on the tensors it carries
        if pend is not None and instances is pend[1] and [id(r) for r in instances] == pend[2]:
            cand = pend[0]
        if cand is None:
            from ...layers.nms import batched_nms

            kept = []
            for r in instances:
                keep = batched_nms(r.pred_boxes.tensor, r.scores, r.pred_classes, self.nms_thresh)[: self.max_detections_per_image]
                kept.append(r[keep])
        else:
            kept = self.nms_candidates(cand, image_sizes)
        return super().postprocess(kept, batched_inputs, image_sizes)
