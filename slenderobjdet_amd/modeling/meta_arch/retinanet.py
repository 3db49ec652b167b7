"""RetinaNet on the HIP kernels (BASELINE config 3).

Mirror of detectron2's ``RetinaNet`` / ``RetinaNetHead`` as documented by the reference's in-tree copy
slender_det/modeling/meta_arch/retina/retina_rotated.py:38-474 (forward :129-183, losses :185-249, label_anchors :251-295,
head :390-474) for axis-aligned boxes, smooth-L1 regression.  Same ``cls(cfg)`` / ``forward(batched_inputs)`` contract and
loss keys (``loss_cls``, ``loss_box_reg``).

MI355X-first: anchors are labelled by one fused IoU+Matcher kernel per image that never materialises the G x 201 600 matrix;
the per-level prediction convs are one multi-level launch each writing the concatenated (N, R, K) logits / pitched deltas;
ignored anchors are skipped by label inside the loss kernels (no boolean gathers); the EMA loss normaliser lives on the device
(the reference calls ``.item()`` at :210).
"""
import math
import os

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers.nn import ConvReluML, HipConv2d, _arena_of
from ...structures import ImageList
from ..anchor_generator import grid_anchors
from ..backbone import build_backbone
from .build import META_ARCH_REGISTRY
from .fcos import FCOSV2


def _ceil8(v):
    return (v + 7) // 8 * 8


# PRED_DGRAD_PAD = False: the data gradient of the class-score conv on the per-chunk gather path (its 720 channels are no multiple of 64)
PRED_DGRAD_PAD = True
# SOD_FOCAL_FUSED=0: the focal loss as a forward pass (sum) and a backward pass (scaled gradient) over the fp32 logits - 1 GB each at batch 16
FOCAL_FUSED = os.environ.get("SOD_FOCAL_FUSED", "1") != "0"

class RetinaNetHead(nn.Module):
    def __init__(self, cfg, in_channels, num_anchors):
        super().__init__()
        self.num_classes = cfg.MODEL.RETINANET.NUM_CLASSES
        self.num_anchors = num_anchors
        n = cfg.MODEL.RETINANET.NUM_CONVS
        self.cls_subnet = nn.ModuleList([ConvReluML(in_channels) for _ in range(n)])
        self.bbox_subnet = nn.ModuleList([ConvReluML(in_channels) for _ in range(n)])
        self.kc = num_anchors * self.num_classes            # 720, a multiple of 8
        assert self.kc % 8 == 0
        self.box_pitch = _ceil8(num_anchors * 4)            # 36 -> 40
        self.cls_score = HipConv2d(in_channels, self.kc, 3, 1, 1, bias=True)
        self.bbox_pred = HipConv2d(in_channels, self.box_pitch, 3, 1, 1, bias=True)
        # rows a reference checkpoint holds (checkpoint.py drops / restores the pad): A*K scores, A*4 deltas (retina_rotated.py:432-437)
        self.cls_score.ckpt_rows, self.bbox_pred.ckpt_rows = num_anchors * self.num_classes, num_anchors * 4
        for u in list(self.cls_subnet) + list(self.bbox_subnet):
            u.conv.init_normal(0.01, 0.0)
        prior = cfg.MODEL.RETINANET.PRIOR_PROB
        with torch.no_grad():
            self.cls_score.init_normal(0.01, -math.log((1 - prior) / prior))
            self.bbox_pred.init_normal(0.01, 0.0)
            self.bbox_pred.weight[num_anchors * 4:].zero_()

    def run_towers(self, feats):
        c, b = list(feats), list(feats)
        prev = None
        for u in self.cls_subnet:          # consecutive units: the consumer's data gradient applies the producer's ReLU mask
            c, prev = u(c, chained=prev), u
        prev = None
        for u in self.bbox_subnet:
            b, prev = u(b, chained=prev), u
        return c, b

    def predict(self, cls_t, box_t):
        self.cls_score.prepare()
        self.bbox_pred.prepare()
        N = cls_t[0].shape[0]
        hw = [(t.shape[1], t.shape[2]) for t in cls_t]
        P = sum(h * w for h, w in hw)
        dev = cls_t[0].device
        cls_buf = torch.empty((N, P, self.kc), dtype=torch.float32, device=dev)        # == (N, R, num_classes)
        box_buf = torch.empty((N, P, self.box_pitch), dtype=torch.float32, device=dev)
        offs, off = [], 0
        for h, w in hw:
            offs.append(off)
            off += h * w
        HF.conv2d_fwd_ml(list(cls_t), self.cls_score.w_bf16, self.cls_score.bias_eff, 1, 1, 1, out_f32=True,
                         outs=[cls_buf.view(-1)[o * self.kc:] for o in offs], y_img_stride=P * self.kc)
        HF.conv2d_fwd_ml(list(box_t), self.bbox_pred.w_bf16, self.bbox_pred.bias_eff, 1, 1, 1, out_f32=True,
                         outs=[box_buf.view(-1)[o * self.box_pitch:] for o in offs], y_img_stride=P * self.box_pitch)
        return cls_buf, box_buf, hw, offs


class _RetinaLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, weight, gt_labels, gt_deltas, *towers):
        head = model.head
        nl = len(towers) // 2
        cls_t, box_t = list(towers[:nl]), list(towers[nl:])
        cls_buf, box_buf, hw, offs = head.predict(cls_t, box_t)
        N, P = cls_buf.shape[0], cls_buf.shape[1]
        A, K = head.num_anchors, head.num_classes
        R = P * A
        if model.box_reg_loss_type == "giou":       # gt_deltas then holds the matched gt BOXES (N,R,4)
            sums = HF.retina_giou_loss_fwd(box_buf, head.box_pitch, gt_labels, model.anchors_for(hw), gt_deltas, N, R, A, K, model.bbox_reg_weights,
                                           model.scale_clamp, model.loss_normalizer, model.loss_normalizer_momentum)
        else:
            sums = HF.retina_box_loss_fwd(box_buf, head.box_pitch, gt_labels, gt_deltas, N, R, A, K, model.smooth_l1_loss_beta,
                                          model.loss_normalizer, model.loss_normalizer_momentum)    # also advances the EMA normaliser
        dcls_u = None
        # the one-pass kernel is vectorised only (sod_sigmoid_focal_loss_fwd_grad: 4 classes per lane, 32-bit element offsets): other
        # layouts (NUM_CLASSES % 4 != 0, N * R * K >= 2^31) take the two-pass entry points as before round 4
        fusable = K % 4 == 0 and N * R * K < 2 ** 31
        if FOCAL_FUSED and fusable and not HF.is_f32() and any(ctx.needs_input_grad[4:]):      # a backward pass follows
            focal_sum, dcls_u = HF.focal_loss_fwd_grad(cls_buf.view(N * R, K), gt_labels.view(-1), model.focal_loss_alpha, model.focal_loss_gamma)
        else:
            focal_sum, _ = HF.focal_loss_fwd(cls_buf.view(N * R, K), gt_labels.view(-1), None, model.focal_loss_alpha, model.focal_loss_gamma)
        out = torch.stack([focal_sum[0], sums[0]]) / model.loss_normalizer
        ctx.model, ctx.geo = model, (hw, offs, N, P, A, K, R)
        ctx.fused = dcls_u is not None
        ctx.save_for_backward(cls_buf if dcls_u is None else dcls_u, box_buf, gt_labels, gt_deltas, model.loss_normalizer.clone(), *towers)
        arena = _arena_of(head)
        if arena is not None:
            for p in (head.cls_score.weight, head.cls_score.bias, head.bbox_pred.weight, head.bbox_pred.bias):
                arena.note_use(p)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g2):
        model = ctx.model
        head = model.head
        hw, offs, N, P, A, K, R = ctx.geo
        cls_buf, box_buf, gt_labels, gt_deltas, norm = ctx.saved_tensors[:5]
        towers = ctx.saved_tensors[5:]
        nl = len(towers) // 2
        cls_t, box_t = towers[:nl], towers[nl:]
        g2 = g2.contiguous().float()
        arena = _arena_of(head)
        dev = cls_buf.device
        if ctx.fused:
            # the forward pass already wrote the un-scaled class-score gradient (sod_sigmoid_focal_loss_fwd_grad); what is left of the
            # backward pass is the scalar g / normaliser, applied by the three consumers below
            dcls, cls_scale = cls_buf.view(N, P, head.kc), (g2[0:1].contiguous(), norm)
        else:
            dcls, cls_scale = HF.focal_loss_bwd(cls_buf.view(N * R, K), gt_labels.view(-1), None, model.focal_loss_alpha, model.focal_loss_gamma,
                                                scale_num=g2[0:1], scale_den=norm, den_mul=1.0, den_min=1e-12,
                                                out_bf16=not HF.is_f32()).view(N, P, head.kc), None
        dbox = torch.zeros((N, P, head.box_pitch), dtype=HF.ACT_DTYPE, device=dev)
        if model.box_reg_loss_type == "giou":
            HF.retina_giou_loss_bwd(box_buf, head.box_pitch, gt_labels, model.anchors_for(hw), gt_deltas, N, R, A, K, model.bbox_reg_weights,
                                    model.scale_clamp, g2[1:2], norm, dbox)
        else:
            HF.retina_box_loss_bwd(box_buf, head.box_pitch, gt_labels, gt_deltas, N, R, A, K, model.smooth_l1_loss_beta, g2[1:2], norm, dbox)
        grads = []
        for pred, dbuf, kk, tower, scale in ((head.cls_score, dcls, head.kc, cls_t, cls_scale), (head.bbox_pred, dbox, head.box_pitch, box_t, None)):
            dys = [dbuf.view(-1)[o * kk:] for o in offs]
            wt = pred.wt_bf16
            if scale is None:
                HF.conv2d_wgrad_ml(dys, list(tower), arena.grad_view(pred.weight), 3, 3, 1, 1, 1, dy_img_stride=P * kk, K=kk)
                HF.bias_grad(dbuf, arena.grad_view(pred.bias), N, P, kk)
            else:
                qs = (scale[0] / scale[1].clamp_min(1e-12)).expand(kk).contiguous()          # the scalar per output channel, on the device
                HF.conv2d_wgrad_ml(dys, list(tower), arena.grad_view(pred.weight), 3, 3, 1, 1, 1, dy_img_stride=P * kk, K=kk, qscale=qs)
                HF.bias_grad(dbuf, arena.grad_view(pred.bias), N, P, kk, scale_num=scale[0], scale_den=scale[1], den_mul=1.0, den_min=1e-12)
                # the data gradient is linear in the weights: (w * s) rounded to bf16 once, from the fp32 master, as the ordinary copy is
                _, wt = HF.weight_prep(pred.weight.detach().contiguous(), qs, want_krsc=False)
            arena.mark_ready(pred.weight)
            arena.mark_ready(pred.bias)
            if PRED_DGRAD_PAD and not HF.is_f32() and kk % 64 and kk >= 256:
                # 720 class scores: contract over a 768-wide zero-padded copy of the transposed weights on the linear K loops (the 256x256
                # kernel) instead of the per-chunk gather path of the 128x128 kernel (sod_conv2d_dgrad_ml_kpitch)
                wt_pad = torch.nn.functional.pad(wt, (0, (kk + 63) // 64 * 64 - kk))
                grads.append(HF.conv2d_dgrad_ml(dys, wt_pad, hw, 1, 1, 1, dy_img_stride=P * kk, N=N, k_pitch=kk))
            else:
                grads.append(HF.conv2d_dgrad_ml(dys, wt, hw, 1, 1, 1, dy_img_stride=P * kk, N=N))
        return (None, None, None, None, *grads[0], *grads[1])


@META_ARCH_REGISTRY.register()
class RetinaNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RETINANET
        self.num_classes = r.NUM_CLASSES
        self.in_features = r.IN_FEATURES
        self.focal_loss_alpha, self.focal_loss_gamma = r.FOCAL_LOSS_ALPHA, r.FOCAL_LOSS_GAMMA
        self.smooth_l1_loss_beta = r.SMOOTH_L1_LOSS_BETA
        self.box_reg_loss_type = r.BBOX_REG_LOSS_TYPE
        if self.box_reg_loss_type not in ("smooth_l1", "giou"):
            raise ValueError(f"Invalid bbox reg loss type '{self.box_reg_loss_type}'")          # retina_rotated.py:243-244
        self.scale_clamp = math.log(1000.0 / 16)
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        self.bbox_reg_weights = tuple(r.BBOX_REG_WEIGHTS)
        self.score_threshold, self.topk_candidates, self.nms_threshold = r.SCORE_THRESH_TEST, r.TOPK_CANDIDATES_TEST, r.NMS_THRESH_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.strides = [shapes[f].stride for f in self.in_features]
        ag = cfg.MODEL.ANCHOR_GENERATOR
        self.anchor_sizes, self.anchor_ratios, self.anchor_offset = [list(s) for s in ag.SIZES], [list(a) for a in ag.ASPECT_RATIOS], ag.OFFSET
        num_anchors = len(self.anchor_sizes[0]) * len(self.anchor_ratios[0])
        self.head = RetinaNetHead(cfg, shapes[self.in_features[0]].channels, num_anchors)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.register_buffer("loss_normalizer", torch.tensor([100.0]))     # retina_rotated.py:87-88
        self.loss_normalizer_momentum = 0.9
        self._anchor_cache = {}

    @property
    def device(self):
        return self.pixel_mean.device

    preprocess_image = FCOSV2.preprocess_image
    prefetch, _take_prefetched = FCOSV2.prefetch, FCOSV2._take_prefetched

    def anchors_for(self, level_hw):
        key = tuple(level_hw)
        if key not in self._anchor_cache:
            per_level = grid_anchors(level_hw, self.strides, self.anchor_sizes, self.anchor_ratios, self.anchor_offset, self.device)
            self._anchor_cache[key] = torch.cat(per_level).contiguous()
        return self._anchor_cache[key]

    @torch.no_grad()
    def label_anchors(self, anchors, gt_instances):
        """retina_rotated.py:251-295 — returns gt_labels (N,R) int32 in {-1, 0..K-1, K} and gt_deltas (N,R,4)."""
        N, R = len(gt_instances), anchors.shape[0]
        labels = torch.empty((N, R), dtype=torch.int32, device=anchors.device)
        deltas = torch.empty((N, R, 4), dtype=torch.float32, device=anchors.device)
        for i, g in enumerate(gt_instances):
            boxes = g.gt_boxes.tensor.float().contiguous()
            classes = g.gt_classes.to(torch.int32).contiguous()
            _, matches, mlab = HF.anchor_match(boxes, anchors, self.iou_thresholds, self.iou_labels, True)
            HF.retina_targets(anchors, boxes, classes, matches, mlab, self.num_classes, self.bbox_reg_weights, labels[i], deltas[i])
            if self.box_reg_loss_type == "giou":      # the GIoU loss compares decoded boxes with the matched gt boxes themselves
                deltas[i] = boxes[matches.long()] if len(boxes) else 0.0
        return labels, deltas

    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        features = [features[f] for f in self.in_features]
        level_hw = [tuple(f.shape[1:3]) for f in features]
        anchors = self.anchors_for(level_hw)
        cls_t, box_t = self.head.run_towers(features)
        if self.training:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            gt_labels, gt_deltas = self.label_anchors(anchors, gt_instances)
            out = _RetinaLossFn.apply(self, self.head.cls_score.weight, gt_labels, gt_deltas, *cls_t, *box_t)
            return {"loss_cls": out[0], "loss_box_reg": out[1]}
        with torch.no_grad():
            cls_buf, box_buf, hw, offs = self.head.predict(cls_t, box_t)
            results = self.inference(level_hw, cls_buf, box_buf, offs, images.image_sizes)
        return self.postprocess(results, batched_inputs, images.image_sizes)

    @torch.no_grad()
    def inference(self, level_hw, cls_buf, box_buf, offs, image_sizes):
        """retina_rotated.py:296-377 for the whole batch on the device: per level sigmoid over (HWA x K), top-k, score threshold
        (one selection launch), decode the surviving anchors (Box2BoxTransform.apply_deltas, one kernel), class-aware NMS and
        the top detections (batched NMS); the only host read is the final per-image detection count."""
        from ...structures import Boxes, Instances
        from ..box_regression import Box2BoxTransform

        A, K = self.head.num_anchors, self.num_classes
        N, P = cls_buf.shape[:2]
        transform = Box2BoxTransform(weights=self.bbox_reg_weights)
        anchors = torch.cat(grid_anchors(level_hw, self.strides, self.anchor_sizes, self.anchor_ratios, self.anchor_offset, self.device))   # (P*A, 4)
        rows_per_level = [h * w * A for h, w in level_hw]
        logits = cls_buf[..., : A * K].reshape(N, P * A, K) if cls_buf.shape[-1] != A * K else cls_buf.view(N, P * A, K)
        rows, scores, classes, _counts = HF.dense_topk_select(logits.contiguous(), rows_per_level, K, self.score_threshold, self.topk_candidates)
        top_n = self.topk_candidates
        row0 = torch.tensor([sum(rows_per_level[:l]) for l in range(len(rows_per_level))], dtype=torch.int64, device=rows.device)
        grow = rows.long() + row0.repeat_interleave(top_n)[None]                       # anchor index over all levels, (N, M)
        deltas = box_buf[..., : A * 4].reshape(N, P * A, 4)
        sel_d = torch.gather(deltas, 1, grow[:, :, None].expand(-1, -1, 4)).reshape(-1, 4).contiguous()
        sel_a = anchors[grow.reshape(-1)].contiguous()
        boxes = transform.apply_deltas(sel_d, sel_a).view(N, -1, 4)
        boxes = torch.where(torch.isfinite(scores)[:, :, None], boxes, torch.zeros_like(boxes)).contiguous()       # empty slots
        keep, nkeep = HF.batched_nms_topk(boxes, scores, classes, self.nms_threshold, self.max_detections_per_image)
        kb = torch.gather(boxes, 1, keep[:, :, None].expand(-1, -1, 4))
        ks, kc = torch.gather(scores, 1, keep), torch.gather(classes, 1, keep)
        nk = nkeep.cpu().tolist()
        results = []
        for i, image_size in enumerate(image_sizes):
            r = Instances(tuple(image_size))
            r.pred_boxes, r.scores, r.pred_classes = Boxes(kb[i, : nk[i]]), ks[i, : nk[i]], kc[i, : nk[i]].long()
            results.append(r)
        return results

    postprocess = FCOSV2.postprocess
