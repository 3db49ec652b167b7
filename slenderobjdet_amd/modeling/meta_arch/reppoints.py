"""RepPointsDetector on the HIP kernels (BASELINE config 4, configs/rep-points/rep_points_detector_R_50_FPN_1x.yaml).

Mirror of slender_det/modeling/meta_arch/reppoints/rpd.py:45-798: same ``cls(cfg)`` constructor contract (RETINANET.* keys,
``MODEL.PROPOSAL_GENERATOR.SAMPLE_MODE`` in {points, nearest_points, inside}), same ``forward(batched_inputs)`` contract and
loss keys (``loss_cls``, ``loss_localization_init``, ``loss_localization_refine``; rpd.py:400-402).

How it maps to the MI355X kernels:
  * ``cls_conv`` / ``reg_conv`` (3 x [conv3x3, GN(32), ReLU], rpd.py:191-204) are multi-level ConvGnRelu launches;
  * point offsets are fp32 NHWC rows pitched 18 -> 24; the xy->yx flip, ``- dcn_base_offset`` and the 0.1 gradient multiplier
    (rpd.py:621-635) are one kernel whose backward is the same kernel with another scale;
  * the two DeformConv layers are gather + MFMA GEMM with the ReLU of ``logits[0]`` / ``offsets_refine[0]`` in the epilogue;
  * ``logits`` / ``offsets_refine`` 1x1 convs, points2bbox (minmax), label assignment, the three losses and the EMA normaliser are
    one autograd node; labels never leave the device and nothing calls ``.item()`` (the reference does at rpd.py:367).
"""
import math

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers.deform_conv import DeformConv
from ...layers.nn import ConvGnRelu, ConvML, HipConv2d, _arena_of
from ...structures import Boxes, Instances
from ..backbone import build_backbone
from .build import META_ARCH_REGISTRY
from .fcos import FCOSV2


class _DcnOffsetFn(torch.autograd.Function):
    """dcn_offset = flip_xy((1 - m) * pts.detach() + m * pts) - dcn_base_offset   (rpd.py:621-635)."""

    @staticmethod
    def forward(ctx, pts, num_points, gradient_mul, flip_xy=True):
        ctx.cfg = (num_points, gradient_mul, flip_xy)
        return HF.reppoints_dcn_offset(pts, num_points, 1.0, True, flip_xy)

    @staticmethod
    @once_differentiable
    def backward(ctx, doff):
        num_points, gradient_mul, flip_xy = ctx.cfg
        return HF.reppoints_dcn_offset(doff.contiguous(), num_points, gradient_mul, False, flip_xy), None, None, None


class _RepPointsLossFn(torch.autograd.Function):
    """logits / offsets_refine convs + points2bbox + get_ground_truth + losses (rpd.py:636-671) as one node over
    (offsets_init, relu(dcn_cls), relu(dcn_reg)) of every level."""

    @staticmethod
    def forward(ctx, model, weight, gt_instances, image_sizes, *tensors):
        nl = len(tensors) // 3
        oi, cf, rf = list(tensors[:nl]), list(tensors[nl:2 * nl]), list(tensors[2 * nl:])
        logits_buf, rdelta, init_boxes, init_arg, refine_boxes, refine_arg, geo = model.predict(oi, cf, rf)
        hw, offs, X = geo
        N, K = logits_buf.shape[0], model.num_classes
        centers, strides, lvl_start = model.point_grid(hw)
        obj, init_lab, cls, refine_lab = model.get_ground_truth(centers, strides, lvl_start, init_boxes, gt_instances, image_sizes)
        focal_sum, _ = HF.focal_loss_fwd(logits_buf.view(N * X, K), cls.view(-1), None, model.focal_loss_alpha, model.focal_loss_gamma)
        strides = model.box_norm(strides)        # the box losses divide by 4 * this
        init_sums = HF.reppoints_box_loss_fwd(init_boxes, init_lab, obj, strides, -1, model.smooth_l1_beta)
        refine_sums = HF.reppoints_box_loss_fwd(refine_boxes, refine_lab, cls, strides, K, model.smooth_l1_beta)
        out3 = HF.reppoints_finalize(focal_sum, init_sums, refine_sums, model.loss_normalizer, model.loss_normalizer_momentum,
                                     model.normalizer_images(N), model.loss_init_weight)
        ctx.model, ctx.geo, ctx.nl = model, geo, nl
        ctx.moment = getattr(model, "transform_method", "minmax") == "moment"
        ctx.save_for_backward(logits_buf, init_boxes, init_arg, refine_boxes, refine_arg, obj, init_lab, cls, refine_lab, strides,
                              init_sums, model.loss_normalizer.clone(), *oi, *cf, *rf, *(rdelta if ctx.moment else ()))
        model.last_targets = (obj, init_lab, cls, refine_lab)
        arena = _arena_of(model.logits)
        if arena is not None:
            for p in (model.logits.weight, model.logits.bias, model.offsets_refine.weight, model.offsets_refine.bias):
                arena.note_use(p)
            if ctx.moment:
                arena.note_use(model.moment_transfer)
        return out3

    @staticmethod
    @once_differentiable
    def backward(ctx, g3):
        model, (hw, offs, X), nl = ctx.model, ctx.geo, ctx.nl
        logits_buf, init_boxes, init_arg, refine_boxes, refine_arg, obj, init_lab, cls, refine_lab, strides, init_sums, norm = ctx.saved_tensors[:12]
        rest = ctx.saved_tensors[12:]
        oi, cf, rf = rest[:nl], rest[nl:2 * nl], rest[2 * nl:3 * nl]
        rdelta = rest[3 * nl:]
        g3 = g3.contiguous().float()
        N, K, P, ld = logits_buf.shape[0], model.num_classes, getattr(model, "box_points", model.num_points), model.pts_ld
        arena = _arena_of(model.logits)
        # classification branch
        dlogits = HF.focal_loss_bwd(logits_buf.view(N * X, K), cls.view(-1), None, model.focal_loss_alpha, model.focal_loss_gamma,
                                    scale_num=g3[0:1], scale_den=norm, den_mul=1.0, den_min=1.0, out_bf16=not HF.is_f32())
        dys = [dlogits.view(-1)[o * K:] for o in offs]
        HF.conv2d_wgrad_ml(dys, list(cf), arena.grad_view(model.logits.weight), 1, 1, 1, 0, 1, dy_img_stride=X * K, K=K)
        arena.mark_ready(model.logits.weight)
        HF.bias_grad(dlogits, arena.grad_view(model.logits.bias), N, X, K)
        arena.mark_ready(model.logits.bias)
        dcf = HF.conv2d_dgrad_ml(dys, model.logits.wt_bf16, hw, 1, 0, 1, dy_img_stride=X * K, N=N)
        # localisation branches: d(box) -> arg points of the minmax transform
        d_refine = HF.reppoints_box_loss_bwd(refine_boxes, refine_lab, cls, strides, K, model.smooth_l1_beta, g3[2:3], norm, 1.0, 1.0)
        d_init = HF.reppoints_box_loss_bwd(init_boxes, init_lab, obj, strides, -1, model.smooth_l1_beta, g3[1:2], init_sums[1:2], 1.0,
                                           model.loss_init_weight)
        doi, drd = [], []
        f32 = HF.is_f32()          # validation mode: the gradient rows of offsets_refine stay fp32
        for l, (h, w) in enumerate(hw):
            o, ps = offs[l], model.point_scales[l]
            shape = (N, h, w, ld)
            if ctx.moment:
                mt, dmt = model.moment_transfer.detach(), arena.grad_view(model.moment_transfer)
                add = oi[l] if model.res_refine else None
                r32, d16 = HF.points2bbox_moment_bwd(d_refine.view(-1)[o * 4:], X * 4, rdelta[l], add, model.strides[l], ps, model.num_points, mt,
                                                     model.moment_mul, dmt, want_f32=f32, want_bf16=not f32)
                d32, _ = HF.points2bbox_moment_bwd(d_init.view(-1)[o * 4:], X * 4, oi[l], None, model.strides[l], ps, model.num_points, mt,
                                                   model.moment_mul, dmt)
                drd.append(r32 if f32 else d16)
                doi.append(d32)
                continue
            r32, d16 = HF.points2bbox_bwd(d_refine.view(-1)[o * 4:], X * 4, refine_arg.view(-1)[o:], X, shape, ps, P, want_f32=f32, want_bf16=not f32)
            d32, _ = HF.points2bbox_bwd(d_init.view(-1)[o * 4:], X * 4, init_arg.view(-1)[o:], X, shape, ps, P)
            drd.append(r32 if f32 else d16)
            doi.append(d32)
        if ctx.moment:
            arena.mark_ready(model.moment_transfer)
        ref = model.offsets_refine
        HF.conv2d_wgrad_ml(drd, list(rf), arena.grad_view(ref.weight), 1, 1, 1, 0, 1)
        arena.mark_ready(ref.weight)
        dbias = arena.grad_view(ref.bias)
        for (h, w), d in zip(hw, drd):
            HF.bias_grad(d, dbias, N, h * w, ld)
        arena.mark_ready(ref.bias)
        drf = HF.conv2d_dgrad_ml(drd, ref.wt_bf16, hw, 1, 0, 1)
        return (None, None, None, None, *doi, *dcf, *drf)


@META_ARCH_REGISTRY.register()
class RepPointsDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RETINANET                                  # rpd.py:50-60: shared with RetinaNet
        self.num_classes = r.NUM_CLASSES
        self.in_features = r.IN_FEATURES
        self.focal_loss_alpha, self.focal_loss_gamma = r.FOCAL_LOSS_ALPHA, r.FOCAL_LOSS_GAMMA
        self.topk_candidates, self.score_threshold, self.nms_threshold = r.TOPK_CANDIDATES_TEST, r.SCORE_THRESH_TEST, r.NMS_THRESH_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        # rpd.py:62-77: fixed RepPoints hyper-parameters
        self.point_feat_channels = 256
        self.num_stacked_convs = 3
        self.num_points = 9
        self.gradient_mul = 0.1
        self.point_base_scale = 4
        self.point_strides = [8, 16, 32, 64, 128]
        self.point_scales = [1, 2, 4, 8, 16]                     # rpd.py:646-647
        self.smooth_l1_beta = 0.11
        self.loss_init_weight, self.loss_refine_weight = 0.5, 1.0
        self.transform_method = "minmax"
        self.pts_ld = (2 * self.num_points + 7) // 8 * 8          # 18 -> 24 floats per row
        self.dcn_kernel = int(math.sqrt(self.num_points))
        self.dcn_pad = (self.dcn_kernel - 1) // 2
        assert self.dcn_kernel * self.dcn_kernel == self.num_points and self.dcn_kernel % 2 == 1

        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.strides = [shapes[f].stride for f in self.in_features]
        in_ch = shapes[self.in_features[0]].channels
        assert in_ch == self.point_feat_channels, "RepPointsDetector expects 256-channel FPN features"
        C = self.point_feat_channels
        self.cls_conv = nn.ModuleList([ConvGnRelu(C) for _ in range(self.num_stacked_convs)])
        self.reg_conv = nn.ModuleList([ConvGnRelu(C) for _ in range(self.num_stacked_convs)])
        # the ReLU that opens ``self.logits`` / ``self.offsets_refine`` (rpd.py:161-167) is each DeformConv's only consumer
        self.deform_cls_conv = DeformConv(C, C, self.dcn_kernel, 1, self.dcn_pad, relu=True)
        self.deform_reg_conv = DeformConv(C, C, self.dcn_kernel, 1, self.dcn_pad, relu=True)
        self.offsets_init = nn.ModuleList([ConvML(C, C, 3, 1, relu=True), ConvML(C, self.pts_ld, 1, 0, out_f32=True)])
        self.offsets_refine = HipConv2d(C, self.pts_ld, 1, 1, 0, bias=True)
        self.logits = HipConv2d(C, self.num_classes, 1, 1, 0, bias=True)
        assert self.num_classes % 8 == 0, "RETINANET.NUM_CLASSES must be a multiple of 8 for the fused logits conv"
        # rpd.py:169-185 (only nn.Conv2d instances are re-initialised; DeformConv keeps its own init, logits keeps torch's default)
        for u in list(self.cls_conv) + list(self.reg_conv):
            u.conv.init_normal(0.01, 0.0)
        npt = 2 * self.num_points
        with torch.no_grad():
            self.offsets_init[0].conv.init_normal(0.01, 0.0)
            self.offsets_init[1].conv.init_normal(0.01, 0.0)
            self.offsets_init[1].conv.weight[npt:].zero_()
            self.offsets_refine.init_normal(0.01, 0.0)
            self.offsets_refine.weight[npt:].zero_()
            bound = 1.0 / math.sqrt(C)                               # nn.Conv2d default: kaiming_uniform_(a=sqrt(5))
            self.logits.weight.uniform_(-bound, bound)
            self.logits.bias.fill_(float(-math.log((1 - 0.01) / 0.01)))

        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.register_buffer("loss_normalizer", torch.tensor([20.0]))   # rpd.py:123
        self.loss_normalizer_momentum = 0.9
        mode = cfg.MODEL.PROPOSAL_GENERATOR.SAMPLE_MODE                  # rpd.py:126-135
        if mode not in HF.RP_MATCH_MODES:
            raise AssertionError(f"MODEL.PROPOSAL_GENERATOR.SAMPLE_MODE {mode!r} not in {sorted(HF.RP_MATCH_MODES)}")
        self.sample_mode = mode
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)   # bbox_matcher, rpd.py:138-142
        self._grid_cache = {}
        self.last_targets = None

    @property
    def device(self):
        return self.pixel_mean.device

    preprocess_image = FCOSV2.preprocess_image
    prefetch, _take_prefetched = FCOSV2.prefetch, FCOSV2._take_prefetched
    postprocess = FCOSV2.postprocess
    res_refine = True                       # offsets_refine(...) + offsets_init.detach()  (rpd.py:639-642)

    def box_norm(self, strides):            # smooth-L1 inputs are divided by 4 * stride (rpd.py:384,392)
        return strides

    def normalizer_images(self, n):         # num_foreground = #fg / N feeds the EMA normaliser (rpd.py:366-376)
        return n

    # ------------------------------------------------------------------ geometry
    def point_grid(self, hw):
        """get_center_grid (rpd.py:206-219) concatenated over levels: centers (X,2) = (j, i) * stride, strides (X,), level starts."""
        key = tuple(hw)
        if key not in self._grid_cache:
            dev = self.device
            cs, ss, starts = [], [], [0]
            for (h, w), s in zip(hw, self.strides):
                ys = torch.arange(h, dtype=torch.float32, device=dev) * s
                xs = torch.arange(w, dtype=torch.float32, device=dev) * s
                gy, gx = torch.meshgrid(ys, xs, indexing="ij")
                cs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1))
                ss.append(torch.full((h * w,), float(s), dtype=torch.float32, device=dev))
                starts.append(starts[-1] + h * w)
            self._grid_cache[key] = (torch.cat(cs).contiguous(), torch.cat(ss).contiguous(),
                                     torch.tensor(starts, dtype=torch.int32, device=dev))
        return self._grid_cache[key]

    # ------------------------------------------------------------------ head
    def run_head(self, features):
        # (The two towers on two streams, as the FCOS head runs its towers: 448.7 vs 450.2 img/s over three alternating pairs of 60-step
        # runs in round 5 - no gain here: the DeformConv kernels behind the towers already keep the second stream's share of the chip busy.)
        cls_f, reg_f = list(features), list(features)
        for u in self.cls_conv:
            cls_f = u(cls_f)
        for u in self.reg_conv:
            reg_f = u(reg_f)
        oi = self.offsets_init[1](self.offsets_init[0](reg_f))         # per level (N,H,W,24) fp32, x/y interleaved
        cf, rf = [], []
        for l in range(len(features)):
            off = _DcnOffsetFn.apply(oi[l], self.num_points, self.gradient_mul)
            cf.append(self.deform_cls_conv(cls_f[l], off, off_ld=self.pts_ld))
            rf.append(self.deform_reg_conv(reg_f[l], off, off_ld=self.pts_ld))
        return oi, cf, rf

    def predict(self, oi, cf, rf):
        """logits (N,X,K) fp32, refine deltas per level, init / refine boxes (N,X,4) with the arg-point indices."""
        self.logits.prepare()
        self.offsets_refine.prepare()
        N = cf[0].shape[0]
        hw = [(t.shape[1], t.shape[2]) for t in cf]
        offs, off = [], 0
        for h, w in hw:
            offs.append(off)
            off += h * w
        X, K, dev = off, self.num_classes, cf[0].device
        logits_buf = torch.empty((N, X, K), dtype=torch.float32, device=dev)
        HF.conv2d_fwd_ml(list(cf), self.logits.w_bf16, self.logits.bias_eff, 1, 0, 1, out_f32=True,
                         outs=[logits_buf.view(-1)[o * K:] for o in offs], y_img_stride=X * K)
        rdelta = HF.conv2d_fwd_ml(list(rf), self.offsets_refine.w_bf16, self.offsets_refine.bias_eff, 1, 0, 1, out_f32=True)
        init_boxes = torch.empty((N, X, 4), dtype=torch.float32, device=dev)
        refine_boxes = torch.empty((N, X, 4), dtype=torch.float32, device=dev)
        init_arg = torch.empty((N, X), dtype=torch.int32, device=dev)
        refine_arg = torch.empty((N, X), dtype=torch.int32, device=dev)
        method = getattr(self, "transform_method", "minmax")
        bp = getattr(self, "box_points", self.num_points)         # "partial_minmax": the first four points only (pointset_head.py:322-327)
        for l in range(len(hw)):
            o, s, ps = offs[l], self.strides[l], self.point_scales[l]
            if method == "moment":                                # mean -+ std * exp(moment_transfer) (pointset_head.py:328-343)
                mt = self.moment_transfer.detach()
                HF.points2bbox_moment_fwd(oi[l], None, s, ps, self.num_points, mt, init_boxes.view(-1)[o * 4:], X * 4)
                HF.points2bbox_moment_fwd(rdelta[l], oi[l] if self.res_refine else None, s, ps, self.num_points, mt,
                                          refine_boxes.view(-1)[o * 4:], X * 4)
                continue
            HF.points2bbox_fwd(oi[l], None, s, ps, bp, init_boxes.view(-1)[o * 4:], X * 4, init_arg.view(-1)[o:], X)
            # offsets_refine(...) + offsets_init.detach()  (rpd.py:639-642)
            HF.points2bbox_fwd(rdelta[l], oi[l] if self.res_refine else None, s, ps, bp, refine_boxes.view(-1)[o * 4:], X * 4,
                               refine_arg.view(-1)[o:], X)
        return logits_buf, rdelta, init_boxes, init_arg, refine_boxes, refine_arg, (hw, offs, X)

    # ------------------------------------------------------------------ targets (rpd.py:276-333)
    @torch.no_grad()
    def get_ground_truth(self, centers, strides, lvl_start, init_boxes, gt_instances, image_sizes):
        """Returns objectness (N,X) int32, init box labels (N,X,4), cls labels (N,X) int32 in {-1, 0..K-1, K}, refine box labels."""
        dev = centers.device
        N, X = init_boxes.shape[:2]
        counts = [len(g) for g in gt_instances]
        if min(counts) == 0:
            raise ValueError("No gt or bboxes")                      # rep_matcher.py:38-39
        box_off = torch.tensor([0] + counts, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev, non_blocking=True)
        boxes = torch.cat([g.gt_boxes.tensor for g in gt_instances]).float().contiguous()
        classes = torch.cat([g.gt_classes for g in gt_instances]).to(torch.int32).contiguous()
        obj, init_lab = HF.reppoints_point_match(centers, strides, lvl_start, boxes, box_off, N, max(counts), self.sample_mode,
                                                 float(self.point_base_scale))
        vals = torch.empty((N, X), dtype=torch.float32, device=dev)
        matches = torch.empty((N, X), dtype=torch.int32, device=dev)
        mlab = torch.empty((N, X), dtype=torch.int8, device=dev)
        b0 = 0
        for i, c in enumerate(counts):       # pairwise_iou(gt, init boxes) + Matcher(allow_low_quality_matches=True), rpd.py:311-314
            HF.anchor_match(boxes[b0:b0 + c], init_boxes[i], self.iou_thresholds, self.iou_labels, True, out=(vals[i], matches[i], mlab[i]))
            b0 += c
        image_hw = torch.tensor([[float(h), float(w)] for h, w in image_sizes], dtype=torch.float32).to(dev, non_blocking=True)
        cls, refine_lab = HF.reppoints_labels(matches, mlab, boxes, classes, box_off, centers, image_hw, self.num_classes, obj)
        return obj, init_lab, cls, refine_lab

    # ------------------------------------------------------------------ forward (rpd.py:589-681)
    def forward(self, batched_inputs):
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs] if "instances" in batched_inputs[0] else None
        features = self.backbone(images.tensor)
        features = [features[f] for f in self.in_features]
        oi, cf, rf = self.run_head(features)
        if self.training:
            out3 = _RepPointsLossFn.apply(self, self.logits.weight, gt_instances, images.image_sizes, *oi, *cf, *rf)
            return {"loss_cls": out3[0], "loss_localization_init": out3[1], "loss_localization_refine": out3[2]}
        with torch.no_grad():
            logits_buf, _, init_boxes, _, refine_boxes, _, geo = self.predict(oi, cf, rf)
            results = self.inference(logits_buf, init_boxes, refine_boxes, geo, images.image_sizes)
        return self.postprocess(results, batched_inputs, images.image_sizes)

    @torch.no_grad()
    def inference(self, logits, init_boxes, refine_boxes, geo, image_sizes):
        """rpd.py:701-789 for the whole batch on the device: per level the best class of every point, top-k by score, threshold
        (one selection launch), then class-aware NMS + top detections (batched NMS); one host read of the detection counts."""
        hw, offs, X = geo
        N = logits.shape[0]
        rows_per_level = [h * w for h, w in hw]
        K = self.num_classes
        rows, scores, classes, _counts = HF.dense_topk_select(logits.float().contiguous(), rows_per_level, K, self.score_threshold, self.topk_candidates,
                                                              by_row_max=True)
        top_n = self.topk_candidates
        row0 = torch.tensor(list(offs), dtype=torch.int64, device=rows.device)
        grow = rows.long() + row0.repeat_interleave(top_n)[None]
        valid = torch.isfinite(scores)[:, :, None]
        boxes = torch.where(valid, torch.gather(refine_boxes.float(), 1, grow[:, :, None].expand(-1, -1, 4)), torch.zeros((), device=rows.device)).contiguous()
        init = torch.gather(init_boxes.float(), 1, grow[:, :, None].expand(-1, -1, 4))
        keep, nkeep = HF.batched_nms_topk(boxes, scores, classes, self.nms_threshold, self.max_detections_per_image)
        kb = torch.gather(boxes, 1, keep[:, :, None].expand(-1, -1, 4))
        ki = torch.gather(init, 1, keep[:, :, None].expand(-1, -1, 4))
        ks, kc = torch.gather(scores, 1, keep), torch.gather(classes, 1, keep)
        nk = nkeep.cpu().tolist()
        results = []
        for i, image_size in enumerate(image_sizes):
            r = Instances(tuple(image_size))
            r.pred_boxes = Boxes(kb[i, : nk[i]])
            r.scores = ks[i, : nk[i]]
            r.pred_classes = kc[i, : nk[i]].long()
            r.init_boxes = ki[i, : nk[i]]
            results.append(r)
        return results
