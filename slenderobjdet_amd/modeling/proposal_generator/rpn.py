"""RPN / RRPN on the HIP kernels (BASELINE config 5: configs/rotated/Base-RRCNN-FPN.yaml selects ``RRPN`` + ``StandardRPNHead`` +
``RotatedAnchorGenerator``; the reference's own subclass slender_det/modeling/proposal_generator/rpn.py:26-356 extends d2's ``RPN``).

detectron2's sources are absent everywhere; semantics restated: shared 3x3 conv + ReLU, 1x1 objectness
(A channels) and 1x1 anchor deltas (A*box_dim channels, anchor-major); anchors labelled by IoU + Matcher([0.3, 0.7], [0, -1, 1],
low-quality on), 256 sampled per image at <= 50 % positives; ``loss_rpn_cls`` = BCE-with-logits(sum) and ``loss_rpn_loc`` =
smooth-L1(sum, beta 0) both divided by 256 * N; proposals = per-level top-k by logit, decode, clip, drop empty, class(level)-aware
NMS 0.7, top post_nms_topk.

MI355X-first: the three head convs are one multi-level launch each (levels share weights); anchor labelling is the fused
IoU + Matcher kernel (axis-aligned or rotated) that never builds the G x A matrix; the random subsample is one launch for the batch
that also lists the drawn anchors, and both losses run over the rows of those <= 256 anchors per image only (gathered from the head
outputs, gradients scattered back): nothing of the size (N, 1.6 M anchors) is built beyond the int8 labels and the int32 matches.
"""
import os

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ...layers import functional as HF
from ...layers import nn as _nn
from ...layers.nn import ConvML, _arena_of
from ...structures import Boxes, Instances, RotatedBoxes
from ...utils.registry import Registry
from ..anchor_generator import build_anchor_generator
from ..box_regression import Box2BoxTransform, Box2BoxTransformRotated

PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")


def _ceil8(v):
    return (v + 7) // 8 * 8


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    def __init__(self, cfg, input_shape, num_anchors, box_dim):
        super().__init__()
        C = input_shape[0].channels
        self.num_anchors, self.box_dim = num_anchors, box_dim
        self.obj_pad, self.delta_pad = _ceil8(num_anchors), _ceil8(num_anchors * box_dim)
        self.conv = ConvML(C, C, 3, 1, relu=True)
        self.objectness_logits = ConvML(C, self.obj_pad, 1, 0, out_f32=True)
        self.anchor_deltas = ConvML(C, self.delta_pad, 1, 0, out_f32=True)
        self.objectness_logits.conv.ckpt_rows, self.anchor_deltas.conv.ckpt_rows = num_anchors, num_anchors * box_dim     # without the pad
        with torch.no_grad():
            for m in (self.conv, self.objectness_logits, self.anchor_deltas):
                m.conv.init_normal(0.01, 0.0)
            self.objectness_logits.conv.weight[num_anchors:].zero_()
            self.anchor_deltas.conv.weight[num_anchors * box_dim:].zero_()

    def forward(self, features):
        if RPN_HEAD_FUSED and not HF.is_f32() and features[0].is_cuda:
            for m in (self.conv, self.objectness_logits, self.anchor_deltas):
                m.conv.prepare()
            outs = _RpnHeadFn.apply(self.conv.conv.weight, self, *features)
            nl = len(features)
            return list(outs[:nl]), list(outs[nl:])
        t = self.conv(features)
        return self.objectness_logits(t), self.anchor_deltas(t)      # per level (N,H,W,obj_pad) / (N,H,W,delta_pad) fp32


# RPN_HEAD_FUSED = False: the three convolutions of the RPN head as three autograd nodes (autograd then adds the two data gradients of the hidden
# tensor per level and the hidden conv applies its ReLU mask in a pass of its own)
RPN_HEAD_FUSED = True


class _RpnHeadFn(torch.autograd.Function):
    """StandardRPNHead as ONE autograd node: hidden = relu(conv3x3(x)); objectness = conv1x1(hidden); deltas = conv1x1(hidden), every conv one
    multi-level launch (the levels share the weights).  Backward: the anchor-delta conv's data gradient adds the objectness conv's in its
    epilogue and applies hidden's ReLU mask to the sum (sod_conv2d_dgrad_ml_accum with a mask) - instead of autograd's per-level add
    (read 2, write 1) and a relu_bwd pass (read 2, write 1) over the 256-channel hidden tensor of every level (P2: 550 MB each)."""

    @staticmethod
    def forward(ctx, weight, head, *xs):
        c3, co, cd = head.conv.conv, head.objectness_logits.conv, head.anchor_deltas.conv
        hid = HF.conv2d_fwd_ml(list(xs), c3.w_bf16, c3.bias_eff, 1, c3.padding, 1, relu=True)
        obj = HF.conv2d_fwd_ml(hid, co.w_bf16, co.bias_eff, 1, 0, 1, out_f32=True)
        dlt = HF.conv2d_fwd_ml(hid, cd.w_bf16, cd.bias_eff, 1, 0, 1, out_f32=True)
        ctx.head, ctx.nl = head, len(xs)
        ctx.save_for_backward(*xs, *hid)
        ctx.park = None
        park = _nn.GradPark.current
        if park is not None and all(ctx.needs_input_grad[2:]):
            ctx.park, ctx.ptrs = park, [x.data_ptr() for x in xs]
            park.consumer_ptrs.update(ctx.ptrs)
        arena = _arena_of(c3)
        if arena is not None and c3.weight.requires_grad:
            for m in (c3, co, cd):
                arena.note_use(m.weight)
                arena.note_use(m.bias)
        return (*obj, *dlt)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        head, nl = ctx.head, ctx.nl
        c3, co, cd = head.conv.conv, head.objectness_logits.conv, head.anchor_deltas.conv
        saved = ctx.saved_tensors
        xs, hid = list(saved[:nl]), list(saved[nl:])
        arena = _arena_of(c3)
        hw = [(h.shape[1], h.shape[2]) for h in hid]

        def bf16(ts, like):      # a missing gradient (an output nobody differentiated) is a zero tensor
            out = []
            for t, ref in zip(ts, like):
                if t is None:
                    out.append(None)
                else:
                    t = t.contiguous()
                    out.append(HF.f32_to_bf16(t) if t.dtype == torch.float32 else t)
            return out

        g_obj, g_dlt = bf16(gs[:nl], hid), bf16(gs[nl:], hid)
        have_dlt = any(t is not None for t in g_dlt)
        dh, masked = None, False
        for conv, g in ((co, g_obj), (cd, g_dlt)):
            if all(t is None for t in g):
                arena.mark_ready(conv.weight); arena.mark_ready(conv.bias)
                continue
            g = [t if t is not None else torch.zeros((h.shape[0], h.shape[1], h.shape[2], conv.out_channels), dtype=h.dtype, device=h.device)
                 for t, h in zip(g, hid)]
            HF.conv2d_wgrad_ml(g, hid, arena.grad_view(conv.weight), 1, 1, 1, 0, 1)
            arena.mark_ready(conv.weight)
            HF.bias_grad_ml(g, arena.grad_view(conv.bias))
            arena.mark_ready(conv.bias)
            if dh is None and conv is co and have_dlt:
                dh = HF.conv2d_dgrad_ml(g, conv.wt_bf16, hw, 1, 0, 1)                       # plain: the delta conv's launch carries the mask
            elif dh is None:
                dh, masked = HF.conv2d_dgrad_ml(g, conv.wt_bf16, hw, 1, 0, 1, relu_masks=hid), True
            else:
                dh, masked = HF.conv2d_dgrad_ml(g, conv.wt_bf16, hw, 1, 0, 1, accums=dh, relu_masks=hid), True
        if dh is None:
            arena.mark_ready(c3.weight); arena.mark_ready(c3.bias)
            return (None, None, *([None] * nl))
        assert masked
        HF.conv2d_wgrad_ml(dh, xs, arena.grad_view(c3.weight), 3, 3, 1, c3.padding, 1)
        arena.mark_ready(c3.weight)
        HF.bias_grad_ml(dh, arena.grad_view(c3.bias))
        arena.mark_ready(c3.bias)
        dxs = [None] * nl
        park = ctx.park
        if any(ctx.needs_input_grad[2:]):
            accums = None
            if park is not None:
                park.done = True
                got = [park.parked.pop(p, None) for p in ctx.ptrs]
                if any(t is not None for t in got):      # the ROI pooler's gradients of the same tensors (levels it does not read: zeros)
                    accums = [t if t is not None else torch.zeros_like(x) for t, x in zip(got, xs)]
            dxs = HF.conv2d_dgrad_ml(dh, c3.wt_bf16, [(x.shape[1], x.shape[2]) for x in xs], 1, c3.padding, 1, accums=accums)
        elif park is not None:
            park.done = True
        return (None, None, *dxs)


class _RpnLossFn(torch.autograd.Function):
    """RPN.losses over per-level padded head outputs -> [loss_rpn_cls, loss_rpn_loc] (before loss weights).  detectron2's sums run over
    the sampled anchors (<= BATCH_SIZE_PER_IMAGE per image; everything else is label -1): the rows of those anchors are gathered from
    the head outputs (sod_rpn_gather_sampled), the two loss kernels see (N, S) / (N, S, D) tensors, and backward scatters the row
    gradients into zero tensors of the head outputs' shapes - instead of compacting, labelling and differentiating all ~1.6 M anchors
    of an image (the dense formulation moved ~3 GB per step at batch 16)."""

    @staticmethod
    def forward(ctx, rpn, idx, labels_s, deltas_s, *outs):
        nl = len(outs) // 2
        A, D = rpn.head.num_anchors, rpn.head.box_dim
        outs = [o.contiguous() for o in outs]
        logits, deltas = HF.rpn_gather_sampled(outs[:nl], outs[nl:], idx, A, D)
        N = logits.shape[0]
        norm = float(rpn.batch_size_per_image * N)
        s_cls = HF.bce_logits_loss_fwd(logits, labels_s)
        s_loc = HF.rpn_loc_loss_fwd(deltas, deltas_s, labels_s, rpn.smooth_l1_beta)
        ctx.rpn, ctx.nl, ctx.norm = rpn, nl, norm
        ctx.shapes = [tuple(o.shape) for o in outs]
        ctx.save_for_backward(logits, deltas, labels_s, deltas_s, idx)
        return torch.cat([s_cls, s_loc]) / norm

    @staticmethod
    @once_differentiable
    def backward(ctx, g2):
        rpn, nl = ctx.rpn, ctx.nl
        logits, deltas, labels_s, deltas_s, idx = ctx.saved_tensors
        g2 = g2.contiguous().float()
        dl = HF.bce_logits_loss_bwd(logits, labels_s, g2[0:1], 1.0 / ctx.norm)
        dd = HF.rpn_loc_loss_bwd(deltas, deltas_s, labels_s, rpn.smooth_l1_beta, g2[1:2], 1.0 / ctx.norm)
        g_log, g_del = HF.rpn_scatter_sampled(ctx.shapes[:nl], ctx.shapes[nl:], idx, rpn.head.num_anchors, rpn.head.box_dim, dl, dd)
        return (None, None, None, None, *g_log, *g_del)


class _RpnTargets:
    """What RPN.forward keeps of its targets (``RPN.last_targets``): the sampled labels (N, R) and, on demand, the dense matched boxes /
    deltas detectron2's formulation carries (tests and the oracle comparison read them; the training step itself only needs the sampled
    rows).  Iterates as (gt_labels, matched_gt_boxes, gt_deltas)."""

    def __init__(self, rpn, anchors, labels, matches, gt_cat, gt_off):
        self.rpn, self.anchors, self.labels, self.matches, self.gt_cat, self.gt_off = rpn, anchors, labels, matches, gt_cat, gt_off

    def dense(self):
        N, R = self.labels.shape
        D = self.anchors.shape[1]
        matched = torch.zeros((N, R, D), dtype=torch.float32, device=self.labels.device)
        for i in range(N):
            lo, hi = self.gt_off[i], self.gt_off[i + 1]
            if hi > lo:
                matched[i] = self.gt_cat[lo:hi][self.matches[i].long()]
        deltas = torch.stack([self.rpn.box2box_transform.get_deltas(self.anchors, m) for m in matched])
        return matched, deltas

    def __iter__(self):
        matched, deltas = self.dense()
        return iter((self.labels, matched, deltas))

    def __getitem__(self, i):
        return self.labels if i == 0 else self.dense()[i - 1]


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    box_dim = 4
    rotated = False

    def __init__(self, cfg, input_shape):
        super().__init__()
        r = cfg.MODEL.RPN
        self.in_features = list(r.IN_FEATURES)
        shapes = [input_shape[f] for f in self.in_features]
        self.anchor_generator = build_anchor_generator(cfg, shapes)
        assert self.anchor_generator.box_dim == self.box_dim, "ANCHOR_GENERATOR.NAME does not match the proposal generator's box type"
        na = self.anchor_generator.num_cell_anchors
        assert len(set(na)) == 1, "each level must have the same number of cell anchors"
        self.head = RPN_HEAD_REGISTRY.get(r.HEAD_NAME)(cfg, shapes, na[0], self.box_dim)
        self.box2box_transform = (Box2BoxTransformRotated if self.rotated else Box2BoxTransform)(weights=r.BBOX_REG_WEIGHTS)
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        self.batch_size_per_image, self.positive_fraction = r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION
        self.smooth_l1_beta = r.SMOOTH_L1_BETA
        self.loss_weight = {"loss_rpn_cls": r.LOSS_WEIGHT, "loss_rpn_loc": r.BBOX_REG_LOSS_WEIGHT * r.LOSS_WEIGHT}
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh = r.NMS_THRESH
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.last_targets = None

    # ------------------------------------------------------------------ helpers
    def _box_type(self):
        return RotatedBoxes if self.rotated else Boxes

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors, gt_instances):
        """Returns the sampled gt_labels (N, R) int8 in {-1, 0, 1}, the index of every anchor's best gt box (N, R) int32 and the list of
        the sampled anchors (N, S) int32 (-1 padded)."""
        R = anchors.shape[0]
        N = len(gt_instances)
        dev = anchors.device
        mlabs = torch.empty((N, R), dtype=torch.int8, device=dev)
        matches = torch.empty((N, R), dtype=torch.int32, device=dev)
        vals = torch.empty((R,), dtype=torch.float32, device=dev)          # the matched IoUs are not used beyond the labels
        for i, g in enumerate(gt_instances):
            boxes = g.gt_boxes.tensor.float().contiguous()
            HF.anchor_match(boxes, anchors, self.iou_thresholds, self.iou_labels, True, out=(vals, matches[i], mlabs[i]))
        # the random subsample of the whole batch in one launch, nothing read back (detectron2: nonzero + randperm per image); the
        # kernel also lists the drawn anchors (N, S): the losses only ever touch those rows
        labels, _, idx = HF.sample_labels_list(mlabs, self.batch_size_per_image, self.positive_fraction, 0)
        return labels, matches, idx

    @torch.no_grad()
    def sampled_targets(self, anchors, gt_instances, labels, matches, idx):
        """For the sampled anchors idx (N, S) (-1 padded): their labels (N, S) int8 and regression targets (N, S, D)."""
        S = self.batch_size_per_image
        safe = idx.clamp(min=0).long()
        labels_s = torch.where(idx >= 0, torch.gather(labels, 1, safe), torch.full_like(safe, -1, dtype=torch.int8)).contiguous()
        gts = [g.gt_boxes.tensor.float() for g in gt_instances]
        counts = [len(g) for g in gts]
        off = [0]
        for c in counts:
            off.append(off[-1] + c)
        D = anchors.shape[1]
        if off[-1] == 0:
            return idx, labels_s, torch.zeros((len(gts), S, D), dtype=torch.float32, device=anchors.device), torch.zeros((0, D), device=anchors.device), off
        gt_cat = torch.cat(gts).contiguous()
        base = torch.tensor(off[:-1], dtype=torch.int64, device=anchors.device)[:, None]
        m_s = torch.gather(matches, 1, safe).long()
        # images without boxes have no positives: their rows point at box 0 of the batch and are never read (label != 1)
        tgt = gt_cat[(base + m_s).clamp(max=off[-1] - 1).reshape(-1)]
        deltas_s = self.box2box_transform.get_deltas(anchors[safe.reshape(-1)].contiguous(), tgt.contiguous()).view(len(gts), S, D)
        return idx, labels_s, deltas_s, gt_cat, off


    @torch.no_grad()
    def predict_proposals(self, anchors_l, logits_l, deltas_l, image_sizes):
        A, D = self.head.num_anchors, self.head.box_dim
        N = logits_l[0].shape[0]
        training = self.training
        topk_scores, topk_props, level_ids = [], [], []
        batch_idx = torch.arange(N, device=logits_l[0].device)
        for lvl, (anc, lg, dl) in enumerate(zip(anchors_l, logits_l, deltas_l)):
            lg = lg[..., :A].reshape(N, -1)
            dl = dl[..., :A * D].reshape(-1, D).contiguous()
            props = self.box2box_transform.apply_deltas(dl, anc.unsqueeze(0).expand(N, -1, -1).reshape(-1, D).contiguous()).view(N, -1, D)
            num = min(self.pre_nms_topk[training], lg.shape[1])
            sc, idx = lg.topk(num, dim=1)            # d2: sort(descending) then the first num
            topk_scores.append(sc)
            topk_props.append(props[batch_idx[:, None], idx])
            level_ids.append(torch.full((num,), lvl, dtype=torch.int64, device=lg.device))
        topk_scores, topk_props, level_ids = torch.cat(topk_scores, 1), torch.cat(topk_props, 1), torch.cat(level_ids, 0)
        # find_top_rpn_proposals' per-image loop (proposal_utils.py:77-120: finite check, clip, min-size filter, batched NMS,
        # post-NMS top-k) for the whole batch on the device: one filter launch, one sort, one NMS; the only host read is the
        # final (counts, non-finite counter) tensor.
        dev = topk_scores.device
        boxes = topk_props.float().contiguous()
        scores = topk_scores.float().contiguous()
        hw = torch.tensor([[float(s[0]), float(s[1])] for s in image_sizes], dtype=torch.float32, device=dev)
        bad = HF.rpn_clip_filter(boxes, scores, hw, self.min_box_size)
        classes = level_ids.to(torch.int32)[None].expand(N, -1).contiguous()
        post = self.post_nms_topk[training]
        keep, nkeep = HF.batched_nms_topk(boxes, scores, classes, self.nms_thresh, post)
        kb = torch.gather(boxes, 1, keep[:, :, None].expand(-1, -1, D))
        ks = torch.gather(scores, 1, keep)
        host = torch.cat([nkeep, bad]).cpu().tolist()
        if host[-1] and training:
            raise FloatingPointError(f"Predicted boxes or scores contain Inf/NaN. Training has diverged. ({host[-1]} proposal entries)")
        results = []
        BoxT = self._box_type()
        for n, image_size in enumerate(image_sizes):
            res = Instances(tuple(image_size))
            res.proposal_boxes = BoxT(kb[n, : host[n]])
            res.objectness_logits = ks[n, : host[n]]
            results.append(res)
        return results

    # ------------------------------------------------------------------ forward
    def forward(self, images, features, gt_instances=None):
        feats = [features[f] for f in self.in_features]
        hw = [(f.shape[1], f.shape[2]) for f in feats]
        anchors_l = self.anchor_generator(hw, feats[0].device)
        logits_l, deltas_l = self.head(feats)
        losses = {}
        if self.training:
            anchors = torch.cat(anchors_l).contiguous()
            gt_labels, matches, idx = self.label_and_sample_anchors(anchors, gt_instances)
            idx, labels_s, deltas_s, gt_cat, gt_off = self.sampled_targets(anchors, gt_instances, gt_labels, matches, idx)
            self.last_targets = _RpnTargets(self, anchors, gt_labels, matches, gt_cat, gt_off)
            out = _RpnLossFn.apply(self, idx, labels_s, deltas_s, *logits_l, *deltas_l)
            losses = {"loss_rpn_cls": out[0] * self.loss_weight["loss_rpn_cls"], "loss_rpn_loc": out[1] * self.loss_weight["loss_rpn_loc"]}
        proposals = self.predict_proposals(anchors_l, [x.detach() for x in logits_l], [x.detach() for x in deltas_l], images.image_sizes)
        return proposals, losses


@PROPOSAL_GENERATOR_REGISTRY.register()
class RRPN(RPN):
    """d2 RRPN: rotated anchors, pairwise_iou_rotated matching, Box2BoxTransformRotated, rotated NMS."""
    box_dim = 5
    rotated = True


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPNWNM(RPN):
    """slender_det/modeling/proposal_generator/rpn.py:26-356 "Region Proposal Network With New Matcher": d2's RPN whose anchor matcher
    comes from ``cfg.MODEL.RPN.MATCHER.TYPE`` (matchers/__init__.py:8-22): "Origin" = Matcher(low-quality on), "TopK" = TopKMatcher
    (matchers/topk_matcher.py:7-85: threshold labels, then the TOPK best anchors of every gt box become positive)."""

    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        m = cfg.MODEL.RPN.MATCHER
        if m.TYPE not in ("Origin", "TopK"):
            raise AssertionError(f"Matcher Type doesn't exist! Expected one in ['Origin', 'TopK'], But got {m.TYPE}")
        self.matcher_type, self.matcher_topk = m.TYPE, m.TOPK
        if self.matcher_type == "TopK":
            assert self.iou_thresholds[0] > 0

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors, gt_instances):
        if self.matcher_type == "Origin":
            return super().label_and_sample_anchors(anchors, gt_instances)
        from ...structures import pairwise_iou

        R, N, dev = anchors.shape[0], len(gt_instances), anchors.device
        mlabs = torch.empty((N, R), dtype=torch.int8, device=dev)
        matches = torch.empty((N, R), dtype=torch.int32, device=dev)
        vals = torch.empty((R,), dtype=torch.float32, device=dev)
        for i, g in enumerate(gt_instances):
            boxes = g.gt_boxes.tensor.float().contiguous()
            HF.anchor_match(boxes, anchors, self.iou_thresholds, self.iou_labels, False, out=(vals, matches[i], mlabs[i]))
            if len(boxes):      # top-k anchors of every gt (rows of the G x R IoU matrix; G is small)
                q = pairwise_iou(Boxes(boxes), Boxes(anchors))
                mlabs[i][q.topk(k=self.matcher_topk, dim=1)[1].reshape(-1)] = 1
        labels, _, idx = HF.sample_labels_list(mlabs, self.batch_size_per_image, self.positive_fraction, 0)
        return labels, matches, idx


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
