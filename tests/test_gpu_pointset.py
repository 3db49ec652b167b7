"""GPU parity of AblationMetaArch + PointSetHead against oracle/pointset.py, which is pinned to the reference's own
Python by tests/test_oracle_pointset.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

MODES = ["Empty", "Supervised Offset", "Unsupervised Offset", "Split Unsup Offset"]


def _cfg(mode, res_refine=True):
    from bench import make_cfg

    cfg = make_cfg(18)
    cfg.MODEL.META_ARCHITECTURE = "AblationMetaArch"
    cfg.MODEL.BACKBONE.NAME = "build_retinanet_resnet_fpn_backbone"
    cfg.MODEL.META_ARCH.NAME = "PointSetHead"
    cfg.MODEL.META_ARCH.FEAT_ADAPTION = mode
    cfg.MODEL.META_ARCH.RES_REFINE = res_refine
    return cfg


@pytest.mark.parametrize("mode", MODES)
def test_pointset_head_vs_oracle(cuda, mode):
    from oracle import pointset as ops
    from oracle.reppoints import center_grid
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg(mode, res_refine=mode != "Unsupervised Offset")
    torch.manual_seed(4)
    model = build_model(cfg)
    model.train()
    head = model.head
    with torch.no_grad():      # beyond-init scale so the boxes are not degenerate (same trick as the golden generator)
        for m in (head.loc_init_out.conv, head.offsets_refine):
            m.weight.mul_(6.0)
    model.arena.bump()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 192, 256, 12, device="cuda")
    got = model(data)
    assert set(got) == {"loss_cls", "loss_pts_init", "loss_pts_refine"}
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in head.in_features]
        oi, cf, rf = head.run_head(feats)
        logits, rdelta, init_boxes, _, refine_boxes, _, (hw, offs, X) = head.predict(oi, cf, rf)
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    centers, st = center_grid(hw, head.fpn_strides)
    N = 2
    pts_i = torch.cat([o[..., :18].reshape(N, -1, 18) for o in oi], 1).cpu()
    pts_r = torch.cat([(r + (o if head.res_refine else 0))[..., :18].reshape(N, -1, 18) for r, o in zip(rdelta, oi)], 1).cpu()
    # 1. the box decoding kernel agrees with pts_to_bbox(pts * stride + centre)
    rep, s1 = centers.repeat(1, 9), st.reshape(-1, 1)
    for i in range(N):
        assert torch.allclose(init_boxes[i].cpu(), ops.pts_to_bbox(pts_i[i] * s1 + rep), rtol=1e-5, atol=1e-3)
        assert torch.allclose(refine_boxes[i].cpu(), ops.pts_to_bbox(pts_r[i] * s1 + rep), rtol=1e-5, atol=1e-3)
    # 2. targets (bit-exact) and losses from the product path's own predictions
    obj, init_lab, cls, refine_lab = (t.cpu() for t in head.last_targets)
    for i in range(N):
        ib, il = ops.point_targets(centers, st, gtb[i], gtc[i], 80, 4)
        assert torch.equal(obj[i] > 0, il != 80) and torch.equal(init_lab[i], ib)
        rb, rl = ops.bbox_targets(init_boxes[i].cpu(), gtb[i], gtc[i], 80)
        assert torch.equal(cls[i].long(), rl) and torch.equal(refine_lab[i][rl != 80], rb[rl != 80])
    ref = ops.losses(centers, st, logits.cpu(), pts_i, pts_r, gtb, gtc, 80)
    for k in ref:
        a, b = float(got[k].detach()), float(ref[k])
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (k, a, b)
    # 3. the head's forward against the oracle head (bf16 storage emulated) on the same FPN features
    o = ops.OraclePointSetHead.from_hip_head(head, emulate_bf16=True)
    with torch.no_grad():
        c2, i2, r2, _ = o.forward([f.float().cpu().permute(0, 3, 1, 2) for f in feats])
    for name, a, b in (("logits", logits.cpu(), c2), ("pts_init", pts_i, i2), ("pts_refine", pts_r, r2)):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
        assert err < 3e-2, (name, err)
    # 4. gradients reach every trainable head parameter, padding rows stay zero, and a few steps reduce the loss
    total = sum(got.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    for name, p in head.named_parameters():
        assert torch.isfinite(p.grad).all(), name
        if name.endswith("weight") and p.dim() == 4:
            assert p.grad.abs().sum() > 0, name
    for m in (head.loc_init_out.conv, head.offsets_refine):
        assert (m.weight.grad[18:] == 0).all()
    for g in opt.param_groups:
        g["lr"] = 0.002
    ls = []
    for _ in range(6):
        losses = model(data)
        t = sum(losses.values())
        opt.zero_grad()
        model.arena.begin_backward(); t.backward(); model.arena.finish_backward()
        opt.step()
        ls.append(float(t.detach()))
    assert all(v == v for v in ls) and ls[-1] < ls[0], ls


def test_pointset_inference_matches_oracle(cuda):
    from oracle import pointset as ops
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model

    cfg = _cfg("Supervised Offset")
    cfg.MODEL.META_ARCH.SCORE_THRESH_TEST = 0.005
    torch.manual_seed(5)
    model = build_model(cfg)
    model.eval()
    head = model.head
    data = synthetic_batch(2, 192, 256, 13, device="cuda")
    for d in data:
        d.pop("instances")
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        oi, cf, rf = head.run_head([feats[f] for f in head.in_features])
        logits, _, _, _, refine_boxes, _, geo = head.predict(oi, cf, rf)
        res = head.inference(logits, refine_boxes, geo, imgs.image_sizes)
    bounds = list(geo[1]) + [geo[2]]
    for i, r in enumerate(res):
        B, S, C = ops.inference_single_image(logits[i].cpu(), refine_boxes[i].cpu(), bounds, imgs.image_sizes[i], head.topk_candidates,
                                             head.score_threshold, head.nms_threshold, head.max_detections_per_image)
        assert len(r) == len(B) and len(B) > 0
        key = lambda b, c: sorted(zip(c.tolist(), [tuple(round(v, 2) for v in x) for x in b.tolist()]))
        assert key(r.pred_boxes.tensor.cpu(), r.pred_classes.cpu()) == key(B, C)
    out = model(data)
    assert len(out) == 2 and "instances" in out[0]


@pytest.mark.parametrize("method", ["partial_minmax", "moment"])
def test_pointset_transform_methods_vs_oracle(cuda, method):
    """TRANSFORM_METHOD "partial_minmax" / "moment" (pointset_head.py:322-343): boxes, losses, and the gradient of the box decoding
    (d loss / d points, d loss / d moment_transfer) against oracle/pointset.py, which the reference's own Python pins."""
    from oracle import pointset as ops
    from oracle.reppoints import center_grid
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg("Empty", res_refine=method == "partial_minmax")
    cfg.MODEL.META_ARCH.TRANSFORM_METHOD = method
    torch.manual_seed(6)
    model = build_model(cfg)
    model.train()
    head = model.head
    with torch.no_grad():
        for m in (head.loc_init_out.conv, head.offsets_refine):
            m.weight.mul_(6.0)
        if method == "moment":
            head.moment_transfer.copy_(torch.tensor([0.375, -0.25]))
    model.arena.bump()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 192, 256, 12, device="cuda")
    got = model(data)
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in head.in_features]
        oi, cf, rf = head.run_head(feats)
        logits, rdelta, init_boxes, _, refine_boxes, _, (hw, offs, X) = head.predict(oi, cf, rf)
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    centers, st = center_grid(hw, head.fpn_strides)
    N = 2
    pts_i = torch.cat([o[..., :18].reshape(N, -1, 18) for o in oi], 1).cpu()
    pts_r = torch.cat([(r + (o if head.res_refine else 0))[..., :18].reshape(N, -1, 18) for r, o in zip(rdelta, oi)], 1).cpu()
    mt = head.moment_transfer.detach().cpu().clone().requires_grad_(True) if method == "moment" else None
    rep, s1 = centers.repeat(1, 9), st.reshape(-1, 1)
    for i in range(N):
        assert torch.allclose(init_boxes[i].cpu(), ops.pts_to_bbox(pts_i[i] * s1 + rep, method, mt).detach(), rtol=1e-5, atol=2e-3)
        assert torch.allclose(refine_boxes[i].cpu(), ops.pts_to_bbox(pts_r[i] * s1 + rep, method, mt).detach(), rtol=1e-5, atol=2e-3)
    ref = ops.losses(centers, st, logits.cpu(), pts_i, pts_r, gtb, gtc, 80, method=method, moment_transfer=mt)
    for k in ref:
        a, b = float(got[k].detach()), float(ref[k].detach())
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (k, a, b)
    # the decoding kernels' backward on one level against autograd through the oracle transform
    l = 1
    h, w = hw[l]
    s, ps = head.strides[l], head.point_scales[l]
    g = torch.Generator().manual_seed(9)
    dbox = torch.randn(N, h * w, 4, generator=g)
    pts = oi[l].detach().float().cpu()[..., :18].reshape(N, h * w, 18).clone().requires_grad_(True)
    c_l, st_l = center_grid([hw[l]], [s])
    boxes = torch.stack([ops.pts_to_bbox(pts[i] * st_l.reshape(-1, 1) * (ps / s) + c_l.repeat(1, 9), method, mt) for i in range(N)])
    wants = [pts] + ([mt] if mt is not None else [])
    grads = torch.autograd.grad((boxes * dbox).sum(), wants)
    dboxes = dbox.to(cuda).contiguous()
    if method == "moment":
        dmt = torch.zeros(2, device=cuda)
        d32, _ = HF.points2bbox_moment_bwd(dboxes, h * w * 4, oi[l].detach(), None, s, ps, 9, head.moment_transfer.detach(), head.moment_mul, dmt)
        assert torch.allclose(dmt.cpu(), grads[1], rtol=2e-4, atol=1e-6), (dmt.cpu(), grads[1])
    else:
        arg = torch.empty((N, h * w), dtype=torch.int32, device=cuda)
        tmp = torch.empty((N, h * w, 4), dtype=torch.float32, device=cuda)
        HF.points2bbox_fwd(oi[l].detach(), None, s, ps, 4, tmp, h * w * 4, arg, h * w)
        d32, _ = HF.points2bbox_bwd(dboxes, h * w * 4, arg, h * w, tuple(oi[l].shape), ps, 4)
    got_d = d32.reshape(N, h * w, -1)[..., :18].cpu()
    # fp32: gradients are O(1-10); the moment backward divides by (n - 1) * std, so allow a few 1e-4 of absolute rounding
    assert torch.allclose(got_d, grads[0], rtol=2e-4, atol=1e-5 if method != "moment" else 3e-4), float((got_d - grads[0]).abs().max())
    assert (d32.reshape(N, h * w, -1)[..., 18:] == 0).all()
    # a training step: every head parameter (incl. moment_transfer) receives a finite gradient and the loss goes down
    total = sum(got.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    for name, p in head.named_parameters():
        assert torch.isfinite(p.grad).all(), name
    if method == "moment":
        assert head.moment_transfer.grad.abs().sum() > 0
    for gpar in opt.param_groups:
        gpar["lr"] = 0.002
    ls = []
    for _ in range(6):
        t = sum(model(data).values())
        opt.zero_grad()
        model.arena.begin_backward(); t.backward(); model.arena.finish_backward()
        opt.step()
        ls.append(float(t.detach()))
    assert all(v == v for v in ls) and ls[-1] < ls[0], ls
