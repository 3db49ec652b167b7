"""GPU parity of the RetinaNet path (BASELINE config 3): anchor labels bit-exact, regression targets and both losses vs the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg():
    from bench import make_cfg

    cfg = make_cfg(18)
    cfg.MODEL.META_ARCHITECTURE = "RetinaNet"
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
    return cfg


def test_retinanet_labels_losses_and_step(cuda):
    from bench import train_step
    from oracle import retinanet as orn
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.modeling.anchor_generator import grid_anchors
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg()
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    data = synthetic_batch(2, 256, 320, 11, device="cuda")
    hw = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    anchors = torch.cat(grid_anchors(hw, [8, 16, 32, 64, 128], model.anchor_sizes, model.anchor_ratios))
    assert anchors.shape[0] == sum(h * w for h, w in hw) * 9
    assert torch.equal(model.anchors_for(hw).cpu(), anchors)
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    ref_l, ref_b = orn.label_anchors(anchors, gtb, gtc, [0.4, 0.5], [0, -1, 1], 80)
    lab, deltas = model.label_anchors(model.anchors_for(hw), [d["instances"] for d in data])
    assert torch.equal(lab.cpu().long(), ref_l), "anchor labels must be bit-exact"
    pos = (ref_l >= 0) & (ref_l != 80)
    assert pos.sum() > 0
    ref_d = torch.stack([orn.get_deltas(anchors, b, (1.0, 1.0, 1.0, 1.0)) for b in ref_b])
    assert (deltas.cpu()[pos] - ref_d[pos]).abs().max() < 1e-5

    # losses: feed the HIP model's own prediction buffers to the oracle loss (isolates the loss kernels), then a full train step
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in model.in_features]
        ct, bt = model.head.run_towers(feats)
        cls_buf, box_buf, _, _ = model.head.predict(ct, bt)
    N, P = cls_buf.shape[:2]
    logits = cls_buf.cpu().view(N, P * 9, 80)
    pdel = box_buf.cpu()[..., :36].reshape(N, P * 9, 4)
    ref, norm = orn.losses(anchors, logits, pdel, ref_l, ref_b, 80, 0.25, 2.0, cfg.MODEL.RETINANET.SMOOTH_L1_LOSS_BETA, (1, 1, 1, 1), 100.0)
    got = model(data)
    for k in ref:
        a, b = float(got[k].detach()), float(ref[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), (k, a, b)
    assert abs(float(model.loss_normalizer) - norm) < 1e-3
    opt = build_optimizer(cfg, model)
    l0 = float(train_step_retina(model, opt, data))
    l1 = float(train_step_retina(model, opt, data))
    assert l0 == l0 and l1 == l1
    assert model.head.cls_score.weight.grad.abs().sum() > 0 and model.head.bbox_pred.weight.grad.abs().sum() > 0
    assert (model.head.bbox_pred.weight.grad[36:] == 0).all()


def train_step_retina(model, opt, data):
    losses = model(data)
    total = sum(losses.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    opt.step()
    return total.detach()


def _cpu(data):
    return [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]


@pytest.mark.parametrize("box_reg", ["smooth_l1", "giou"])
def test_retinanet_whole_step_losses_and_gradients_vs_oracle(cuda, box_reg):
    """The whole RetinaNet training step (image -> ResNet-FPN with P6 from res5 -> conv+ReLU subnets -> A*K / A*4 predictions ->
    anchor labels -> focal + box loss -> every parameter gradient) against oracle.model.OracleRetinaNet, the same bars as the FCOS step
    (test_gpu_model.py): both losses within north_star's 1e-3 of the bf16-storage-emulating oracle; each
    gradient no further from the fp32 oracle than 1.5x an independent bf16 emulation is (+1 %), the prediction convs within 1.5 %.
    (Distance of the losses to the plain fp32 oracle: 3x what bf16 storage costs the emulating oracle + 1e-3, and 3e-3 outright.)"""
    from oracle.model import OracleRetinaNet
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg()
    cfg.MODEL.RETINANET.BBOX_REG_LOSS_TYPE = box_reg
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 320, 384, 3, device="cuda")
    ref, grads = {}, {}
    for emu in (True, False):
        oracle = OracleRetinaNet.from_hip_model(model, emulate_bf16=emu)
        losses = oracle.losses(_cpu(data))
        names = list(oracle.trainable().keys())
        grads[emu] = dict(zip(names, torch.autograd.grad(sum(losses.values()), list(oracle.trainable().values()))))
        ref[emu] = {k: float(v) for k, v in losses.items()}
        norm = oracle.new_normalizer
    prev_det, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        got = model(data)
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    finally:
        HF.DETERMINISTIC = prev_det
    assert abs(float(model.loss_normalizer) - norm) < 1e-3
    for k, e in ref[True].items():
        a, f = float(got[k].detach()), ref[False][k]
        assert abs(a - e) <= 1e-3 * max(abs(e), 1e-3), (k, a, e)
        # the un-normalised subnets at random init put loss_cls at ~40 (saturated logits): bf16 STORAGE alone moves it by 1.2e-3 relative
        # (the emulating oracle does the same), so the distance to fp32 is bounded by what the emulation shows plus the kernel tolerance
        assert abs(a - f) <= 3.0 * abs(e - f) + 1e-3 * max(abs(f), 1e-3), (k, a, e, f)
        assert abs(a - f) <= 3e-3 * max(abs(f), 1e-3), (k, a, f)
        print(f"\n{k}: hip {a:.6f}  bf16-emulating oracle {e:.6f}  fp32 oracle {f:.6f}")
    checked, worst = 0, 0.0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = p.grad.detach().float().cpu()
        if g.dim() == 4:
            g = g.permute(0, 3, 1, 2)
        r32, remu = grads[False][name], grads[True][name]
        n = max(r32.norm().item(), 1e-12)
        d_hip, d_emu = (g - r32).norm().item() / n, (remu - r32).norm().item() / n
        worst = max(worst, d_hip)
        # GIoU's gradient is discontinuous in the deltas (min / max selections of the intersection and the enclosing box): the few-ulp
        # differences between two bf16 realisations of the tower activations move single anchors' rows by O(1) and the 36-value bias
        # gradient (a cancelling sum over ~400 positives) by several %, in the emulating oracle (4.6e-2 from fp32 on bbox_pred.weight
        # with or without rounding the gradient rows) as on the HIP path (6.7e-2).  The kernel itself is checked against float64 on
        # identical deltas below (test_retinanet_giou_backward_rows_vs_float64: 1.8e-3 with bf16 rows, 1.3e-7 with fp32 rows).
        factor = 2.5 if box_reg == "giou" and name.startswith(("head.bbox_pred", "head.bbox_subnet")) else 1.5
        assert d_hip <= factor * d_emu + 0.01, (name, d_hip, d_emu)
        if name.startswith(("head.cls_score", "head.bbox_pred")):
            print(f"{name}: hip-fp32 {d_hip:.3e}  emu-fp32 {d_emu:.3e}  hip-emu {(g - remu).norm().item() / max(remu.norm().item(), 1e-12):.3e}")
        if name.startswith("head.cls_score"):
            assert (g - remu).norm().item() / max(remu.norm().item(), 1e-12) < 1.5e-2, name
        checked += 1
    assert checked == len(grads[True])
    print(f"\nRetinaNet {box_reg}: worst relative gradient distance to the fp32 oracle {worst:.3e}")


def test_retinanet_inference_matches_oracle(cuda):
    """retina_rotated.py:296-377: decode + class-aware NMS of the product path's own predictions against the oracle contract."""
    from oracle import detection as od
    from oracle import rcnn as orc
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.modeling.anchor_generator import grid_anchors

    cfg = _cfg()
    cfg.MODEL.RETINANET.SCORE_THRESH_TEST = 0.02
    torch.manual_seed(6)
    model = build_model(cfg)
    model.eval()
    data = synthetic_batch(2, 256, 320, 31, device="cuda")
    for d in data:
        d.pop("instances")
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in model.in_features]
        hw = [tuple(f.shape[1:3]) for f in feats]
        ct, bt = model.head.run_towers(feats)
        cls_buf, box_buf, _, offs = model.head.predict(ct, bt)
        res = model.inference(hw, cls_buf, box_buf, offs, imgs.image_sizes)
    anchors_l = grid_anchors(hw, [8, 16, 32, 64, 128], model.anchor_sizes, model.anchor_ratios)
    bounds = list(offs) + [cls_buf.shape[1]]
    for i, r in enumerate(res):
        B, S, C = [], [], []
        for l, anc in enumerate(anchors_l):
            sl = slice(bounds[l], bounds[l + 1])
            p = cls_buf[i, sl].cpu().reshape(-1).sigmoid()
            deltas = box_buf[i, sl, :36].cpu().reshape(-1, 4)
            k = min(model.topk_candidates, deltas.shape[0])
            prob, idx = p.sort(descending=True)
            prob, idx = prob[:k], idx[:k]
            keep = prob > model.score_threshold
            prob, idx = prob[keep], idx[keep]
            B.append(orc.apply_deltas(deltas[idx // 80], anc[idx // 80], (1.0, 1.0, 1.0, 1.0))); S.append(prob); C.append(idx % 80)
        B, S, C = torch.cat(B), torch.cat(S), torch.cat(C)
        keep = od.batched_nms(B, S, C, model.nms_threshold)[: model.max_detections_per_image]
        assert len(r) == len(keep) and len(keep) > 0
        key = lambda b, c: sorted(zip(c.tolist(), [tuple(round(v, 1) for v in x) for x in b.tolist()]))
        assert key(r.pred_boxes.tensor.cpu(), r.pred_classes.cpu()) == key(B[keep], C[keep])
    out = model(data)
    assert len(out) == 2 and "instances" in out[0]


def test_retinanet_giou_regression_vs_oracle(cuda):
    """RETINANET.BBOX_REG_LOSS_TYPE "giou" (retina_rotated.py:236-245): fused decode + GIoU loss value and its delta gradient."""
    from oracle import retinanet as orn
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.modeling.anchor_generator import grid_anchors
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg()
    cfg.MODEL.RETINANET.BBOX_REG_LOSS_TYPE = "giou"
    torch.manual_seed(1)
    model = build_model(cfg)
    model.train()
    data = synthetic_batch(2, 256, 320, 12, device="cuda")
    hw = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    anchors = torch.cat(grid_anchors(hw, [8, 16, 32, 64, 128], model.anchor_sizes, model.anchor_ratios))
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    ref_l, ref_b = orn.label_anchors(anchors, gtb, gtc, [0.4, 0.5], [0, -1, 1], 80)
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        ct, bt = model.head.run_towers([feats[f] for f in model.in_features])
        cls_buf, box_buf, _, _ = model.head.predict(ct, bt)
    N, P = cls_buf.shape[:2]
    pdel = (box_buf.cpu()[..., :36].reshape(N, P * 9, 4) * 30).requires_grad_(True)       # x30: decoded boxes move away from the anchors
    ref, norm = orn.losses(anchors, cls_buf.cpu().view(N, P * 9, 80), pdel, ref_l, ref_b, 80, 0.25, 2.0, 0.1, (1, 1, 1, 1), 100.0,
                           box_reg_loss_type="giou")
    (gref,) = torch.autograd.grad(ref["loss_box_reg"], pdel)
    # kernel-level: same scaled deltas through the fused kernels
    buf = torch.zeros_like(box_buf)
    buf[..., :36] = box_buf[..., :36] * 30
    nrm = torch.tensor([100.0], device=cuda)
    lab = ref_l.to(torch.int32).to(cuda).contiguous()
    mb = ref_b.to(cuda).contiguous()
    sums = HF.retina_giou_loss_fwd(buf, 40, lab, anchors.to(cuda), mb, N, P * 9, 9, 80, (1, 1, 1, 1), model.scale_clamp, nrm, 0.9)
    assert abs(float(sums[0] / nrm) - float(ref["loss_box_reg"])) <= 2e-4 * float(ref["loss_box_reg"]) and abs(float(nrm) - norm) < 1e-3
    d = torch.zeros((N, P, 40), dtype=torch.bfloat16, device=cuda)
    HF.retina_giou_loss_bwd(buf, 40, lab, anchors.to(cuda), mb, N, P * 9, 9, 80, (1, 1, 1, 1), model.scale_clamp, torch.ones(1, device=cuda), nrm, d)
    got = d.float().cpu()[..., :36].reshape(N, P * 9, 4)
    pos = (ref_l >= 0) & (ref_l != 80)
    assert torch.allclose(got[pos], gref[pos], rtol=2e-2, atol=2e-5) and (got[~pos] == 0).all()
    # model-level: losses are finite, gradients flow, steps run
    opt = build_optimizer(cfg, model)
    l0 = float(train_step_retina(model, opt, data))
    l1 = float(train_step_retina(model, opt, data))
    assert l0 == l0 and l1 == l1 and model.head.bbox_pred.weight.grad[:36].abs().sum() > 0


@pytest.mark.parametrize("scale", [1.0, 30.0])
def test_retinanet_giou_backward_rows_vs_float64(cuda, scale):
    """sod_retina_giou_loss_bwd (+ _f32) on the product path's own deltas (x1 and x30) against float64 autograd of the oracle loss:
    the rows of the positives to 4e-3 of their norm with bf16 storage (2^-9 per element; measured 1.8e-3) and to 2e-6 with fp32
    storage (measured 1.3e-7 / 3.5e-7; the CPU fp32 oracle itself: 1.5e-7 / 2.7e-7), the column sums (= the bias gradient) likewise,
    non-positive rows exactly zero."""
    from oracle import retinanet as orn
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model

    cfg = _cfg()
    cfg.MODEL.RETINANET.BBOX_REG_LOSS_TYPE = "giou"
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    data = synthetic_batch(2, 320, 384, 3, device="cuda")
    hw = [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)]
    anchors = model.anchors_for(hw).cpu()
    ref_l, ref_b = orn.label_anchors(anchors, [d["instances"].gt_boxes.tensor.cpu() for d in data], [d["instances"].gt_classes.cpu() for d in data],
                                     [0.4, 0.5], [0, -1, 1], 80)
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        ct, bt = model.head.run_towers([feats[f] for f in model.in_features])
        cls_buf, box_buf, _, _ = model.head.predict(ct, bt)
    N, P = cls_buf.shape[:2]
    pd = (box_buf.cpu()[..., :36].reshape(N, P * 9, 4) * scale).double().requires_grad_(True)
    ref64, norm = orn.losses(anchors.double(), cls_buf.cpu().view(N, P * 9, 80).double(), pd, ref_l, ref_b.double(), 80, 0.25, 2.0, 0.1, (1, 1, 1, 1),
                             100.0, box_reg_loss_type="giou")
    (g64,) = torch.autograd.grad(ref64["loss_box_reg"], pd)
    buf = torch.zeros_like(box_buf)
    buf[..., :36] = box_buf[..., :36] * scale
    nrm = torch.tensor([float(norm)], device=cuda)
    lab, mb = ref_l.to(torch.int32).to(cuda).contiguous(), ref_b.to(cuda).contiguous()
    pos = (ref_l >= 0) & (ref_l != 80)
    assert int(pos.sum()) > 100
    for mode, tol in (("bf16", 4e-3), ("fp32", 2e-6)):
        prev = HF.set_precision(mode)
        try:
            d = torch.zeros((N, P, 40), dtype=HF.ACT_DTYPE, device=cuda)
            HF.retina_giou_loss_bwd(buf, 40, lab, anchors.to(cuda), mb, N, P * 9, 9, 80, (1, 1, 1, 1), model.scale_clamp, torch.ones(1, device=cuda), nrm, d)
        finally:
            HF.set_precision(prev)
        got = d.double().cpu()[..., :36].reshape(N, P * 9, 4)
        e, r = (got - g64)[pos], g64[pos]
        assert float(e.norm() / r.norm()) <= tol, (mode, float(e.norm() / r.norm()))
        assert float(e.sum(0).norm() / r.sum(0).norm()) <= tol, (mode, "column sums")
        assert (got[~pos] == 0).all() and (d[..., 36:] == 0).all()


def test_retinanet_r50_full_size_step(cuda):
    """BASELINE configs[2] at its real depth and resolution (R50-FPN, 800x1344, 201 600 anchors; batch reduced to 2): anchor labels
    bit-exact, both losses within 1e-4 of the oracle loss on the product's own predictions, and one finite training step."""
    from bench import damp_residual_branches, make_cfg
    from oracle import retinanet as orn
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(50, "retinanet")
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    damp_residual_branches(model)       # random-init R50 without a checkpoint overflows the un-normalised head (bench.py)
    data = synthetic_batch(2, 800, 1333, 77, device="cuda")
    hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    anchors = model.anchors_for(hw).cpu()
    assert anchors.shape[0] == 201600
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    ref_l, ref_b = orn.label_anchors(anchors, gtb, gtc, [0.4, 0.5], [0, -1, 1], 80)
    lab, _ = model.label_anchors(model.anchors_for(hw), [d["instances"] for d in data])
    assert torch.equal(lab.cpu().long(), ref_l)
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        ct, bt = model.head.run_towers([feats[f] for f in model.in_features])
        cls_buf, box_buf, _, _ = model.head.predict(ct, bt)
    N, P = cls_buf.shape[:2]
    ref, _ = orn.losses(anchors, cls_buf.cpu().view(N, P * 9, 80), box_buf.cpu()[..., :36].reshape(N, P * 9, 4), ref_l, ref_b, 80, 0.25, 2.0,
                        cfg.MODEL.RETINANET.SMOOTH_L1_LOSS_BETA, (1, 1, 1, 1), 100.0)
    got = model(data)
    for k in ref:
        a, b = float(got[k].detach()), float(ref[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), (k, a, b)
    opt = build_optimizer(cfg, model)
    l0 = float(train_step_retina(model, opt, data))
    assert l0 == l0 and abs(l0) < 1e6


def test_focal_loss_forward_with_gradient_in_one_pass(cuda):
    """sod_sigmoid_focal_loss_fwd_grad (the loss sum and the un-scaled gradient from one pass over the logits) against the two-pass entry
    points at op level, and in the RetinaNet step: same losses bit for bit, every gradient within bf16 rounding of the two-pass step
    (the scalar g / normaliser moves from the gradient rows into the consumers: scaled weights in the data gradient, a per-channel factor
    in the weight gradient, a scalar in the bias gradient)."""
    from bench import make_cfg
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.modeling.meta_arch import retinanet as RN
    from slenderobjdet_amd.solver import build_optimizer

    g = torch.Generator().manual_seed(5)
    M, K = 3000, 80
    logits = (torch.randn(M, K, generator=g) * 3).to(cuda)
    labels = torch.randint(-1, K + 1, (M,), generator=g, dtype=torch.int32).to(cuda)          # -1 ignored, K background
    s_ref, _ = HF.focal_loss_fwd(logits, labels, None, 0.25, 2.0)
    g_ref = HF.focal_loss_bwd(logits, labels, None, 0.25, 2.0, out_bf16=True)
    s_got, g_got = HF.focal_loss_fwd_grad(logits, labels, 0.25, 2.0)
    assert abs(float(s_got) - float(s_ref)) <= 1e-6 * abs(float(s_ref))
    assert torch.equal(g_got, g_ref)                                                          # same arithmetic, scale 1
    s2, g2 = HF.focal_loss_fwd_grad(logits, labels, 0.25, 2.0, ld_out=K + 8)                   # padded gradient rows: zeros in the pad
    assert torch.equal(g2[:, :K], g_ref) and float(g2[:, K:].float().abs().max()) == 0.0

    cfg = make_cfg(18, "retinanet")
    torch.manual_seed(3)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 21, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    keep = RN.FOCAL_FUSED
    try:
        norm0 = model.loss_normalizer.clone()

        def step(on):
            RN.FOCAL_FUSED = on
            with torch.no_grad():
                model.loss_normalizer.copy_(norm0)
            opt.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            torch.cuda.synchronize()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        ref_l, ref_g = step(False)
        got_l, got_g = step(True)
        again_l, again_g = step(True)
        assert got_l == ref_l and torch.equal(got_g, again_g)
        worst = 0.0
        for name, off, n in model.arena.names:
            a, b = got_g[off:off + n], ref_g[off:off + n]
            if "bbox" in name:
                assert torch.equal(a, b), name                      # the regression branch does not see the change
                continue
            d = (a - b).norm().item() / max(b.norm().item(), 1e-12)
            worst = max(worst, d)
            assert d <= 2e-2, (name, d)
        assert worst > 0.0
    finally:
        RN.FOCAL_FUSED = keep
        HF.DETERMINISTIC = prev


def test_relu_chain_in_towers_is_bit_identical(cuda):
    """Consecutive [conv3x3 -> ReLU] units of the RetinaNet towers: the consumer's data gradient applies the producer's ReLU mask in its
    epilogue (layers/nn.py _ReluToken, sod_conv2d_dgrad_ml_mask) instead of one relu_bwd launch per level and unit.  Masking before
    or after the bf16 rounding is the same value: losses and every gradient are bit-identical (deterministic mode)."""
    from bench import make_cfg
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.layers import nn as HN
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(18, "retinanet")
    torch.manual_seed(3)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 21, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    prev_chain = HN.RELU_CHAIN
    calls = {"masked": 0, "relu_bwd": 0}
    o1, o2 = HF.conv2d_dgrad_ml, HF.relu_bwd

    def c1(*a, **k):
        calls["masked"] += int(k.get("relu_masks") is not None)
        return o1(*a, **k)

    def c2(*a, **k):
        calls["relu_bwd"] += 1
        return o2(*a, **k)

    try:
        norm0 = model.loss_normalizer.clone()

        def step(on):
            HN.RELU_CHAIN = on
            with torch.no_grad():
                model.loss_normalizer.copy_(norm0)          # the EMA normaliser advances in every training forward
            opt.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        ref_l, ref_g = step(False)
        HF.conv2d_dgrad_ml, HF.relu_bwd = c1, c2
        got_l, got_g = step(True)
        HF.conv2d_dgrad_ml, HF.relu_bwd = o1, o2
        n = cfg.MODEL.RETINANET.NUM_CONVS
        assert calls["masked"] == 2 * (n - 1), calls           # every unit but the first of each tower masks its producer's gradient
        assert got_l == ref_l
        assert torch.equal(got_g, ref_g)
    finally:
        HF.conv2d_dgrad_ml, HF.relu_bwd = o1, o2
        HN.RELU_CHAIN = prev_chain
        HF.DETERMINISTIC = prev


def test_retinanet_step_with_class_count_not_a_multiple_of_four(cuda):
    """Round-4 advisor finding: the one-pass focal kernel (sod_sigmoid_focal_loss_fwd_grad) is vectorised over 4 classes and was called
    unconditionally, so NUM_CLASSES = 6 (K % 4 != 0) raised in training.  Such layouts take the two-pass entry points again: the step
    runs, and equals the step with the fused path switched off bit for bit (it IS that path)."""
    from bench import make_cfg
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.modeling.meta_arch import retinanet as RN

    cfg = make_cfg(18, "retinanet")
    cfg.MODEL.RETINANET.NUM_CLASSES = 6
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[x, x * 2 ** 0.5] for x in [32, 64, 128, 256, 512]]       # 2 sizes x 2 ratios = 4 anchors: A * K = 24
    cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 2.0]]
    torch.manual_seed(3)
    model = build_model(cfg)
    model.train()
    data = synthetic_batch(2, 256, 320, 21, num_classes=6, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    keep = RN.FOCAL_FUSED
    try:
        norm0 = model.loss_normalizer.clone()

        def step(on):
            RN.FOCAL_FUSED = on
            with torch.no_grad():
                model.loss_normalizer.copy_(norm0)
            model.arena.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            torch.cuda.synchronize()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        got_l, got_g = step(True)
        ref_l, ref_g = step(False)
        assert all(v == v and abs(v) < 1e4 for v in got_l.values()), got_l
        assert got_l == ref_l and torch.equal(got_g, ref_g)
        assert float(got_g.abs().sum()) > 0
    finally:
        RN.FOCAL_FUSED = keep
        HF.DETERMINISTIC = prev
