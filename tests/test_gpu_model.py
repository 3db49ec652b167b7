"""End-to-end GPU parity: the HIP FCOS training step vs the CPU oracle with bf16 storage emulation, plus
size-independent properties at the full BASELINE size (800x1344)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(depth, seed=0):
    from bench import make_cfg
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(depth)
    torch.manual_seed(seed)
    model = build_model(cfg)
    model.train()
    return cfg, model, build_optimizer(cfg, model)


def _cpu(data):
    return [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]


@pytest.mark.parametrize("depth", [18, 50])
def test_fcos_r18_losses_and_gradients_vs_oracle(cuda, depth):
    """BASELINE configs[0] shape family (FCOS R18-FPN, 2 synthetic images).
    Losses: within 1e-3 relative of the bf16-storage-emulating oracle (north_star tolerance).
    Gradients: activation gradients are STORED in bf16 on the product path; at random init that storage noise alone moves
    deep-layer weight gradients by 5-20 % (measured: fp32 oracle vs bf16-emulating oracle).  The meaningful bar is
    therefore relative: the HIP gradient may be no further from the fp32 oracle than 1.5x the distance of an independent
    bf16 emulation (+1 % floor), and the last layers (no accumulated storage noise) must agree to 1 %."""
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch

    cfg, model, opt = _build(depth)     # 50 exercises the fused bottleneck-stage backward (modeling/backbone/resnet.py)
    data = synthetic_batch(2, 320, 384, 3, device="cuda")
    grads = {}
    for emu in (True, False):
        oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
        ref = oracle.losses(_cpu(data))
        names = list(oracle.trainable().keys())
        grads[emu] = dict(zip(names, torch.autograd.grad(sum(ref.values()), list(oracle.trainable().values()))))
        if emu:
            ref_emu = {k: float(v) for k, v in ref.items()}
        else:
            ref_f32 = {k: float(v) for k, v in ref.items()}
    # Deterministic mode (fixed summation order everywhere, no float atomics): the deep-layer gradient bar below sits at 0.99 of its
    # bound for one res3 weight (measured 0.988 .. 0.997 over the atomic paths, tools/gradcheck_r50.py), so run-to-run noise of the
    # default paths must not decide it.  The atomic paths are compared with the deterministic ones by the A/B tests of this file and
    # of test_gpu_conv.py.
    from slenderobjdet_amd.layers import functional as HF
    prev_det, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        got = model(data)
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    finally:
        HF.DETERMINISTIC = prev_det
    for k, b in ref_emu.items():
        a = float(got[k].detach())
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (k, a, b)
    # distance to the plain fp32 oracle (= the reference's CPU path restated): north_star's 1e-3 relative per loss (measured
    # 1e-4 .. 6e-4: what bf16 storage of weights and activations costs in ONE forward pass) - and never more than 3x the distance
    # of the bf16-emulating oracle from fp32 plus the 1e-3 kernel tolerance above
    for k, f in ref_f32.items():
        a, e = float(got[k].detach()), ref_emu[k]
        assert abs(a - f) <= 1e-3 * max(abs(f), 1e-3), (k, a, f)
        assert abs(a - f) <= 3.0 * abs(e - f) + 1e-3 * max(abs(f), 1e-3), (k, a, e, f)
    checked = 0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = p.grad.detach().float().cpu()
        if g.dim() == 4:
            g = g.permute(0, 3, 1, 2)
        r32, remu = grads[False][name], grads[True][name]
        n = max(r32.norm().item(), 1e-12)
        d_hip, d_emu = (g - r32).norm().item() / n, (remu - r32).norm().item() / n
        assert d_hip <= 1.5 * d_emu + 0.01, (name, d_hip, d_emu)
        if name.startswith(("head.cls_pred", "head.box_pred", "head.scales")):   # scales: five ~5e-5 values, relatively noisier
            assert (g - remu).norm().item() / max(remu.norm().item(), 1e-12) < (3e-2 if "scales" in name else 1.5e-2), name
        checked += 1
    assert checked == len(grads[True])


def test_training_trajectory_matches_oracle(cuda):
    """Ten SGD steps from identical init/data: total-loss trajectory stays within 1e-2 of the bf16-emulating oracle
    (the 100-iteration < 1e-3 target of north_star is for fp32 references; bf16 rounding differs in accumulation order)."""
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch

    from bench import train_step

    cfg, model, opt = _build(18, seed=1)
    for g in opt.param_groups:      # a small step keeps the comparison out of the chaotic regime of a batch-2, no-warm-up run
        g["lr"] = 0.002
    lr = 0.002
    data = synthetic_batch(2, 256, 256, 9, device="cuda")
    oracle = OracleFCOS.from_hip_model(model, emulate_bf16=True)
    cpu_data, state = _cpu(data), {}
    for it in range(10):
        ref = oracle.losses(cpu_data)
        rt = sum(ref.values())
        grads = dict(zip(oracle.trainable().keys(), torch.autograd.grad(rt, list(oracle.trainable().values()))))
        oracle.sgd_step(grads, state, lr, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM)
        got = float(train_step(model, opt, data))
        assert abs(got - float(rt)) <= 1e-2 * abs(float(rt)), (it, got, float(rt))


def test_full_size_assignment_matches_oracle(cuda):
    """BASELINE full size (800x1344, L = 22400): labels bit-exact and positives count equal to the oracle."""
    from oracle import fcos_targets as ot
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF

    data = synthetic_batch(2, 800, 1333, 1234)
    hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    strides = [8, 16, 32, 64, 128]
    boxes = [d["instances"].gt_boxes.tensor for d in data]
    classes = [d["instances"].gt_classes for d in data]
    ref_l, ref_r = ot.targets_for_batch(hw, strides, boxes, classes, 1.5, 80)
    offs = torch.tensor([0] + [len(b) for b in boxes]).cumsum(0).int()
    lab, reg, ctr, stats = HF.fcos_assign(torch.cat(boxes).to(cuda), torch.cat(classes).int().to(cuda), offs.to(cuda), 2, hw, strides,
                                          ot.SIZES_OF_INTEREST, 1.5, 80)
    assert lab.shape == (2, 22400)
    assert torch.equal(lab.cpu().long(), ref_l) and torch.equal(reg.cpu(), ref_r)


def test_full_size_conv_linearity_and_adjointness(cuda):
    """Size-independent properties on the real head shape (N=2, 100x168x256 -> 256, 3x3): linearity of fwd in x, and
    <dy, conv(x)> == <dgrad(dy), x> == <wgrad(dy, x), w> (adjointness), to bf16-output tolerance."""
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = 2, 100, 168, 256, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.randn(N, H, W, C, device=cuda, generator=g).bfloat16()
    x2 = torch.randn(N, H, W, C, device=cuda, generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, device=cuda, generator=g) * 0.02)
    wk, wt = HF.weight_prep(w)
    y1 = HF.conv2d_fwd(x1, wk, None, stride=1, pad=1, out_f32=True)
    y2 = HF.conv2d_fwd(x2, wk, None, stride=1, pad=1, out_f32=True)
    y12 = HF.conv2d_fwd(HF.add_bf16(x1, x2), wk, None, stride=1, pad=1, out_f32=True)
    xs = (x1.float() + x2.float())
    exact_sum = (xs.bfloat16().float() == xs).float().mean().item()   # where the bf16 add was exact
    err = (y12 - (y1 + y2)).abs().max().item() / y12.abs().max().item()
    assert err < 2e-2, (err, exact_sum)   # the bf16 add of x1+x2 is itself rounded for ~half of the elements
    dy = torch.randn(N, H, W, K, device=cuda, generator=g).bfloat16()
    lhs = (dy.float() * y1).sum().item()
    dx = HF.conv2d_dgrad(dy, wt, (H, W), 1, 1, 1)
    mid = (dx.float() * x1.float()).sum().item()
    dw = torch.zeros(K, 3, 3, C, device=cuda)
    HF.conv2d_wgrad(dy, x1, dw, 3, 3, 1, 1, 1)
    rhs = (dw * wk.float()).sum().item()
    scale = max(abs(lhs), 1.0)
    assert abs(lhs - mid) / scale < 5e-3 and abs(lhs - rhs) / scale < 1e-3, (lhs, mid, rhs)


@pytest.mark.parametrize("v2", [True, False])
def test_fcos_with_dcn_tower_trains(cuda, v2):
    """configs/fcos/*dcn*.yaml path: MODEL.FCOS.USE_DCN_IN_TOWER (DFConv2d as the last tower conv) runs fwd+bwd+step."""
    from bench import make_cfg, train_step
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(18)
    cfg.MODEL.FCOS.USE_DCN_IN_TOWER = True
    cfg.MODEL.FCOS.USE_DCN_V2 = v2
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 256, 5, device="cuda")
    l0 = float(train_step(model, opt, data))
    l1 = float(train_step(model, opt, data))
    assert l0 == l0 and l1 == l1 and l1 < l0 * 1.5
    dcn = model.head.cls_tower[-1].conv
    assert dcn.conv.weight.grad.abs().sum() > 0 and dcn.offset.weight.grad.abs().sum() > 0


@pytest.mark.parametrize("modulated", [True])      # (DEFORM_MODULATED false: tests/test_gpu_f32_mode.py, backbone case; the oracle's DeformConv
def test_fcos_r50_dcn_backbone_step_vs_oracle(cuda, modulated):      #  is a Python loop - each variant here costs the GPU box half a minute)
    """configs/fcos/fcos_R_50_FPN_2x_dcnv2.yaml semantics (MODEL.RESNETS.DEFORM_ON_PER_STAGE [F, T, T, T], DEFORM_MODULATED, and
    USE_DCN_IN_TOWER): one training step of FCOS R50 with detectron2's DeformBottleneckBlock in res3..res5 against the oracle
    (oracle/model.py with oracle/deform_conv.py, the restated op pinned by the reference's known-answer test).  The offset convs are
    zero-initialised by detectron2 - a trained state is imitated by small random offset weights so that real sub-pixel sampling, the
    mask path and the offset gradients are exercised.  Losses: 2e-3 of the bf16-emulating oracle, 5e-3 of the fp32 oracle (thirteen
    sampled convolutions whose gathered columns are rounded to bf16 sit between the input and the losses: measured 3.4e-3); the DCN
    blocks' gradients (offset conv, sampled conv) no further from fp32 than 2x an independent bf16 emulation + 2 %."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(50)
    cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE = [False, True, True, True]
    cfg.MODEL.RESNETS.DEFORM_MODULATED = modulated
    cfg.MODEL.FCOS.USE_DCN_IN_TOWER = True
    cfg.MODEL.FCOS.USE_DCN_V2 = modulated
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    data = synthetic_batch(2, 192, 256, 3, device="cuda")
    # random offset weights, calibrated block by block to offsets of ~0.5 px: without a checkpoint the ResNet activations grow by orders
    # of magnitude from stage to stage, a fixed weight scale gives offsets of tens of pixels in res4 / res5, and sampling that far away
    # turns bf16 rounding of the offset conv's input into O(1) feature differences (measured: 10 % in res4, 36 % in res5) - a property
    # of that input, not of either implementation
    g = torch.Generator(device="cuda").manual_seed(5)
    hooks = []

    def calibrate(mod, inp, out):
        s = 0.5 / float(out[..., :mod.ckpt_rows].float().std())
        mod.weight.mul_(s)
        return out * s

    with torch.no_grad():
        for name, mod in model.named_modules():
            if name.endswith("conv2_offset"):
                mod.weight[:mod.ckpt_rows].copy_(torch.randn(mod.weight[:mod.ckpt_rows].shape, device="cuda", generator=g) * (9 * mod.in_channels) ** -0.5)
                hooks.append(mod.register_forward_hook(calibrate))
        model.arena.bump()
        model.backbone.bottom_up(model.preprocess_image(data).tensor)
        for h in hooks:
            h.remove()
        model.arena.bump()
    opt = build_optimizer(cfg, model)
    refs = {}
    for emu in (True, False):
        oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
        if emu:      # the backbone features themselves: within bf16 storage noise of the emulating oracle
            with torch.no_grad():
                bu = oracle._bottom_up(oracle.preprocess(_cpu(data)))
                feats = model.backbone.bottom_up(model.preprocess_image(data).tensor)
            for k in ("res3", "res4", "res5"):
                a, b = feats[k].float().cpu().permute(0, 3, 1, 2), bu[k]
                assert float((a - b).norm() / b.norm()) <= 3e-2, (k, float((a - b).norm() / b.norm()))
        losses = oracle.losses(_cpu(data))
        names = list(oracle.trainable().keys())
        refs[emu] = ({k: float(v.detach()) for k, v in losses.items()},
                     dict(zip(names, torch.autograd.grad(sum(losses.values()), list(oracle.trainable().values())))))
    prev_det, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        got = model(data)
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    finally:
        HF.DETERMINISTIC = prev_det
    for k in refs[True][0]:
        a, e, f = float(got[k].detach()), refs[True][0][k], refs[False][0][k]
        assert abs(a - e) <= 2e-3 * max(abs(e), 1e-3), (k, a, e)
        assert abs(a - f) <= 5e-3 * max(abs(f), 1e-3), (k, a, f)
    checked = 0
    for name, p in model.named_parameters():
        if not p.requires_grad or not (".conv2_offset." in name or ".conv2." in name or ".conv.offset." in name or ".conv.conv." in name):
            continue
        gq = p.grad.detach().float().cpu()
        if gq.dim() == 4:
            gq = gq.permute(0, 3, 1, 2)
        r32, remu = refs[False][1][name], refs[True][1][name]
        if gq.shape != r32.shape:          # padded offset rows: the pad stays inert
            assert float(gq[r32.shape[0]:].abs().sum()) == 0, name
            gq = gq[:r32.shape[0]]
        n = max(r32.norm().item(), 1e-12)
        d_hip, d_emu = (gq - r32).norm().item() / n, (remu - r32).norm().item() / n
        assert d_hip <= 2.0 * d_emu + 0.02, (name, d_hip, d_emu)
        checked += 1
    assert checked >= 3 * (4 + 6 + 3) + 4          # offset weight / bias + sampled conv of 13 blocks, two tower units


def test_fcos_resnext_step_vs_oracle(cuda):
    """ResNeXt bottlenecks (MODEL.RESNETS.NUM_GROUPS 32 / WIDTH_PER_GROUP 8 / STRIDE_IN_1X1 false: the X101 ablation configs' backbone
    family, here at depth 50): one FCOS training step against the oracle (oracle/model.py with F.conv2d(groups=32)).  Losses 1e-3 of the
    bf16-emulating oracle; the grouped 3x3 weight gradients (the diagonal blocks of the dense scratch gradient, in the reference's
    (K, C / 32, 3, 3) shape) no further from fp32 than 1.5x the bf16 emulation + 1 %."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.layers.nn import HipGroupedConv2d
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(50)
    cfg.MODEL.RESNETS.NUM_GROUPS, cfg.MODEL.RESNETS.WIDTH_PER_GROUP, cfg.MODEL.RESNETS.STRIDE_IN_1X1 = 32, 8, False
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    assert isinstance(model.backbone.bottom_up.res4[2].conv2, HipGroupedConv2d)
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 3, device="cuda")
    refs = {}
    for emu in (True, False):
        oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
        losses = oracle.losses(_cpu(data))
        names = list(oracle.trainable().keys())
        refs[emu] = ({k: float(v.detach()) for k, v in losses.items()},
                     dict(zip(names, torch.autograd.grad(sum(losses.values()), list(oracle.trainable().values())))))
    prev_det, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        got = model(data)
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    finally:
        HF.DETERMINISTIC = prev_det
    for k, e in refs[True][0].items():
        a = float(got[k].detach())
        assert abs(a - e) <= 1e-3 * max(abs(e), 1e-3), (k, a, e)
    checked = 0
    for name, p in model.named_parameters():
        if not p.requires_grad or not name.endswith("conv2.weight"):
            continue
        gq = p.grad.detach().float().cpu().permute(0, 3, 1, 2)
        r32, remu = refs[False][1][name], refs[True][1][name]
        assert gq.shape == r32.shape and gq.shape[1] * 32 == gq.shape[0]
        n = max(r32.norm().item(), 1e-12)
        d_hip, d_emu = (gq - r32).norm().item() / n, (remu - r32).norm().item() / n
        assert d_hip <= 1.5 * d_emu + 0.01, (name, d_hip, d_emu)
        checked += 1
    assert checked == 4 + 6 + 3


def test_fcos_inference_boxes_scores_and_nms_vs_oracle(cuda):
    """eval-mode forward -> boxes / scores / classes.  The oracle decode + NMS (oracle/inference.py, restating
    fcosv2.py:194-266) runs on the HIP model's own head outputs, so the comparison covers decode, top-k, sqrt, class-offset NMS
    (keep set bit-exact) and postprocess: boxes and scores within 1e-3 relative (north_star)."""
    from oracle import fcos_targets as ot
    from oracle import inference as oi
    from slenderobjdet_amd.data import synthetic_batch

    cfg, model, _ = _build(18, seed=3)
    with torch.no_grad():   # make the random-init head fire: lower the prior bias so a few thousand candidates pass the 0.05 threshold
        model.head.cls_pred.bias[:80] = -2.0
        model.head.cls_pred.weight[:80] *= 20
    model.arena.bump()
    model.eval()
    data = synthetic_batch(2, 256, 320, 21, device="cuda")
    for d in data:
        d["height"], d["width"] = 512, 640          # exercise detector_postprocess rescaling
    with torch.no_grad():
        assert len(model(data)) == 2                 # the public entry point
        # GroupNorm statistics are accumulated with float atomics, so two forward passes differ in the last bits and a candidate
        # sitting on the 0.05 threshold may flip: decode ONE set of tower outputs on both sides
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in model.in_features]
        ct, bt = model.head.run_towers(feats)
        cls_buf, box_buf, hw = model.head.predict(ct, bt)
        out = model.postprocess(model.inference(hw, ct, bt, imgs.image_sizes), data, imgs.image_sizes)
    locs = ot.locations(hw, model.fpn_strides)
    scales = model.head.scales.detach().cpu()
    bounds = [0]
    for h, w in hw:
        bounds.append(bounds[-1] + h * w)
    for i in range(2):
        cls_l, reg_l, ctr_l = [], [], []
        for l in range(5):
            sl = slice(bounds[l], bounds[l + 1])
            cls_l.append(cls_buf[i, sl, :80].cpu())
            reg_l.append(torch.exp(box_buf[i, sl, :4].cpu() * scales[l]))
            ctr_l.append(box_buf[i, sl, 4:5].cpu())
        rb, rs, rc, _ = oi.fcos_inference_single_image(locs, cls_l, reg_l, ctr_l, (256, 320))
        rb, ne = oi.detector_postprocess(rb, (256, 320), 512, 640)
        rb, rs, rc = rb[ne], rs[ne], rc[ne]
        inst = out[i]["instances"]
        assert len(inst) == len(rb) and len(rb) > 10, (len(inst), len(rb))
        # scores are computed with GPU vs CPU libm (last-ulp differences), so two detections with nearly equal scores may swap
        # places in the score-ordered output: compare after a canonical (class, x1, y1) ordering.  Keep-index bit-exactness on
        # identical inputs is covered by tests/test_gpu_detection_ops.py.
        def canon(b, sc, c):
            key = c.double() * 1e8 + b[:, 0].double().round() * 1e4 + b[:, 1].double().round()
            o = torch.argsort(key)
            return b[o], sc[o], c[o]

        gb, gs, gc = canon(inst.pred_boxes.tensor.cpu(), inst.scores.cpu(), inst.pred_classes.cpu())
        rb, rs, rc = canon(rb, rs, rc)
        assert torch.equal(gc, rc)
        assert (gs - rs).abs().max() <= 1e-3 * rs.abs().max()
        assert (gb - rb).abs().max() <= 1e-3 * rb.abs().max()
        assert (inst.scores[:-1] >= inst.scores[1:] - 1e-6).all()          # still emitted in descending score order


def test_fcos_r50_full_size_step(cuda):
    """BASELINE configs[1] - the configuration bench.py times - at its real depth and resolution (FCOS R50-FPN, 800x1344, L = 22 400
    locations; batch reduced to 2): targets bit-exact, the three losses within 1e-4 of oracle/losses.py evaluated on the product's own
    head outputs (fcosv2.py:104-148 restated; Scale + exp of fcosv2.py:372-378 applied on the CPU), and one finite SGD step that
    changes the loss.  The FCOS twin of test_retinanet_r50_full_size_step."""
    from bench import train_step
    from oracle import fcos_targets as ot
    from oracle import losses as ol
    from slenderobjdet_amd.data import synthetic_batch

    cfg, model, opt = _build(50)
    data = synthetic_batch(2, 800, 1333, 77, device="cuda")
    hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    inst = [d["instances"] for d in data]
    ref_l, ref_r = ot.targets_for_batch(hw, model.fpn_strides, [i.gt_boxes.tensor.cpu() for i in inst], [i.gt_classes.cpu() for i in inst],
                                        model.center_sampling_radius, 80)
    labels, reg_t, ctr_t, stats = model.get_ground_truth(hw, inst)
    assert labels.shape == (2, 22400)
    assert torch.equal(labels.cpu().long(), ref_l) and torch.equal(reg_t.cpu(), ref_r)
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        cls_t, box_t = model.head.run_towers([feats[f] for f in model.in_features])
        assert [tuple(t.shape[1:3]) for t in cls_t] == hw
        cls_buf, box_buf, _ = model.head.predict(cls_t, box_t)
    K = 80
    assert not model.head.norm_reg_targets               # fcosv2.py:377-380: exp(Scale(bbox_pred)) is the regression output
    cls_cpu, box_cpu = cls_buf.float().cpu(), box_buf.float().cpu()
    ctr_cpu = box_cpu[..., 4] if model.head.centerness_on_reg else cls_cpu[..., K]      # fcosv2.py:366-371: which tower carries centerness
    scales = model.head.scales.detach().float().cpu()
    pred, off = [], 0
    for l, (h, w) in enumerate(hw):
        pred.append(torch.exp(box_cpu[:, off:off + h * w, :4] * scales[l]))
        off += h * w
    pred = torch.cat(pred, 1)
    ref = ol.fcos_losses(ref_l.reshape(-1), ref_r.reshape(-1, 4), cls_cpu[..., :K].reshape(-1, K), pred.reshape(-1, 4), ctr_cpu.reshape(-1),
                         K, model.focal_loss_alpha, model.focal_loss_gamma, model.iou_loss_type)
    got = model(data)
    for k in ref:
        a, b = float(got[k].detach()), float(ref[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), (k, a, b)
    l0 = float(train_step(model, opt, data).detach())
    l1 = float(train_step(model, opt, data).detach())
    assert l0 == l0 and l1 == l1 and abs(l0) < 1e4 and abs(l1) < 1e4 and l1 != l0
    gn = float(model.arena.grads.float().norm())
    assert gn == gn and 0 < gn < 1e8


def test_side_streams_gradients_match_single_stream(cuda):
    """The weight gradients enqueued on the side stream (layers/functional.py:_wgrad_stream) and the box tower's nodes on the tower
    stream (FCOSHead.run_towers) must be complete when backward() returns.  In deterministic mode no float atomics are left, so
    every stream configuration must give the single-stream gradients bit for bit; a gradient still in flight (or lost, or read before
    its producer finished) would show up as a difference."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling.meta_arch import fcos as fcos_mod

    cfg, model, opt = _build(18, seed=3)
    data = synthetic_batch(2, 320, 384, 3, device="cuda")
    saved = {}
    prev = HF.WGRAD_SIDE_STREAM, fcos_mod.TOWER_STREAMS, HF.DETERMINISTIC
    HF.DETERMINISTIC = True
    try:
        for side, tower in ((True, True), (False, False), (True, False), (False, True), (True, True)):
            HF.WGRAD_SIDE_STREAM, fcos_mod.TOWER_STREAMS = side, tower
            opt.zero_grad()
            total = sum(model(data).values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            # no synchronize here: reading .grad on the current stream must already be ordered after the other streams
            g = model.arena.grads.clone()
            if side:
                assert HF._side_streams, "side stream was not used"
                assert not HF._side_join_queued, "end-of-backward join did not run"
            if tower:
                assert fcos_mod._tower_streams, "tower stream was not used"
            if (False, False) in saved:
                ref = saved[(False, False)]
                assert torch.equal(g, ref), ((side, tower), float((g - ref).abs().max()))
            saved.setdefault((side, tower), g)
        assert torch.equal(saved[(True, True)], saved[(False, False)])
        assert torch.isfinite(saved[(False, False)]).all() and float(saved[(False, False)].abs().sum()) > 0
    finally:
        HF.WGRAD_SIDE_STREAM, fcos_mod.TOWER_STREAMS, HF.DETERMINISTIC = prev


def test_sgd_kernel_matches_oracle(cuda):
    """Op-level parity of the fused SGD kernel (sod_sgd_step) with oracle/nn.py::sgd_step = torch.optim.SGD's single-tensor update
    (slender_det/solver/build.py:21-25): two segments with different lr multiplier / weight decay, momentum with and without
    nesterov, first step (buffer initialised from the gradient) and a later step, gradient scale 1/world.  fp32, 1e-6 relative."""
    import numpy as np

    from oracle import nn as onn
    from slenderobjdet_amd._C import call, ptr, stream_ptr

    g = torch.Generator().manual_seed(11)
    n0, n1 = 1000, 64 * 37          # segment boundaries are multiples of 64 elements in the arena
    n0p = (n0 + 63) // 64 * 64
    total = n0p + n1
    p = torch.randn(total, generator=g)
    grad = torch.randn(total, generator=g)
    buf = torch.randn(total, generator=g)
    segs = np.zeros(2, dtype=np.dtype([("b", "<i8"), ("e", "<i8"), ("lr", "<f4"), ("wd", "<f4")]))
    segs[0] = (0, n0p, 1.0, 1e-4)
    segs[1] = (n0p, total, 2.0, 0.0)
    segs_dev = torch.from_numpy(segs.view(np.uint8).copy()).to(cuda)
    for nesterov in (False, True):
        for first in (True, False):
            pd, gd, bd = p.clone().to(cuda), grad.clone().to(cuda), buf.clone().to(cuda)
            call("sod_sgd_step", ptr(pd), ptr(gd), ptr(bd), ptr(segs_dev), 2, None, 0.05, 0.9, 1 if nesterov else 0, 1 if first else 0, 0.5, stream_ptr())
            for (b, e, lr_mult, wd) in ((0, n0p, 1.0, 1e-4), (n0p, total, 2.0, 0.0)):
                rp, rb_ = onn.sgd_step(p[b:e], grad[b:e] * 0.5, buf[b:e], 0.05 * lr_mult, 0.9, wd, nesterov=nesterov, first=first)
                assert torch.allclose(pd[b:e].cpu(), rp, rtol=1e-6, atol=1e-7), (nesterov, first, b)
                assert torch.allclose(bd[b:e].cpu(), rb_, rtol=1e-6, atol=1e-7), (nesterov, first, b)


@pytest.mark.parametrize("kind", ["ADAM", "ADAMW", "ADAGRAD"])
def test_adaptive_optimizer_kernel_matches_torch_optim(cuda, kind):
    """SOLVER.OPTIM ADAM / ADAMW / ADAGRAD (slender_det/solver/build.py:26-31): the fused arena kernel (sod_adaptive_step) against the
    torch.optim class the reference constructs, run on the CPU in fp32 on the same parameters / gradients: five steps with changing
    gradients, two segments with different lr multiplier and weight decay, gradient scale 1/world.  1e-6 relative."""
    import numpy as np

    from slenderobjdet_amd._C import call, ptr, stream_ptr

    g = torch.Generator().manual_seed(5)
    n0p, n1 = 1024, 64 * 29
    total = n0p + n1
    p0 = torch.randn(total, generator=g)
    grads = [torch.randn(total, generator=g) * (0.5 + i) for i in range(5)]
    spec = ((0, n0p, 1.0, 1e-2), (n0p, total, 2.0, 0.0))
    segs = np.zeros(2, dtype=np.dtype([("b", "<i8"), ("e", "<i8"), ("lr", "<f4"), ("wd", "<f4")]))
    for i, sp in enumerate(spec):
        segs[i] = sp
    segs_dev = torch.from_numpy(segs.view(np.uint8).copy()).to(cuda)
    lr, scale = 3e-3, 0.5
    ref_p = [torch.nn.Parameter(p0[b:e].clone()) for b, e, _, _ in spec]
    groups = [{"params": [rp], "lr": lr * m, "weight_decay": wd} for rp, (_, _, m, wd) in zip(ref_p, spec)]
    ref = {"ADAM": torch.optim.Adam, "ADAMW": torch.optim.AdamW, "ADAGRAD": torch.optim.Adagrad}[kind](groups, lr)
    pd = p0.clone().to(cuda)
    m = torch.zeros(total, device=cuda) if kind != "ADAGRAD" else None
    v = torch.zeros(total, device=cuda)
    mode = {"ADAM": 0, "ADAMW": 1, "ADAGRAD": 2}[kind]
    eps = 1e-10 if kind == "ADAGRAD" else 1e-8
    for t, gr in enumerate(grads, start=1):
        for rp, (b, e, _, _) in zip(ref_p, spec):
            rp.grad = gr[b:e] * scale
        ref.step()
        bc1, bc2s = (1.0, 1.0) if kind == "ADAGRAD" else (1 - 0.9 ** t, (1 - 0.999 ** t) ** 0.5)
        call("sod_adaptive_step", ptr(pd), ptr(gr.to(cuda)), ptr(m), ptr(v), ptr(segs_dev), 2, mode, lr, 0.9, 0.999, eps, bc1, bc2s, scale, stream_ptr())
        for rp, (b, e, _, _) in zip(ref_p, spec):
            assert torch.allclose(pd[b:e].cpu(), rp.detach(), rtol=2e-6, atol=2e-7), (kind, t, b, float((pd[b:e].cpu() - rp.detach()).abs().max()))


def test_build_optimizer_returns_fused_classes_on_the_arena(cuda):
    """No SOLVER.OPTIM value leaves the HIP path on a GPU model: SGD -> FusedSGD, ADAM / ADAMW / ADAGRAD -> FusedAdaptive; a training
    step with each runs, changes every trainable parameter and keeps the loss finite; state_dict round-trips the moment buffers."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.solver import build_optimizer
    from slenderobjdet_amd.solver.build import FusedAdaptive, FusedSGD

    from bench import make_cfg, train_step

    data = synthetic_batch(2, 256, 256, 5, device="cuda")
    for optim in ("SGD", "ADAM", "ADAMW", "ADAGRAD"):
        cfg, model, _ = _build(18, seed=4)
        cfg.SOLVER.OPTIM = optim
        cfg.SOLVER.BASE_LR = 1e-4
        opt = build_optimizer(cfg, model)
        assert isinstance(opt, FusedSGD if optim == "SGD" else FusedAdaptive), (optim, type(opt))
        before = model.arena.params.detach().clone()
        l0 = float(train_step(model, opt, data))
        l1 = float(train_step(model, opt, data))
        assert l0 == l0 and l1 == l1 and not torch.equal(before, model.arena.params)
        if optim != "SGD":
            sd = opt.state_dict()
            assert sd["fused_adaptive"]["steps"] == 2 and float(sd["fused_adaptive"]["exp_avg_sq"].abs().sum()) > 0
            opt.load_state_dict(sd)
            assert opt._steps == 2


def test_fused_sgd_state_dict_round_trip(cuda):
    """save -> load -> step equals an uninterrupted run (the momentum buffer lives in the arena, not in Optimizer.state)."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.solver import build_optimizer

    from bench import train_step

    prev = HF.DETERMINISTIC
    HF.DETERMINISTIC = True       # bit-exact comparison of two runs
    try:
        data = synthetic_batch(2, 256, 256, 5, device="cuda")
        cfg, model, opt = _build(18, seed=2)
        for _ in range(2):
            train_step(model, opt, data)
        saved_opt = opt.state_dict()
        saved_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
        assert saved_opt["fused_sgd"]["steps"] == 2 and float(saved_opt["fused_sgd"]["momentum"].abs().sum()) > 0
        train_step(model, opt, data)
        want = model.arena.params.detach().clone()
        # "resume": a fresh model + optimizer restored from the checkpoint
        cfg2, model2, opt2 = _build(18, seed=99)
        model2.load_state_dict(saved_model)
        model2.arena.bump()
        opt2.load_state_dict(saved_opt)
        train_step(model2, opt2, data)
        assert torch.equal(model2.arena.params, want)
        # without the momentum the resumed run would differ
        cfg3, model3, opt3 = _build(18, seed=99)
        model3.load_state_dict(saved_model)
        model3.arena.bump()
        train_step(model3, opt3, data)
        assert not torch.equal(model3.arena.params, want)
    finally:
        HF.DETERMINISTIC = prev


def test_losses_method_with_reference_contract(cuda):
    """FCOSV2.losses(gt_classes, reg_targets, per-level NCHW predictions) - the reference's signature (fcosv2.py:104) - gives the
    same three values as the fused training forward on the same predictions, and is differentiable."""
    from slenderobjdet_amd.data import synthetic_batch

    from slenderobjdet_amd.layers import functional as HF

    cfg, model, _ = _build(18, seed=5)
    data = synthetic_batch(2, 256, 320, 13, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True      # GroupNorm statistics without float atomics: both tower passes give the same bits
    try:
        got = model(data)
        with torch.no_grad():
            imgs = model.preprocess_image(data)
            feats = model.backbone(imgs.tensor)
            ct, bt = model.head.run_towers([feats[f] for f in model.in_features])
            cls_buf, box_buf, hw = model.head.predict(ct, bt)
            labels, reg_t, _ctr_t, _stats = model.get_ground_truth(hw, [d["instances"].to("cuda") for d in data])
    finally:
        HF.DETERMINISTIC = prev
    N, K = 2, model.num_classes
    scales = model.head.scales.detach()
    cls_l, reg_l, ctr_l, off = [], [], [], 0
    for l, (h, w) in enumerate(hw):
        sl = slice(off, off + h * w)
        cls_l.append(cls_buf[:, sl, :K].reshape(N, h, w, K).permute(0, 3, 1, 2).contiguous().requires_grad_(True))
        reg_l.append(torch.exp(box_buf[:, sl, :4] * scales[l]).reshape(N, h, w, 4).permute(0, 3, 1, 2).contiguous().requires_grad_(True))
        ctr_l.append(box_buf[:, sl, 4:5].reshape(N, h, w, 1).permute(0, 3, 1, 2).contiguous().requires_grad_(True))
        off += h * w
    out = model.losses(labels.long(), reg_t, cls_l, reg_l, ctr_l)
    for k in ("cls_loss", "reg_loss", "centerness_loss"):
        a, b = float(out[k].detach()), float(got[k].detach())
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-3), (k, a, b)
    sum(out.values()).backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in cls_l + reg_l + ctr_l)
    assert cls_l[0].grad.abs().sum() > 0 and reg_l[0].grad.abs().sum() > 0 and ctr_l[0].grad.abs().sum() > 0


def test_prefetched_frozen_prefix_is_bit_identical(cuda):
    """``model.prefetch(next_batch)`` (preprocess + frozen stem/res2 of the next batch on a side stream, enqueued between forward and
    backward of the current one) must not change anything: losses and gradients of the next step equal the un-pipelined step's bit for
    bit (deterministic mode), and a batch that was not prefetched takes the normal path."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF

    cfg, model, opt = _build(50, seed=2)
    a = synthetic_batch(2, 320, 384, 5, device="cuda")
    b = synthetic_batch(2, 320, 384, 6, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        def step(data, nxt=None):
            opt.zero_grad()
            hits = getattr(model, "prefetch_hits", 0)
            out = model(data)
            took = getattr(model, "prefetch_hits", 0) == hits + 1
            if nxt is not None:
                assert model.prefetch(nxt)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone(), took

        ref_l, ref_g, _ = step(b)
        step(a, nxt=b)
        assert model._prefetched is not None and model._prefetched[0] is b
        got_l, got_g, took = step(b)
        assert took and model._prefetched is None, "the prefetched prefix was not consumed"
        assert got_l == ref_l
        assert torch.equal(got_g, ref_g)
        # a different list object (even with equal contents) is not served from the prefetch slot
        step(a, nxt=b)
        other_l, other_g, took = step(list(b))
        assert not took and other_l == ref_l and torch.equal(other_g, ref_g)
    finally:
        HF.DETERMINISTIC = prev


def test_deferred_lateral_dgrad_matches_separate_passes(cuda):
    """The FPN lateral convs leave their data gradient to the ResNet stage that produced their input (layers/nn.py DeferSlot): the
    stage runs dgrad(lateral) + other consumers' gradient + its output ReLU mask as ONE launch instead of dgrad -> autograd add ->
    relu_bwd.  Same losses bit for bit (forward is untouched); gradients agree to bf16 rounding (the fused form rounds once instead
    of three times), and the optimisation really is taken (the slots are consumed)."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.layers import nn as HN

    cfg, model, opt = _build(50, seed=4)
    data = synthetic_batch(2, 320, 384, 9, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    calls = {"fused": 0}
    orig = HF.conv2d_dgrad

    def counting(dy, wt, x_hw, stride=1, pad=0, dil=1, accum=None, relu_mask=None, **kw):
        if (relu_mask is not None or kw.get("relu_bits") is not None) and wt.data_ptr() in lateral_ptrs:
            calls["fused"] += 1
        return orig(dy, wt, x_hw, stride, pad, dil, accum=accum, relu_mask=relu_mask, **kw)

    try:
        def step(defer):
            HN.DEFER_LATERAL_DGRAD = defer
            opt.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        ref_l, ref_g = step(False)
        lateral_ptrs = {m.wt_bf16.data_ptr() for m in model.backbone.lateral_convs}
        HF.conv2d_dgrad = counting
        got_l, got_g = step(True)
        HF.conv2d_dgrad = orig
        assert calls["fused"] == 3, calls            # res3, res4, res5 outputs
        assert got_l == ref_l
        assert torch.isfinite(got_g).all()
        num = (got_g - ref_g).norm().item()
        den = ref_g.norm().item()
        assert num <= 2e-2 * den, (num, den)
        # per-parameter: the backbone gradients (the only ones that can change) stay within bf16 noise of the three-pass form
        for name, off, n in model.arena.names:
            if "bottom_up" in name:
                a, b = got_g[off:off + n], ref_g[off:off + n]
                assert (a - b).norm().item() <= 5e-2 * max(b.norm().item(), 1e-12), name
    finally:
        HF.conv2d_dgrad = orig
        HN.DEFER_LATERAL_DGRAD = True
        HF.DETERMINISTIC = prev


def test_relu_bit_masks_leave_the_step_bit_identical(cuda):
    """Trainable bottleneck stages record their block-output ReLU masks as bit arrays (resnet.RELU_BITS) instead of re-reading the bf16
    block outputs in backward: losses and every gradient are bit-identical (deterministic mode) with and without."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling.backbone import resnet

    cfg, model, opt = _build(50, seed=6)
    data = synthetic_batch(2, 320, 384, 11, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    try:
        def step(on):
            resnet.RELU_BITS = on
            opt.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        ref_l, ref_g = step(False)
        got_l, got_g = step(True)
        assert got_l == ref_l
        assert torch.equal(got_g, ref_g)
    finally:
        resnet.RELU_BITS = True
        HF.DETERMINISTIC = prev


def test_tower_input_gradients_folded_into_the_second_dgrad(cuda):
    """The two towers of FCOSHead read the same FPN outputs (fcosv2.py:342-361).  The tower whose backward runs second adds the first
    one's data gradient in the epilogue of its own launch (layers/nn.py SiblingFold) instead of leaving five elementwise additions to
    autograd: same losses bit for bit (deterministic mode; the forward pass is untouched), gradients equal to the accumulation form up
    to ONE bf16 rounding of the FPN-output gradient instead of two, the accum launch really runs, and a backward pass that reaches only
    one tower raises instead of losing the parked gradient."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.layers import nn as HN
    from slenderobjdet_amd.modeling.meta_arch import fcos as FC

    prev_det, keep = HF.DETERMINISTIC, FC.TOWER_FOLD
    HF.DETERMINISTIC = True
    cfg, model, opt = _build(50, seed=8)
    data = synthetic_batch(2, 320, 384, 13, device="cuda")
    calls = {"accum": 0}
    orig = HF.conv2d_dgrad_ml

    def counting(*a, **k):
        calls["accum"] += k.get("accums") is not None
        return orig(*a, **k)

    def step(on):
        FC.TOWER_FOLD = on
        opt.zero_grad()
        out = model(data)
        total = sum(out.values())
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
        torch.cuda.synchronize()
        return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

    try:
        ref_l, ref_g = step(False)
        HF.conv2d_dgrad_ml = counting
        got_l, got_g = step(True)
        again_l, again_g = step(True)
        assert calls["accum"] == 2, calls
        assert got_l == ref_l
        assert torch.equal(got_g, again_g)                                  # still deterministic
        head = [(n, o, c) for n, o, c in model.arena.names if n.startswith("head.")]
        for name, off, n in head:                                           # nothing above the FPN outputs changes
            assert torch.equal(got_g[off:off + n], ref_g[off:off + n]), name
        worst = 0.0
        for name, off, n in model.arena.names:
            a, b = got_g[off:off + n], ref_g[off:off + n]
            d = (a - b).norm().item() / max(b.norm().item(), 1e-12)
            worst = max(worst, d)
            assert d <= 3e-2, (name, d)                                     # one bf16 rounding (2^-9 per element) through the backbone
        assert 0.0 < worst
        # a graph that feeds only ONE tower: the parked gradient must not vanish silently
        FC.TOWER_FOLD = True
        images = model.preprocess_image(data)
        feats = model.backbone(images.tensor)
        ct, bt = model.head.run_towers([feats[f] for f in model.in_features], fold_input_grads=True)
        opt.zero_grad()
        model.arena.begin_backward()
        with pytest.raises(RuntimeError, match="SiblingFold"):
            sum(t.float().sum() for t in ct).backward()
        model.arena.finish_backward()
    finally:
        HF.conv2d_dgrad_ml = orig
        FC.TOWER_FOLD = keep
        HF.DETERMINISTIC = prev_det
        torch.cuda.synchronize()


def test_compact_stride2_input_gradient_is_bit_identical(cuda):
    """A bottleneck stage that opens with stride-2 1x1 convolutions leaves its input gradient in compact (N, H/2, W/2, C) form to the
    producing stage, whose fused launch adds it at the even positions (resnet.COMPACT_S2_GRAD, layers/nn.py DeferSlot.comp) instead of
    scattering it into a zero tensor first: losses and every gradient are bit-identical (deterministic mode) with and without."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling.backbone import resnet

    cfg, model, opt = _build(50, seed=10)
    data = synthetic_batch(2, 320, 384, 15, device="cuda")
    prev, HF.DETERMINISTIC = HF.DETERMINISTIC, True
    calls = {"even": 0}
    orig = HF.conv2d_dgrad

    def counting(*a, **k):
        calls["even"] += int(bool(k.get("accum_even")))
        return orig(*a, **k)

    try:
        def step(on):
            resnet.COMPACT_S2_GRAD = on
            opt.zero_grad()
            out = model(data)
            total = sum(out.values())
            model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
            return {k: float(v.detach()) for k, v in out.items()}, model.arena.grads.clone()

        ref_l, ref_g = step(False)
        HF.conv2d_dgrad = counting
        got_l, got_g = step(True)
        HF.conv2d_dgrad = orig
        assert calls["even"] == 2, calls          # res3 <- res4 and res4 <- res5
        assert got_l == ref_l
        assert torch.equal(got_g, ref_g)
    finally:
        HF.conv2d_dgrad = orig
        resnet.COMPACT_S2_GRAD = True
        HF.DETERMINISTIC = prev
