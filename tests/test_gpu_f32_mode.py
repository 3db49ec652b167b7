"""The fp32-STORAGE validation mode (SOD_PRECISION=fp32; csrc/f32_path.hip, layers/functional_f32.py): op-level parity of every
``sod_*_f32`` kernel with the CPU oracle (oracle/nn.py = F.conv2d / F.group_norm in fp32) at 1e-5, and the whole FCOS training step -
losses AND every parameter gradient - against the plain fp32 oracle (= the reference's CPU path restated,
slender_det/modeling/meta_arch/fcos/fcosv2.py:63-148) far inside north_star's 1e-3: with storage rounding out of the picture the layer
code, the fused epilogue semantics (residual, up-sampled residual, accumulate, masks), the target assignment, the loss kernels and the
optimizer are what is left to compare.  The 100-iteration statement lives in tests/test_gpu_parity100.py."""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def f32mode():
    from slenderobjdet_amd.layers import functional as HF

    prev = HF.set_precision("fp32")
    yield HF
    HF.set_precision(prev)


def _close(a, b, tol=1e-5):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, (err, tuple(a.shape))


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad,dil", [
    (2, 13, 17, 8, 24, 3, 1, 1, 1),       # ragged tile edges in every dimension
    (2, 14, 18, 16, 8, 1, 1, 0, 1),
    (1, 16, 20, 8, 72, 7, 2, 3, 1),       # the stem shape family (input channels padded to 8)
    (2, 12, 16, 40, 32, 3, 2, 1, 1),
    (2, 12, 16, 24, 24, 1, 2, 0, 1),      # stride-2 1x1 (STRIDE_IN_1X1 shortcuts)
    (1, 11, 13, 8, 16, 3, 1, 2, 2),       # dilation 2
])
def test_conv_f32_fwd_dgrad_wgrad_vs_oracle(cuda, f32mode, N, H, W, C, K, R, stride, pad, dil):
    from oracle import nn as onn

    HF = f32mode
    g = torch.Generator().manual_seed(H * 100 + K)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, R, R, C, generator=g) * 0.2
    bias = torch.randn(K, generator=g)
    ref = onn.conv2d(x, w, bias, stride, pad, dil)
    Ho, Wo = ref.shape[1], ref.shape[2]
    res = torch.randn(N, Ho, Wo, K, generator=g)
    xd, wd, bd = x.to(cuda), w.to(cuda), bias.to(cuda)
    wk, wt = HF.weight_prep(wd)
    assert wk.dtype == torch.float32 and wt is wk
    _close(HF.conv2d_fwd(xd, wk, bd, None, stride, pad, dil), ref)
    _close(HF.conv2d_fwd(xd, wk, bd, res.to(cuda), stride, pad, dil, relu=True), onn.conv2d(x, w, bias, stride, pad, dil, res=res, relu=True))
    if Ho % 2 == 0 and Wo % 2 == 0:
        small = torch.randn(N, Ho // 2, Wo // 2, K, generator=g)
        _close(HF.conv2d_fwd(xd, wk, bd, small.to(cuda), stride, pad, dil, res_up2=True), onn.conv2d(x, w, bias, stride, pad, dil, res=small, res_up2=True))
    # FrozenBN scale folded into the compute copy
    sc = torch.rand(K, generator=g) + 0.5
    wks, _ = HF.weight_prep(wd, sc.to(cuda))
    _close(HF.conv2d_fwd(xd, wks, None, None, stride, pad, dil), onn.conv2d(x, w * sc.view(-1, 1, 1, 1), None, stride, pad, dil))
    # backward
    dy = torch.randn(N, Ho, Wo, K, generator=g)
    dx_ref, dw_ref = onn.conv2d_backward(x, w, dy, stride, pad, dil)
    dyd = dy.to(cuda)
    _close(HF.conv2d_dgrad(dyd, wt, (H, W), stride, pad, dil), dx_ref)
    acc = torch.randn(N, H, W, C, generator=g)
    mask = torch.randn(N, H, W, C, generator=g)
    _close(HF.conv2d_dgrad(dyd, wt, (H, W), stride, pad, dil, accum=acc.to(cuda), relu_mask=mask.to(cuda)), (dx_ref + acc) * (mask > 0))
    if H % 2 == 0 and W % 2 == 0:
        comp = torch.randn(N, H // 2, W // 2, C, generator=g)
        dense = torch.zeros(N, H, W, C)
        dense[:, ::2, ::2] = comp
        _close(HF.conv2d_dgrad(dyd, wt, (H, W), stride, pad, dil, accum=comp.to(cuda), accum_even=True), dx_ref + dense)
    dw = torch.full((K, R, R, C), 0.25, device=cuda)      # the kernel ACCUMULATES into the gradient arena
    HF.conv2d_wgrad(dyd, xd, dw, R, R, stride, pad, dil)
    _close(dw - 0.25, dw_ref, 2e-5)
    dw2 = torch.zeros((K, R, R, C), device=cuda)
    HF.conv2d_wgrad(dyd, xd, dw2, R, R, stride, pad, dil, qscale=sc.to(cuda))
    _close(dw2, dw_ref * sc.view(-1, 1, 1, 1), 2e-5)


def test_conv_f32_image_strides_into_concatenated_buffers(cuda, f32mode):
    """The prediction convs write all levels into one (N, L, K) buffer and read their gradients from one (fcos.py predict / loss)."""
    from oracle import nn as onn

    HF = f32mode
    g = torch.Generator().manual_seed(3)
    N, C, K = 2, 16, 8
    hws = [(6, 8), (3, 4)]
    L = sum(h * w for h, w in hws)
    w = torch.randn(K, 3, 3, C, generator=g) * 0.2
    xs = [torch.randn(N, h, wd, C, generator=g) for h, wd in hws]
    buf = torch.zeros(N, L, K, device=cuda)
    wk, wt = HF.weight_prep(w.to(cuda))
    offs = [0, hws[0][0] * hws[0][1]]
    HF.conv2d_fwd_ml([x.to(cuda) for x in xs], wk, None, 1, 1, 1, outs=[buf.view(-1)[o * K:] for o in offs], y_img_stride=L * K)
    ref = torch.cat([onn.conv2d(x, w, None, 1, 1, 1).reshape(N, -1, K) for x in xs], 1)
    _close(buf, ref)
    dbuf = torch.randn(N, L, K, generator=g)
    dys = [dbuf.to(cuda).view(-1)[o * K:] for o in offs]
    dxs = HF.conv2d_dgrad_ml(dys, wt, hws, 1, 1, 1, dy_img_stride=L * K, N=N)
    dw = torch.zeros(K, 3, 3, C, device=cuda)
    HF.conv2d_wgrad_ml(dys, [x.to(cuda) for x in xs], dw, 3, 3, 1, 1, 1, dy_img_stride=L * K, K=K)
    dw_ref = torch.zeros(K, 3, 3, C)
    for (h, wd), o, x, dx in zip(hws, offs, xs, dxs):
        dy = dbuf[:, o:o + h * wd].reshape(N, h, wd, K)
        dx_ref, dwl = onn.conv2d_backward(x, w, dy, 1, 1, 1)
        _close(dx, dx_ref)
        dw_ref += dwl
    _close(dw, dw_ref, 2e-5)
    db = torch.zeros(K, device=cuda)
    HF.bias_grad(dbuf.to(cuda), db, N, L, K)
    _close(db, dbuf.sum((0, 1)), 2e-5)


@pytest.mark.parametrize("relu", [False, True])
def test_groupnorm_f32_vs_oracle(cuda, f32mode, relu):
    from oracle import nn as onn

    HF = f32mode
    g = torch.Generator().manual_seed(5)
    N, C, G = 2, 64, 8
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    xs = [torch.randn(N, h, w, C, generator=g) * 2 + 0.5 for h, w in ((9, 7), (4, 5))]
    ys, stats = HF.groupnorm_fwd_ml([x.to(cuda) for x in xs], gamma.to(cuda), beta.to(cuda), G, 1e-5, relu=relu)
    assert tuple(stats.shape) == (2, N, G, 2)
    dys = [torch.randn(x.shape, generator=g) for x in xs]
    dgam, dbet, dxs_sum = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
    dxs = HF.groupnorm_bwd_ml([d.to(cuda) for d in dys], [x.to(cuda) for x in xs], gamma.to(cuda), beta.to(cuda), stats, G, dgam, dbet, relu=relu,
                              dxsum=dxs_sum)
    rg, rb_, rs = torch.zeros(C), torch.zeros(C), torch.zeros(C)
    for x, y, dy, dx in zip(xs, ys, dys, dxs):
        _close(y, onn.group_norm(x, gamma, beta, G, relu=relu))
        xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        out = onn.group_norm(xr, gr, br, G, relu=relu)
        a, b, c = torch.autograd.grad(out, (xr, gr, br), dy)
        _close(dx, a, 2e-5)
        rg += b; rb_ += c; rs += a.sum((0, 1, 2))
    _close(dgam, rg, 2e-5)
    _close(dbet, rb_, 2e-5)
    assert float((dxs_sum.cpu() - rs).abs().max()) <= 2e-4        # sums of signed terms that nearly cancel: absolute bar


def test_elementwise_f32_kernels(cuda, f32mode):
    import torch.nn.functional as F

    HF = f32mode
    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(2, 6, 8, 16, generator=g), torch.randn(2, 6, 8, 16, generator=g)
    ad, bd = a.to(cuda), b.to(cuda)
    assert torch.equal(HF.relu_fwd(ad).cpu(), torch.relu(a))
    assert torch.equal(HF.relu_bwd(ad, bd).cpu(), a * (b > 0))
    assert torch.equal(HF.add_bf16(ad, bd).cpu(), a + b)
    small = torch.randn(2, 3, 4, 16, generator=g)
    up = F.interpolate(small.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(HF.add_up2(ad, small.to(cuda)).cpu(), a + up)
    _close(HF.upsample2x_bwd(ad), a.reshape(2, 3, 2, 4, 2, 16).sum((2, 4)), 1e-6)
    x = torch.randn(2, 9, 11, 8, generator=g)
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(HF.maxpool3x3s2(x.to(cuda)).cpu(), ref)
    img = torch.randint(0, 256, (3, 10, 13), dtype=torch.uint8, generator=g)
    out = torch.empty(12, 16, 8, device=cuda)
    HF.preprocess_image(img.to(cuda), out, (103.53, 116.28, 123.675), (1.0, 2.0, 0.5))
    want = torch.zeros(12, 16, 8)
    want[:10, :13, :3] = ((img.float() - torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)) / torch.tensor([1.0, 2.0, 0.5]).view(3, 1, 1)).permute(1, 2, 0)
    _close(out, want, 1e-6)


FORCED_STATS = []       # (units decided differently, units) of every forced-mask oracle pass of this session (printed by the last test)


@contextlib.contextmanager
def _forced(masks):
    """The oracle code inside takes the PRODUCT's ReLU decisions (oracle.nn.ForcedMasks) wherever ``masks`` has the position."""
    from oracle.nn import ForcedMasks

    st = ForcedMasks.begin(masks)
    try:
        yield st
    finally:
        ForcedMasks.end()
    # shared decisions are a CHECK, not a trust: the product and this oracle run may decide a unit differently only where its
    # pre-activation is zero to rounding (|x| < 1e-4 rms of the activation); a wrong-sign pre-activation in the product fails here
    assert st["outside"] == 0, ("ReLU decisions differ outside the undecided band", st["outside_at"][:5])
    FORCED_STATS.append((st["disagree"], st["units"]))


def _tapped_step(model, opt, data):
    """One forward + backward of the product in the validation mode with its ReLU decisions recorded.  Returns (losses, masks)."""
    from oracle.conditioning import ProductReluTap

    with ProductReluTap() as tap:
        got = model(data)
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    masks, unmatched = tap.masks_for(model)
    assert not unmatched, unmatched[:5]
    return got, masks


def _forced_oracle_grads(make_oracle, losses_of, masks, dtypes=("f32", "f64")):
    """{tag: (losses, grads)} of the oracle built by ``make_oracle()`` in fp32 and float64, with the PRODUCT's ReLU decisions forced on
    (oracle.nn.ForcedMasks); also returns the positions the oracle asked for without getting a mask."""
    refs, missed = {}, []
    for tag in dtypes:
        oracle = make_oracle()
        if tag == "f64":
            oracle.double()
        with _forced(masks) as st:
            losses = losses_of(oracle)
            tr = oracle.trainable()
            grads = dict(zip(tr.keys(), torch.autograd.grad(sum(losses.values()), list(tr.values()), allow_unused=True)))
        missed = st["missed"]
        refs[tag] = ({k: float(v.detach()) for k, v in losses.items()}, grads)
    return refs, missed


def _assert_gradients_tight(model, refs, what, max_pair=1e-4, share_2e5=0.9, real_rows=None, min_checked=1):
    """Every parameter gradient of the HIP fp32 run within ``max_pair`` (1e-4) of its norm from the float64 oracle - and from the CPU
    fp32 oracle where one was run -, ``share_2e5`` of the tensors within 2e-5: bars a localised kernel error of a few 1e-4 cannot pass.
    They are possible because all runs differentiate the same piecewise-linear function (the product's ReLU decisions; measured
    worst case 7e-6 on R18 and R50 alike).  ``real_rows``: {substring of a parameter name: real leading rows of a padded tensor}."""
    rows = []
    for name, p in model.named_parameters():
        if not p.requires_grad or refs["f64"][1].get(name) is None:
            continue
        gq = p.grad.detach().double().cpu()
        if gq.dim() == 4:
            gq = gq.permute(0, 3, 1, 2)
        r64 = refs["f64"][1][name]
        r32 = refs["f32"][1][name].double() if "f32" in refs else r64
        for key, nrow in (real_rows or {}).items():
            if key in name:
                assert (gq[nrow:] == 0).all(), name
                gq, r32, r64 = gq[:nrow], r32[:nrow], r64[:nrow]
        if gq.shape != r64.shape:          # padded prediction / offset rows beyond the reference's
            assert gq.shape[1:] == r64.shape[1:] and (gq[r64.shape[0]:] == 0).all(), (name, gq.shape, r64.shape)
            gq = gq[: r64.shape[0]]
        n = max(r64.norm().item(), 1e-30)
        rows.append(((gq - r32).norm().item() / n, (gq - r64).norm().item() / n, (r32 - r64).norm().item() / n, name))
    assert len(rows) >= min_checked, len(rows)
    worst = max(rows)
    share = sum(r[0] <= 2e-5 and r[1] <= 2e-5 for r in rows) / len(rows)
    print(f"\n{what}: {len(rows)} tensors, worst hip32-cpu32 {worst[0]:.2e} ({worst[3]}), worst hip32-f64 {max(r[1] for r in rows):.2e}, "
          f"worst cpu32-f64 {max(r[2] for r in rows):.2e}, share <= 2e-5: {share:.3f}")
    for d_pair, d64, _, name in rows:
        assert d_pair <= max_pair and d64 <= max_pair, (what, name, d_pair, d64)
    assert share >= share_2e5, (what, share, sorted(rows)[-5:])


@pytest.mark.parametrize("depth", [18, 50])
def test_fcos_step_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode, depth):
    """One FCOS training step (BASELINE configs[0] shape family; R50 = the headline's depth) in the validation mode against the plain
    fp32 oracle (= the reference's CPU path restated) and the same oracle in float64: the three losses to 2e-5 relative - 50x inside
    north_star's 1e-3 - and EVERY parameter gradient to 1e-4 of its norm, 90 % of them to 2e-5 (measured worst case 7e-6).

    What makes bars of that size possible on a random-init ResNet: two correct fp32 implementations decide the sign of a pre-activation
    that cancels to within rounding of zero differently, the regression branch's gradient is sparse (positives only), and ONE such
    unit at P5 moves every backbone weight gradient by 2e-3 ... 1e-2 of its norm (measured: CPU fp32 oracle against float64, R18
    2.3e-3, R50 1.1e-2; with the decisions of one run forced onto the other 7e-6 and 8e-6).  So the oracle takes the PRODUCT's ReLU
    decisions (layers/functional_f32.RELU_TAP -> oracle.conditioning.ProductReluTap -> oracle.nn.ForcedMasks): all three runs then
    differentiate the same piecewise-linear function, and what is left is summation order.  Until round 4 these tests carried caps of
    5e-3 (R18) and 2e-2 (R50) for those discrete events."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(depth)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256 if depth == 18 else 192, 320 if depth == 18 else 256, 3, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    got, masks = _tapped_step(model, opt, data)
    refs, missed = _forced_oracle_grads(lambda: OracleFCOS.from_hip_model(model, emulate_bf16=False), lambda o: o.losses(cpu), masks)
    assert not missed, missed[:5]          # every ReLU position of the oracle took the product's decisions
    for k, b in refs["f32"][0].items():
        a = float(got[k].detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
        assert abs(a - refs["f64"][0][k]) <= 2e-5 * max(abs(b), 1e-3), (k, a, refs["f64"][0][k])
    _assert_gradients_tight(model, refs, f"f32 mode FCOS R{depth}")


@pytest.mark.parametrize("box_reg", ["smooth_l1", "giou"])
def test_retinanet_step_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode, box_reg):
    """BASELINE configs[2]'s step (RetinaNet, R18 stand-in for the depth) in the validation mode against oracle.model.OracleRetinaNet in
    fp32 and float64: both losses to 2e-5 relative, the EMA loss normaliser, and every parameter gradient no further from the float64
    arbiter than 1.5x the CPU fp32 oracle is (+1e-4) and within 1e-3 of the CPU fp32 oracle outright (the R18 bars of the FCOS test)."""
    from bench import make_cfg
    from oracle.model import OracleRetinaNet
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(18, "retinanet")
    cfg.MODEL.RETINANET.BBOX_REG_LOSS_TYPE = box_reg
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 3, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    norm0 = float(model.loss_normalizer)          # the EMA state BEFORE the step: the oracle is built after the product has moved it
    got, masks = _tapped_step(model, opt, data)
    norms = []

    def losses_of(oracle):
        oracle.c["normalizer"] = norm0
        out = oracle.losses(cpu)
        norms.append(oracle.new_normalizer)
        return out

    refs, missed = _forced_oracle_grads(lambda: OracleRetinaNet.from_hip_model(model, emulate_bf16=False), losses_of, masks)
    assert not missed, missed[:5]
    assert abs(float(model.loss_normalizer) - norms[0]) < 1e-4
    for k, b in refs["f32"][0].items():
        a = float(got[k].detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
        assert abs(a - refs["f64"][0][k]) <= 2e-5 * max(abs(b), 1e-3), (k, a, refs["f64"][0][k])
    # the ReLU decisions are the product's in all three runs (see the FCOS test above); the GIoU / smooth-L1 kinks are not forced
    _assert_gradients_tight(model, refs, f"f32 mode RetinaNet {box_reg}")


@pytest.mark.parametrize("v2", [False, True])
def test_dfconv2d_module_in_f32_mode_vs_oracle(cuda, f32mode, v2):
    """DFConv2d (slender_det/layers/df_conv.py:67-78: offset conv -> DeformConv / ModulatedDeformConv) as an autograd module with fp32
    storage against the CPU oracle: output and the gradients w.r.t. x, the offset conv (through dOffset AND dMask) and the DCN weight to
    2e-5 of their norm - the bf16 product path holds 2^-6 / 3e-2 on the same module (test_gpu_deform_conv.py), i.e. those bars are storage
    precision, not kernel error.  Offsets of up to ~3 px so that samples cross pixel boundaries and the image border."""
    import torch.nn.functional as F

    from oracle import deform_conv as odc
    from slenderobjdet_amd.layers.arena import ParamArena
    from slenderobjdet_amd.layers.deform_conv import DFConv2d
    from slenderobjdet_amd.layers.nn import attach_arena

    torch.manual_seed(1)
    C = 64
    m = DFConv2d(C, C, with_modulated_dcn=v2).to(cuda)
    with torch.no_grad():
        m.offset.weight.mul_(2.0)
    arena = ParamArena(m)
    attach_arena(m, arena)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 9, 11, C, generator=g) * 0.5
    dy = torch.randn(2, 9, 11, C, generator=g)
    xd = x.to(cuda).requires_grad_(True)
    arena.zero_grad()
    y = m(xd)
    assert y.dtype == torch.float32
    y.backward(dy.to(cuda))
    torch.cuda.synchronize()
    n = m.n_off
    w_off = m.offset.weight.detach().cpu()[:n].permute(0, 3, 1, 2).double().requires_grad_(True)
    b_off = m.offset.bias.detach().cpu()[:n].double().requires_grad_(True)
    w = m.conv.weight.detach().cpu().permute(0, 3, 1, 2).double().requires_grad_(True)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    om = F.conv2d(xr, w_off, b_off, padding=1)
    assert om[:, :18].abs().max() > 1.0, "offsets must leave the sampling cell"
    if v2:
        ref = odc.deform_conv2d(xr, om[:, :18], w, None, 1, 1, 1, om[:, 18:27].sigmoid(), 1)
    else:
        ref = odc.deform_conv2d(xr, om, w, None, 1, 1, 1, None, 1)
    gx, gwo, gbo, gw = torch.autograd.grad(ref, (xr, w_off, b_off, w), dy.permute(0, 3, 1, 2).double())
    _close(y, ref.detach().permute(0, 2, 3, 1), 2e-5)
    for got, want, name in ((m.offset.weight.grad.cpu()[:n].permute(0, 3, 1, 2), gwo, "d offset.weight"), (m.offset.bias.grad.cpu()[:n], gbo, "d offset.bias"),
                            (m.conv.weight.grad.cpu().permute(0, 3, 1, 2), gw, "d conv.weight"), (xd.grad.cpu().permute(0, 3, 1, 2), gx, "dx")):
        err = (got.double() - want).norm().item() / max(want.norm().item(), 1e-12)
        assert err <= 2e-5, (name, err)
    assert (m.offset.weight.grad[n:] == 0).all()


def test_reppoints_step_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode):
    """BASELINE configs[3]'s step (RepPointsDetector: ResNet + GN-FPN + two DeformConv layers fed by the learned point offsets,
    rpd.py:621-671; R18 stand-in for the depth) in the validation mode against oracle.reppoints.OracleRepPoints in fp32 and float64.
    The point-to-box assignment depends on the PREDICTED init boxes, so the oracle is given the labels of the run under test after they
    were checked bit-exact against the oracle's own assignment on the same boxes.  Losses to 2e-5; every parameter gradient within 1e-4
    of the float64 AND of the CPU fp32 oracle, all three on the product's ReLU decisions (see the FCOS test)."""
    from oracle import reppoints as orp
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer
    from test_gpu_reppoints import _cfg

    cfg = _cfg()
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 256, 320, 5, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    from oracle.conditioning import ProductReluTap
    with ProductReluTap() as tap:
        got = model(data)
        tg_hip = model.last_targets
        total = sum(got.values())
        opt.zero_grad()
        model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    masks, unmatched = tap.masks_for(model)
    assert not unmatched, unmatched[:5]
    refs = {}
    for tag in ("f32", "f64"):
        oracle = orp.OracleRepPoints.from_hip_model(model, emulate_bf16=False)
        oracle.normalizer = 20.0
        with _forced(masks) as st:       # the product's ReLU decisions in both oracle runs (see the FCOS test)
            if tag == "f64":
                oracle.double()
                losses = oracle.losses(cpu, targets=tg)
            else:
                losses = oracle.losses(cpu)              # the CPU fp32 oracle's own assignment on its own init boxes
                tg = oracle.last_targets
                assert torch.equal(tg_hip[0].cpu().float(), tg[0].float()) and torch.equal(tg_hip[2].cpu().long(), tg[2].long()), "labels differ"
                assert torch.equal(tg_hip[1].cpu(), tg[1]) and torch.equal(tg_hip[3].cpu(), tg[3])
            tr = oracle.trainable()
            grads = dict(zip(tr.keys(), torch.autograd.grad(sum(losses.values()), list(tr.values()), allow_unused=True)))
        assert not st["missed"], st["missed"][:5]
        refs[tag] = ({k: float(v.detach()) for k, v in losses.items()}, grads)
    for k, b in refs["f32"][0].items():
        a = float(got[k].detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
        assert abs(a - refs["f64"][0][k]) <= 2e-5 * max(abs(b), 1e-3), (k, a, refs["f64"][0][k])
    _assert_gradients_tight(model, refs, "f32 mode RepPoints", real_rows={"offsets_init.1": 18, "offsets_refine": 18}, min_checked=41)


@pytest.mark.parametrize("rotated", [False, True])
def test_rcnn_step_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode, rotated):
    """BASELINE configs[4]'s step family (GeneralizedRCNN: RPN / RRPN + StandardROIHeads / RROIHeads over ROIAlign / ROIAlignRotated) in
    the validation mode against oracle.rcnn.OracleRCNN in fp32 and float64, with the random anchor / proposal samples and the proposals
    taken from the run under test (they are checked against the oracle's matchers in test_gpu_rcnn.py).  The four losses to 2e-5 and every
    parameter gradient to 1e-4 of its norm against the float64 oracle (measured 3e-6 .. 7e-6)."""
    from oracle import rcnn as orc
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer
    from test_gpu_rcnn import _cfg, _cpu, _data

    cfg = _cfg(rotated)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = _data(2, 96, 128, 21, rotated)
    got, masks = _tapped_step(model, opt, data)
    rpn, roi = model.proposal_generator, model.roi_heads
    D = 5 if rotated else 4
    gt_labels, _, gt_deltas = (t.cpu() for t in rpn.last_targets)
    props = roi.last_proposals
    rois = torch.cat([torch.cat((torch.full((len(p), 1), float(i)), p.proposal_boxes.tensor.cpu()), 1) for i, p in enumerate(props)])
    roi_cls = torch.cat([p.gt_classes.cpu() for p in props])
    roi_gtb = torch.cat([p.gt_boxes.tensor.cpu() for p in props])
    # (one oracle pass, in float64: the oracle's ROI pooling is a pure-Python loop over every sample point - a second pass in fp32 costs
    # the GPU box half a minute and, with hip32 - cpu32 at 3e-6 .. 7e-6 when it was measured, arbitrates nothing)
    oracle = orc.OracleRCNN.from_hip_model(model, emulate_bf16=False).double()
    with _forced(masks) as st:          # the product's ReLU decisions (backbone, RPN conv, the two FC layers); see the FCOS test
        r = oracle.losses(_cpu(data), gt_labels, gt_deltas.double(), rois.double(), roi_cls, roi_gtb.double())
        tr = oracle.trainable()
        ref_grads = dict(zip(tr.keys(), torch.autograd.grad(sum(r.values()), list(tr.values()), allow_unused=True)))
    assert not st["missed"], st["missed"][:5]
    ref_losses = {k: float(v.detach()) for k, v in r.items()}
    for k, b in ref_losses.items():
        a = float(got[k].detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
    K, A = 80, rpn.head.num_anchors
    _assert_gradients_tight(model, {"f64": (ref_losses, ref_grads)}, f"f32 mode R-CNN rotated={rotated}",
                            real_rows={"objectness_logits": A, "anchor_deltas": A * D, "cls_score": K + 1, "bbox_pred": K * D}, min_checked=31)


@pytest.mark.parametrize("which", ["pointset", "lrtb", "anchor"])
def test_ablation_heads_in_f32_mode_vs_oracle(cuda, f32mode, which):
    """AblationMetaArch with PointSetHead / LRTBHead / AnchorHead ("Supervised Offset" feature adaption: the
    DeformConv fed by the head's own init prediction) in the validation mode.  The oracle heads (oracle/{pointset,lrtb,anchor_head}.py,
    pinned to the reference's own heads run on CPU by tests/test_oracle_*.py) get the product's FPN features and must reproduce the
    losses to 2e-5 and the gradient of every head parameter to 1e-4 of its norm (measured 2.5e-6 / 4.3e-6 / 7.7e-6; the bf16 product path
    holds 3e-2 on the forward alone)."""
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    if which == "pointset":
        from oracle import pointset as om
        from test_gpu_pointset import _cfg

        cfg, seed, dseed, Oracle = _cfg("Supervised Offset"), 4, 12, om.OraclePointSetHead
    elif which == "lrtb":
        from oracle import lrtb as om
        from test_gpu_lrtb import _cfg

        cfg, seed, dseed, Oracle = _cfg("Supervised Offset", False, True, True, "giou", True, 1.5), 9, 14, om.OracleLRTBHead
    else:
        from oracle import anchor_head as om
        from test_gpu_anchor_head import _cfg

        cfg, seed, dseed, Oracle = _cfg("supervised", "smooth_l1"), 14, 17, om.OracleAnchorHead
    torch.manual_seed(seed)
    model = build_model(cfg)
    model.train()
    head = model.head
    with torch.no_grad():      # the same away-from-degenerate-init tweaks as the bf16 tests of each head
        if which == "pointset":
            for m in (head.loc_init_out.conv, head.offsets_refine):
                m.weight.mul_(6.0)
        elif which == "lrtb":
            head.loc_init_out.conv.bias[:4].fill_(0.75)
            head.box_pred.conv.bias[:4].fill_(0.75)
            head.scales_init.add_(torch.linspace(-0.2, 0.2, 5, device=head.scales_init.device))
        else:
            head.loc_init_out.conv.weight.mul_(5.0)
            head.bbox_pred.weight.mul_(5.0)
    model.arena.bump()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 192, 256, dseed, device="cuda")
    got = model(data)
    total = sum(got.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f].float().cpu().permute(0, 3, 1, 2).contiguous() for f in head.in_features]
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    o = Oracle.from_hip_head(head, emulate_bf16=False)
    if which == "anchor":
        ref, _ = o.losses(feats, gtb, gtc, [tuple(d["image"].shape[-2:]) for d in data])
    else:
        ref = o.losses(feats, gtb, gtc)
    assert set(ref) == set(got)
    for k, v in ref.items():
        a, b = float(got[k].detach()), float(v.detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
    names = [k for k, v in o.p.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(sum(ref.values()), [o.p[k] for k in names], allow_unused=True)))
    mine = dict(head.named_parameters())
    worst, checked = 0.0, 0
    for k, r in grads.items():
        if r is None or k not in mine or mine[k].grad is None:
            continue
        g = mine[k].grad.detach().float().cpu()
        if g.dim() == 4:
            g = g.permute(0, 3, 1, 2)
        assert g.shape == r.shape, (k, g.shape, r.shape)
        d = (g - r).norm().item() / max(r.norm().item(), 1e-12)
        worst = max(worst, d)
        assert d <= 1e-4, (k, d)
        checked += 1
    assert checked >= 20, checked
    print(f"\nf32 mode AblationMetaArch {which}: {checked} head parameter gradients, worst relative distance to the fp32 oracle head {worst:.2e}")


@pytest.mark.parametrize("where", ["tower_v1", "tower_v2", "backbone", "backbone_v2"])
def test_fcos_with_deformable_convs_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode, where):
    """The DCN configurations of FCOS (configs/fcos/*dcn*.yaml: MODEL.FCOS.USE_DCN_IN_TOWER with DeformConv / ModulatedDeformConv as the
    last tower conv; MODEL.RESNETS.DEFORM_ON_PER_STAGE = detectron2's DeformBottleneckBlock with FrozenBN folded into the deformable conv)
    in the validation mode: one step of FCOS R18 / R50 against the fp32 and float64 oracles - losses to 2e-5 (5e-5 with the backbone
    blocks: offsets of several pixels at random initialisation put samples next to pixel boundaries), every gradient within 1e-4 (2e-4
    with the backbone blocks) of the float64 and the CPU fp32 oracle on the product's ReLU decisions (see the plain FCOS test).
    ``backbone_v2`` = DEFORM_MODULATED (configs/fcos/fcos_R_50_FPN_2x_dcnv2.yaml), back in this file since round 4."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    backbone = where.startswith("backbone")
    cfg = make_cfg(50 if backbone else 18)
    if backbone:
        cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE = [False, True, True, True]
        cfg.MODEL.RESNETS.DEFORM_MODULATED = where == "backbone_v2"      # configs/fcos/fcos_R_50_FPN_2x_dcnv2.yaml
    else:
        cfg.MODEL.FCOS.USE_DCN_IN_TOWER = True
        cfg.MODEL.FCOS.USE_DCN_V2 = where == "tower_v2"
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    if backbone:            # offsets of a fraction of a pixel: at random initialisation the offset convs see activations of ~1e2
        from slenderobjdet_amd.modeling.backbone.resnet import DeformBottleneckBlock
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, DeformBottleneckBlock):      # detectron2 zero-initialises the offset conv: give it something to sample with
                    m.conv2_offset.weight.normal_(0.0, 1e-4)
                    m.conv2_offset.weight[m.n_off:].zero_()
        model.arena.bump()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 192, 256, 3, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    got, masks = _tapped_step(model, opt, data)
    refs, missed = _forced_oracle_grads(lambda: OracleFCOS.from_hip_model(model, emulate_bf16=False), lambda o: o.losses(cpu), masks)
    assert not missed, missed[:5]
    ltol = 5e-5 if backbone else 2e-5
    for k, b in refs["f64"][0].items():
        a = float(got[k].detach())
        assert abs(a - b) <= ltol * max(abs(b), 1e-3), (k, a, b)
    # every ReLU - the DeformConv + FrozenBN + ReLU of DeformBottleneckBlock and GroupNorm + ReLU behind DFConv2d included - takes the
    # product's decisions in the oracle; what is left between the runs is summation order and the bilinear sampling arithmetic
    # (unmodulated backbone blocks: the CPU fp32 oracle itself sits 1.4e-4 from float64 there - offsets of several pixels put samples next
    # to pixel boundaries, where the bilinear weights cancel - and the HIP run at 1.3e-4; the modulated variant: 7e-6 like everything else)
    v1_backbone = where == "backbone"
    _assert_gradients_tight(model, refs, f"f32 mode FCOS DCN {where}", max_pair=5e-4 if v1_backbone else 1e-4, share_2e5=0.2 if v1_backbone else 0.9)


def test_fcos_resnext_in_f32_mode_matches_the_fp32_oracle(cuda, f32mode):
    """ResNeXt bottlenecks (NUM_GROUPS 32 / WIDTH_PER_GROUP 8 / STRIDE_IN_1X1 false, depth 50; the grouped 3x3 as the block-diagonal
    embedding of its weight) in the validation mode: losses to 2e-5, gradients (the grouped convs' in the reference's (K, C / 32, 3, 3)
    shape) within 1e-4 of the float64 and the CPU fp32 oracle on the product's ReLU decisions (see the plain FCOS test)."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = make_cfg(50)
    cfg.MODEL.RESNETS.NUM_GROUPS, cfg.MODEL.RESNETS.WIDTH_PER_GROUP, cfg.MODEL.RESNETS.STRIDE_IN_1X1 = 32, 8, False
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 128, 192, 3, device="cuda")
    cpu = [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]
    got, masks = _tapped_step(model, opt, data)
    refs, missed = _forced_oracle_grads(lambda: OracleFCOS.from_hip_model(model, emulate_bf16=False), lambda o: o.losses(cpu), masks)
    assert not missed, missed[:5]
    for k, b in refs["f64"][0].items():
        a = float(got[k].detach())
        assert abs(a - b) <= 2e-5 * max(abs(b), 1e-3), (k, a, b)
    grouped = sum(int(g is not None and g.dim() == 4 and g.shape[1] * 32 == g.shape[0] and g.shape[2] == 3) for g in refs["f64"][1].values())
    assert grouped >= 10
    _assert_gradients_tight(model, refs, "f32 mode FCOS ResNeXt-50 32x8d")


def test_bf16_only_devices_refuse_in_f32_mode(cuda, f32mode):
    """No silent precision mixing: what exists only on the bf16 product path raises in the validation mode."""
    from slenderobjdet_amd._C import SlenderHipError

    HF = f32mode
    x = torch.zeros(1, 8, 8, 8, device=cuda)
    w = torch.zeros(8, 1, 1, 8, device=cuda)
    with pytest.raises(SlenderHipError):
        HF.conv2d_fwd(x, w, relu_bits=torch.zeros(64, dtype=torch.uint8, device=cuda))
    with pytest.raises(SlenderHipError):
        HF.conv2d_fwd(x.bfloat16(), w)
    with pytest.raises(SlenderHipError):
        HF.conv_gn_fwd_ml([x], w, None, None, None, 1)
