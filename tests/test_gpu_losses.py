"""GPU parity: loss / target-assignment / normalisation kernels (through the C ABI) vs the CPU oracle.

Tolerance stated by BASELINE.json north_star: losses within 1e-3 relative (fp32); integer outputs (labels) bit-exact.
We hold the fp32 kernels to 1e-5 relative (they differ from the oracle only by reduction order and libm ulps).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import fcos_targets as ot
from oracle import losses as ol
from oracle import nn as onn

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rel(got, ref, tol, what):
    got = got.detach().float().cpu().reshape(-1)
    ref = ref.detach().float().reshape(-1)
    err = (got - ref).abs().max().item()
    lim = tol * max(ref.abs().max().item(), 1e-6)
    assert err <= lim, f"{what}: err {err:.4g} > {lim:.4g}"


@pytest.mark.parametrize("M,K", [(1, 1), (37, 80), (5000, 80), (22400 * 2, 80), (300, 3)])
@pytest.mark.parametrize("alpha,gamma", [(0.25, 2.0), (-1.0, 2.0), (0.4, 1.5)])
def test_focal_labels(cuda, M, K, alpha, gamma):
    from slenderobjdet_amd.layers import functional as HF

    x = torch.randn(M, K, generator=_g(0)) * 3
    labels = torch.randint(0, K + 1, (M,), generator=_g(1)).int()   # K == background
    labels[::7] = K
    onehot = ol.one_hot_from_labels(labels.long(), K)
    xr = x.clone().requires_grad_(True)
    ref = ol.sigmoid_focal_loss(xr, onehot, alpha, gamma, "sum")
    (gref,) = torch.autograd.grad(ref, xr)
    s, elem = HF.focal_loss_fwd(x.to(cuda), labels.to(cuda), None, alpha, gamma, want_elem=True)
    _rel(s, ref, 1e-5, "focal sum")
    _rel(elem, ol.sigmoid_focal_loss(x, onehot, alpha, gamma, "none"), 1e-5, "focal elem")
    g = HF.focal_loss_bwd(x.to(cuda), labels.to(cuda), None, alpha, gamma)
    _rel(g, gref, 1e-5, "focal grad")
    # bf16 padded gradient rows with device-side normaliser (training path)
    num = torch.tensor([2.0], device=cuda)
    den = torch.tensor([6.0], device=cuda)
    ld_out = (K + 7) // 8 * 8
    gb = HF.focal_loss_bwd(x.to(cuda), labels.to(cuda), None, alpha, gamma, scale_num=num, scale_den=den, den_mul=0.5,
                           den_min=1.0, ld_out=ld_out, out_bf16=True)
    _rel(gb[:, :K], gref * (2.0 / 3.0), 2 ** -7, "focal grad bf16")
    assert (gb[:, K:] == 0).all()


def test_focal_dense_targets(cuda):
    from slenderobjdet_amd.layers import functional as HF

    x = torch.randn(200, 17, generator=_g(0)) * 2
    t = (torch.rand(200, 17, generator=_g(1)) > 0.8).float()
    ref = ol.sigmoid_focal_loss(x, t, 0.25, 2.0, "sum")
    s, _ = HF.focal_loss_fwd(x.to(cuda), None, t.to(cuda), 0.25, 2.0)
    _rel(s, ref, 1e-5, "focal dense")


@pytest.mark.parametrize("loss_type", ["iou", "linear_iou", "giou"])
@pytest.mark.parametrize("P", [1, 7, 1000])
def test_iou_loss(cuda, loss_type, P):
    from slenderobjdet_amd.layers import functional as HF

    pred = torch.rand(P, 4, generator=_g(0)) * 50 + 1
    tgt = torch.rand(P, 4, generator=_g(1)) * 50 + 1
    w = torch.rand(P, generator=_g(2))
    if P > 3:
        pred[1] = tgt[1]          # exact ties exercise the min/max half-gradient rule
        pred[2, 0] = tgt[2, 0]
    pr = pred.clone().requires_grad_(True)
    ref = ol.iou_loss_ltrb(pr, tgt, w, loss_type)
    (gref,) = torch.autograd.grad(ref, pr)
    s, elem = HF.iou_loss_fwd(pred.to(cuda), tgt.to(cuda), w.to(cuda), loss_type, want_elem=True)
    _rel(s, ref, 1e-5, "iou sum")
    _rel(elem, ol.iou_loss_ltrb(pred, tgt, w, loss_type, reduce=False), 1e-5, "iou elem")
    g = HF.iou_loss_bwd(pred.to(cuda), tgt.to(cuda), w.to(cuda), loss_type)
    _rel(g, gref, 2e-5, "iou grad")
    # unweighted + masked
    mask = (torch.arange(P) % 3 != 0).int()
    s2, _ = HF.iou_loss_fwd(pred.to(cuda), tgt.to(cuda), None, loss_type, mask=(mask * 5).to(cuda), mask_bg=0)
    _rel(s2, ol.iou_loss_ltrb(pred[mask.bool()], tgt[mask.bool()], None, loss_type) if mask.sum() else torch.zeros(()), 1e-5, "iou masked")


def _boxes(n_img, seed, hw=(160, 224)):
    g = _g(seed)
    out_b, out_c = [], []
    for i in range(n_img):
        G = int(torch.randint(1, 9, (1,), generator=g))
        cx = torch.rand(G, generator=g) * hw[1]
        cy = torch.rand(G, generator=g) * hw[0]
        w = torch.exp(torch.rand(G, generator=g) * 4 + 1.5)
        h = torch.exp(torch.rand(G, generator=g) * 4 + 1.5)
        b = torch.stack([(cx - w / 2).clamp(0, hw[1]), (cy - h / 2).clamp(0, hw[0]), (cx + w / 2).clamp(0, hw[1]), (cy + h / 2).clamp(0, hw[0])], 1)
        out_b.append(b)
        out_c.append(torch.randint(0, 80, (G,), generator=g))
    return out_b, out_c


@pytest.mark.parametrize("radius", [0.0, 1.5])
def test_fcos_assign(cuda, radius):
    from slenderobjdet_amd.layers import functional as HF

    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    strides = [8, 16, 32, 64, 128]
    boxes, classes = _boxes(3, 5)
    boxes[1][0] = torch.tensor([-10.0, 4.0, 10.0, 60.0]) if radius > 0 else boxes[1][0]   # first box centred at x == 0: reference quirk
    ref_l, ref_r = ot.targets_for_batch(hw, strides, boxes, classes, radius, 80)
    offs = torch.tensor([0] + [len(b) for b in boxes]).cumsum(0).int()
    lab, reg, ctr, stats = HF.fcos_assign(torch.cat(boxes).to(cuda), torch.cat(classes).int().to(cuda), offs.to(cuda), 3, hw, strides,
                                          ot.SIZES_OF_INTEREST, radius, 80)
    assert torch.equal(lab.cpu().long(), ref_l), "labels must be bit-exact"
    assert torch.equal(reg.cpu(), ref_r), "regression targets must be bit-exact"
    fg = ref_l != 80
    ctr_ref = torch.zeros_like(ref_l, dtype=torch.float32)
    ctr_ref[fg] = ol.centerness_targets(ref_r[fg])
    _rel(ctr, ctr_ref, 1e-6, "centerness targets")
    _rel(stats, torch.stack([fg.sum().float(), ctr_ref.sum()]), 1e-5, "stats")
    assert fg.sum() > 0 or radius > 0


def test_fcos_assign_empty_image(cuda):
    from slenderobjdet_amd.layers import functional as HF

    hw, strides = [(4, 4), (2, 2)], [8, 16]
    offs = torch.tensor([0, 0, 1]).int()
    boxes = torch.tensor([[2.0, 2.0, 30.0, 30.0]])
    lab, reg, ctr, stats = HF.fcos_assign(boxes.to(cuda), torch.tensor([3]).int().to(cuda), offs.to(cuda), 2, hw, strides,
                                          [[-1, 64], [64, 128]], 0.0, 80)
    assert (lab[0] == 80).all() and (reg[0] == 0).all()
    ref_l, ref_r = ot.targets_for_batch(hw, strides, [boxes[:0], boxes], [torch.zeros(0).long(), torch.tensor([3])], 0.0, 80)
    assert torch.equal(lab.cpu().long(), ref_l)


@pytest.mark.parametrize("loss_type,norm_reg", [("giou", False), ("iou", False), ("linear_iou", True)])
def test_fcos_regctr_loss(cuda, loss_type, norm_reg):
    """Fused Scale+exp + iou_loss*centerness + centerness BCE over all locations vs oracle composition."""
    from slenderobjdet_amd.layers import functional as HF

    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    strides = [8, 16, 32, 64, 128]
    N = 2
    L = sum(h * w for h, w in hw)
    boxes, classes = _boxes(N, 11)
    labels, reg_t = ot.targets_for_batch(hw, strides, boxes, classes, 1.5, 80)
    if norm_reg:
        pass
    raw = torch.randn(N * L, 8, generator=_g(0)) * 0.5 + (1.0 if not norm_reg else 0.2)
    scales = torch.tensor([1.0, 0.9, 1.1, 1.2, 0.8])
    lvl_of = torch.cat([torch.full((h * w,), i) for i, (h, w) in enumerate(hw)]).repeat(N)
    st_of = torch.tensor(strides, dtype=torch.float32)[lvl_of]
    rawr = raw.clone().requires_grad_(True)
    sc = scales.clone().requires_grad_(True)
    z = rawr[:, :4] * sc[lvl_of][:, None]
    pred = torch.relu(z) * st_of[:, None] if norm_reg else torch.exp(z)
    lab, rt = labels.reshape(-1), reg_t.reshape(-1, 4)
    fg = lab != 80
    ctr_t = ol.centerness_targets(rt[fg])
    reg_sum = ol.iou_loss_ltrb(pred[fg], rt[fg], ctr_t, loss_type)
    ctr_sum = F.binary_cross_entropy_with_logits(rawr[:, 4][fg], ctr_t, reduction="sum")
    npos, sctr = float(fg.sum()), float(ctr_t.sum())
    total = reg_sum / sctr * 0.7 + ctr_sum / max(npos, 1.0) * 1.3
    graw, gsc = torch.autograd.grad(total, (rawr, sc))

    d = lambda t: t.to(cuda)
    offs = torch.tensor([0] + [len(b) for b in boxes]).cumsum(0).int()
    hl, hr, hc, stats = HF.fcos_assign(d(torch.cat(boxes)), d(torch.cat(classes).int()), d(offs), N, hw, strides, ot.SIZES_OF_INTEREST, 1.5, 80)
    raw_d = d(raw)
    sums = HF.fcos_regctr_loss_fwd(raw_d, 8, raw_d.view(-1)[4:], 8, hl, hr, hc, d(scales), N, hw, strides, 80, loss_type, norm_reg)
    _rel(sums, torch.stack([reg_sum, ctr_sum]), 2e-5, "regctr sums")
    dbox = torch.full((N * L, 8), 9.0, dtype=torch.bfloat16, device=cuda)
    dsc = torch.zeros(5, device=cuda)
    HF.fcos_regctr_loss_bwd(raw_d, 8, raw_d.view(-1)[4:], 8, hl, hr, hc, d(scales), N, hw, strides, 80, loss_type, norm_reg,
                            torch.tensor([0.7], device=cuda), torch.tensor([1.3], device=cuda), stats, 1.0, dbox, 8, 4, dbox, 8, 4, dsc)
    _rel(dbox[:, :5], graw[:, :5], 2 ** -7, "regctr d(raw)")
    assert (dbox[:, 5:] == 0).all()
    _rel(dsc, gsc, 1e-4, "d(scale)")
    out3 = HF.fcos_finalize_losses(torch.tensor([5.0], device=cuda), sums, stats, 1.0)
    _rel(out3, torch.stack([torch.tensor(5.0 / max(npos, 1)), reg_sum / sctr, ctr_sum / max(npos, 1)]), 2e-5, "finalize")


@pytest.mark.parametrize("N,HW,C,G,relu", [(2, 77, 256, 32, True), (1, 1000, 256, 32, False), (3, 64, 64, 8, True), (2, 300, 128, 4, True),
                                           # the backward passes' LDS landing buffers at their edges: fewer pixels than one trip covers, one row per
                                           # wave (512 channels), one row per BLOCK (2048 channels), many blocks per image
                                           (2, 5, 256, 32, True), (1, 333, 512, 32, True), (2, 40, 2048, 32, False), (2, 9000, 256, 32, True)])
def test_groupnorm(cuda, N, HW, C, G, relu):
    from slenderobjdet_amd.layers import functional as HF

    x = onn.rb(torch.randn(N, HW, 1, C, generator=_g(0)) * 2 + 0.3)
    gamma = torch.rand(C, generator=_g(1)) + 0.5
    beta = torch.randn(C, generator=_g(2)) * 0.2
    dy = onn.rb(torch.randn(N, HW, 1, C, generator=_g(3)))
    ref = onn.group_norm(x, gamma, beta, G, 1e-5, relu)
    y, stats = HF.groupnorm_fwd(x.to(cuda).bfloat16(), gamma.to(cuda), beta.to(cuda), G, 1e-5, relu)
    _rel(y, ref, 2 ** -7, "gn fwd")
    dx_ref, dg_ref, db_ref = onn.group_norm_backward(x, gamma, beta, G, dy, 1e-5, relu)
    dg = torch.zeros(C, device=cuda)
    db = torch.zeros(C, device=cuda)
    dsum = torch.zeros(C, device=cuda)
    dx = HF.groupnorm_bwd(dy.to(cuda).bfloat16(), x.to(cuda).bfloat16(), gamma.to(cuda), beta.to(cuda), stats, G, dg, db, relu, dxsum=dsum)
    _rel(dx, dx_ref, 2 ** -6, "gn dx")
    _rel(dsum, dx.float().sum(dim=(0, 1, 2)).cpu(), 1e-4, "fused bias gradient (sum of dx)")
    _rel(dg, dg_ref, 2e-3, "gn dgamma")
    _rel(db, db_ref, 2e-3, "gn dbeta")


def test_elementwise(cuda):
    from slenderobjdet_amd.layers import functional as HF

    x = onn.rb(torch.randn(2, 9, 11, 64, generator=_g(0)))
    dy = onn.rb(torch.randn(2, 9, 11, 64, generator=_g(1)))
    xd, dyd = x.to(cuda).bfloat16(), dy.to(cuda).bfloat16()
    assert torch.equal(HF.relu_fwd(xd).float().cpu(), torch.relu(x))
    assert torch.equal(HF.relu_bwd(dyd, xd).float().cpu(), dy * (x > 0))
    assert torch.equal(HF.add_bf16(xd, dyd).float().cpu(), onn.rb(x + dy))
    assert torch.equal(HF.maxpool3x3s2(xd).float().cpu(), onn.max_pool_3x3_s2(x))
    g = onn.rb(torch.randn(2, 8, 12, 64, generator=_g(2)))
    _rel(HF.upsample2x_bwd(g.to(cuda).bfloat16()), onn.upsample2x_backward(g), 2 ** -7, "upsample bwd")
    db = torch.zeros(64, device=cuda)
    HF.bias_grad(dyd, db, 2, 99, 64)
    _rel(db, dy.sum(dim=(0, 1, 2)), 1e-5, "bias grad")
    dy80 = onn.rb(torch.randn(3, 50, 80, generator=_g(4)))
    db80 = torch.zeros(80, device=cuda)
    HF.bias_grad(dy80.to(cuda).bfloat16(), db80, 3, 50, 80)
    _rel(db80, dy80.sum(dim=(0, 1)), 1e-5, "bias grad C=80")


def test_preprocess(cuda):
    from slenderobjdet_amd.layers import functional as HF

    img = torch.randint(0, 256, (3, 37, 53), dtype=torch.uint8, generator=_g(0))
    mean, std = [103.53, 116.28, 123.675], [1.0, 1.0, 57.0]
    out = torch.empty((64, 64, 8), dtype=torch.bfloat16, device=cuda)
    HF.preprocess_image(img.to(cuda), out, mean, std)
    ref = onn.rb(onn.preprocess(img, mean, std, 64, 64))
    assert torch.equal(out[..., :3].float().cpu(), ref)
    assert (out[..., 3:] == 0).all()
    HF.preprocess_image(img.float().to(cuda), out, mean, std)
    assert torch.equal(out[..., :3].float().cpu(), ref)


def test_preprocess_batch_ragged(cuda):
    """One launch for the whole batch (sod_preprocess_batch): images of different sizes, bit-equal to the oracle per image."""
    from slenderobjdet_amd.layers import functional as HF

    sizes = [(37, 53), (64, 40), (1, 1), (50, 64)]
    imgs = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=_g(10 + i)) for i, (h, w) in enumerate(sizes)]
    mean, std = [103.53, 116.28, 123.675], [1.0, 1.0, 57.0]
    for cast in (lambda t: t, lambda t: t.float()):
        out = torch.full((len(imgs), 64, 64, 8), 7.0, dtype=torch.bfloat16, device=cuda)
        HF.preprocess_batch([cast(i).to(cuda) for i in imgs], out, mean, std)
        for i, im in enumerate(imgs):
            ref = onn.rb(onn.preprocess(im, mean, std, 64, 64))
            assert torch.equal(out[i, ..., :3].float().cpu(), ref), i
        assert (out[..., 3:] == 0).all()


def test_reference_signature_wrappers(cuda):
    """layers.losses.{sigmoid_focal_loss_jit, iou_loss} keep the reference signatures and are differentiable."""
    from slenderobjdet_amd.layers.losses import iou_loss, sigmoid_focal_loss_jit

    x = torch.randn(300, 80, generator=_g(0))
    labels = torch.randint(0, 81, (300,), generator=_g(1))
    onehot = ol.one_hot_from_labels(labels, 80)
    xr = x.clone().requires_grad_(True)
    ref = ol.sigmoid_focal_loss(xr, onehot, 0.25, 2.0, "sum") / 7.0
    (gref,) = torch.autograd.grad(ref, xr)
    for tgt in (onehot.to(cuda), labels.to(cuda)):
        xd = x.to(cuda).requires_grad_(True)
        out = sigmoid_focal_loss_jit(xd, tgt, alpha=0.25, gamma=2.0, reduction="sum") / 7.0
        out.backward()
        _rel(out, ref, 1e-5, "focal wrapper")
        _rel(xd.grad, gref, 1e-5, "focal wrapper grad")
    _rel(sigmoid_focal_loss_jit(x.to(cuda), onehot.to(cuda), 0.25, 2.0, "mean"), ol.sigmoid_focal_loss(x, onehot, 0.25, 2.0, "mean"), 1e-5, "mean")
    pred = torch.rand(50, 4, generator=_g(2)) * 20 + 1
    tgt = torch.rand(50, 4, generator=_g(3)) * 20 + 1
    w = torch.rand(50, generator=_g(4))
    pr = pred.clone().requires_grad_(True)
    ref = ol.iou_loss_ltrb(pr, tgt, w, "giou") * 0.5
    (gref,) = torch.autograd.grad(ref, pr)
    pd = pred.to(cuda).requires_grad_(True)
    out = iou_loss(pd, tgt.to(cuda), w.to(cuda), loss_type="giou") * 0.5
    out.backward()
    _rel(out, ref, 1e-5, "iou wrapper")
    _rel(pd.grad, gref, 2e-5, "iou wrapper grad")
    with pytest.raises(NotImplementedError):
        iou_loss(pd, tgt.to(cuda), None, loss_type="diou")


def test_bias_grad_multi_level(cuda):
    """sod_bias_grad_ml (one launch over all levels of a shared conv) against per-level torch sums."""
    from slenderobjdet_amd.layers import functional as HF

    g = torch.Generator().manual_seed(5)
    N, C = 3, 256
    dys = [(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16).to(cuda) for h, w in ((23, 37), (12, 19), (6, 10), (3, 5), (2, 3))]
    db = torch.zeros(C, device=cuda)
    HF.bias_grad_ml(dys, db)
    torch.cuda.synchronize()
    ref = sum(t.float().sum(dim=(0, 1, 2)) for t in dys)
    assert (db - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-4
