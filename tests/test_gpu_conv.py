"""GPU parity: HIP implicit-GEMM convolution (through the C ABI) vs the CPU oracle (F.conv2d / autograd, fp32).

Operands are rounded to bf16 before BOTH paths, so the only differences are fp32 accumulation order and (for bf16
outputs) the final rounding.  Tolerances: fp32 output 2e-4 * max|ref| ; bf16 output 2^-7 * max|ref|
(one bf16 ulp at the top of the range is 2^-8 relative).
"""
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return onn.rb(torch.randn(shape, generator=g) * scale)


def _close(got, ref, rel, what):
    got = got.float().cpu()
    tol = rel * max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol, f"{what}: max abs err {err:.4g} > tol {tol:.4g} (ref max {ref.abs().max().item():.4g})"


# (N, H, W, C, K, R, stride, pad, dil)
FWD_CASES = [
    (2, 13, 21, 64, 128, 3, 1, 1, 1),
    (1, 9, 11, 256, 64, 1, 1, 0, 1),
    (2, 14, 18, 64, 64, 3, 2, 1, 1),
    (2, 16, 12, 128, 256, 1, 2, 0, 1),
    (2, 32, 40, 8, 64, 7, 2, 3, 1),      # stem geometry (C padded 3 -> 8): generic contraction path
    (1, 12, 10, 80, 256, 3, 1, 1, 1),    # C % 64 != 0
    (2, 11, 13, 256, 80, 3, 1, 1, 1),    # cls_logits
    (2, 11, 13, 256, 8, 3, 1, 1, 1),     # bbox_pred+centerness (padded to 8)
    (1, 7, 9, 64, 5, 3, 1, 1, 1),        # Nout % 4 != 0: scalar epilogue
    (1, 10, 10, 64, 64, 3, 1, 2, 2),     # dilation
    (3, 25, 42, 256, 256, 3, 1, 1, 1),   # P5-sized level, 2 q-tiles x many p-tiles
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("out_f32", [True, False])
def test_conv_fwd(cuda, case, out_f32):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 1)
    w = _rand((K, R, R, C), 2, 0.05)
    b = torch.randn(K, generator=torch.Generator().manual_seed(3))
    ref = onn.conv2d(x, w, b, st, pad, dil)
    y = HF.conv2d_fwd(x.to(cuda).bfloat16(), w.to(cuda).bfloat16(), b.to(cuda), stride=st, pad=pad, dil=dil, out_f32=out_f32)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    _close(y, ref, 2e-4 if out_f32 else 2 ** -7, f"conv fwd {case}")


def test_conv_fwd_epilogue(cuda):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = 2, 12, 20, 64, 128
    x, w = _rand((N, H, W, C), 4), _rand((K, 3, 3, C), 5, 0.05)
    b = torch.randn(K, generator=torch.Generator().manual_seed(6))
    res = _rand((N, H, W, K), 7)
    ref = onn.conv2d(x, w, b, 1, 1, 1, res=res, relu=True)
    y = HF.conv2d_fwd(x.to(cuda).bfloat16(), w.to(cuda).bfloat16(), b.to(cuda), res=res.to(cuda).bfloat16(), stride=1, pad=1, relu=True)
    _close(y, ref, 2 ** -7, "bias+res+relu")
    assert (y.float() >= 0).all()
    # FPN top-down: lateral 1x1 + nearest-2x upsampled coarser map
    w1 = _rand((K, 1, 1, C), 8, 0.1)
    prev = _rand((N, H // 2, W // 2, K), 9)
    ref = onn.conv2d(x, w1, b, 1, 0, 1, res=prev, res_up2=True)
    y = HF.conv2d_fwd(x.to(cuda).bfloat16(), w1.to(cuda).bfloat16(), b.to(cuda), res=prev.to(cuda).bfloat16(), res_up2=True, out_f32=True)
    _close(y, ref, 2e-4, "lateral + upsample2x residual")


def test_conv_fwd_into_concat_buffer(cuda):
    """Per-level head outputs land directly in the (N, sum HW, K) buffer (replaces permute_and_concat)."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 64, 80
    hw = [(8, 12), (4, 6), (2, 3)]
    L = sum(h * w for h, w in hw)
    w = _rand((K, 3, 3, C), 10, 0.05)
    buf = torch.full((N, L, K), 7.0, dtype=torch.float32, device=cuda)
    refs, off = [], 0
    for i, (h, ww) in enumerate(hw):
        x = _rand((N, h, ww, C), 20 + i)
        refs.append(onn.conv2d(x, w, None, 1, 1, 1).reshape(N, h * ww, K))

        view = buf.view(-1)[off * K:]
        HF.conv2d_fwd(x.to(cuda).bfloat16(), w.to(cuda).bfloat16(), None, stride=1, pad=1, out_f32=True, out=view, y_img_stride=L * K)
        off += h * ww
    _close(buf, torch.cat(refs, dim=1), 2e-4, "concat buffer")


DGRAD_CASES = [
    (2, 13, 21, 64, 128, 3, 1, 1, 1),
    (1, 9, 11, 256, 64, 1, 1, 0, 1),
    (2, 14, 18, 64, 64, 3, 2, 1, 1),
    (2, 16, 12, 128, 256, 1, 2, 0, 1),
    (2, 15, 13, 64, 128, 3, 2, 1, 1),    # odd input with stride 2
    (2, 11, 13, 256, 80, 3, 1, 1, 1),    # dy has 80 channels: generic contraction
    (2, 11, 13, 256, 8, 3, 1, 1, 1),
    (1, 10, 10, 64, 64, 3, 1, 2, 2),
    (3, 25, 42, 256, 256, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad(cuda, case):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 1)
    w = _rand((K, R, R, C), 2, 0.05)
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, dil)
    dy = _rand((N, Ho, Wo, K), 3)
    dx_ref, _ = onn.conv2d_backward(x, w, dy, st, pad, dil)
    _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
    dx = HF.conv2d_dgrad(dy.to(cuda).bfloat16(), wt, (H, W), st, pad, dil)
    _close(dx, dx_ref, 2 ** -7, f"dgrad {case}")


def test_conv_dgrad_accum_mask(cuda):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = 2, 10, 14, 64, 64
    x, w, dy = _rand((N, H, W, C), 1), _rand((K, 3, 3, C), 2, 0.05), _rand((N, H, W, K), 3)
    acc, act = _rand((N, H, W, C), 4), _rand((N, H, W, C), 5)
    dx_ref, _ = onn.conv2d_backward(x, w, dy, 1, 1, 1)
    ref = (dx_ref + acc) * (act > 0)
    _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
    dx = HF.conv2d_dgrad(dy.to(cuda).bfloat16(), wt, (H, W), 1, 1, 1, accum=acc.to(cuda).bfloat16(), relu_mask=act.to(cuda).bfloat16())
    _close(dx, ref, 2 ** -7, "dgrad accum+mask")


WGRAD_CASES = [
    (1, 5, 70, 64, 64, 3, 1, 1, 1, 0),      # Wo >= 64: incremental row tracking path, column wraps
    (3, 2, 64, 64, 128, 3, 1, 1, 1, 2),     # Ho*Wo = 128: an image wrap every second step
    (2, 3, 100, 128, 64, 1, 1, 0, 1, 0),    # 1x1, incremental path
    (2, 13, 21, 64, 128, 3, 1, 1, 1, 0),
    (2, 13, 21, 64, 128, 3, 1, 1, 1, 1),
    (1, 9, 11, 256, 64, 1, 1, 0, 1, 0),
    (2, 14, 18, 64, 64, 3, 2, 1, 1, 3),
    (2, 16, 12, 128, 256, 1, 2, 0, 1, 0),
    (2, 11, 13, 256, 80, 3, 1, 1, 1, 0),
    (2, 11, 13, 256, 8, 3, 1, 1, 1, 2),
    (1, 10, 10, 64, 64, 3, 1, 2, 2, 0),
    (3, 25, 42, 256, 256, 3, 1, 1, 1, 0),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(cuda, case):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil, splits = case
    x = _rand((N, H, W, C), 1)
    w = _rand((K, R, R, C), 2, 0.05)
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, dil)
    dy = _rand((N, Ho, Wo, K), 3)
    _, dw_ref = onn.conv2d_backward(x, w, dy, st, pad, dil)
    dw = torch.zeros((K, R, R, C), dtype=torch.float32, device=cuda)
    HF.conv2d_wgrad(dy.to(cuda).bfloat16(), x.to(cuda).bfloat16(), dw, R, R, st, pad, dil, splits=splits)
    _close(dw, dw_ref, 2e-4, f"wgrad {case}")
    # accumulation semantics: a second call doubles the buffer
    HF.conv2d_wgrad(dy.to(cuda).bfloat16(), x.to(cuda).bfloat16(), dw, R, R, st, pad, dil, splits=splits)
    _close(dw, 2 * dw_ref, 2e-4, f"wgrad accumulate {case}")


# conv_wgrad_ring.hip: G groups of four waves split one tile's pixel range and combine through LDS.  G*1000 + NSTAGE*100 + EPI*10 (atomic /
# slab epilogue; the fragment-double-buffered, four-slot and one-group variants of round 4 measured neutral and left the tree)
RING_VARIANTS = [2300, 2310]


@pytest.mark.parametrize("variant", RING_VARIANTS)
def test_conv_wgrad_ring_variants(cuda, variant):
    """Every variant of the in-workgroup split-over-pixels kernel against the oracle on every weight-gradient geometry of this file
    (incremental / division row paths, strides, dilation, partial q / c tiles, ragged last split, explicit split counts), with
    accumulation semantics, the folded FrozenBN scale, and - for the slab epilogue - bit-identical repeats."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    _C.call("sod_conv_set_wgrad_variant", variant)
    try:
        for case in WGRAD_CASES + [(2, 50, 84, 128, 256, 1, 1, 0, 1, 0), (1, 40, 66, 128, 128, 3, 1, 1, 1, 5), (2, 7, 9, 136, 72, 3, 1, 1, 1, 3)]:
            N, H, W, C, K, R, st, pad, dil, splits = case
            x = _rand((N, H, W, C), 1)
            w = _rand((K, R, R, C), 2, 0.05)
            Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, dil)
            dy = _rand((N, Ho, Wo, K), 3)
            _, dw_ref = onn.conv2d_backward(x, w, dy, st, pad, dil)
            dyd, xd = dy.to(cuda).bfloat16(), x.to(cuda).bfloat16()
            dw = torch.zeros((K, R, R, C), dtype=torch.float32, device=cuda)
            HF.conv2d_wgrad(dyd, xd, dw, R, R, st, pad, dil, splits=splits)
            _close(dw, dw_ref, 2e-4, f"ring {variant} wgrad {case}")
            first = dw.clone()
            HF.conv2d_wgrad(dyd, xd, dw, R, R, st, pad, dil, splits=splits)
            _close(dw, 2 * dw_ref, 2e-4, f"ring {variant} wgrad accumulate {case}")
            qs = torch.rand(K, generator=torch.Generator().manual_seed(5)) + 0.5
            dw3 = torch.zeros_like(dw)
            HF.conv2d_wgrad(dyd, xd, dw3, R, R, st, pad, dil, splits=splits, qscale=qs.to(cuda))
            _close(dw3, dw_ref * qs.view(-1, 1, 1, 1), 2e-4, f"ring {variant} wgrad qscale {case}")
            if (variant // 10) % 10 == 1:
                dw2 = torch.zeros_like(dw)
                HF.conv2d_wgrad(dyd, xd, dw2, R, R, st, pad, dil, splits=splits)
                assert torch.equal(dw2, first), f"ring {variant}: slab epilogue must be bit-identical from run to run {case}"
        # three levels that share the weights: the virtual pixel index crosses levels inside a group's range
        N, C, K = 2, 128, 128
        hws = [(20, 68), (10, 34), (5, 17)]
        xs = [_rand((N, h, w, C), 10 + i) for i, (h, w) in enumerate(hws)]
        dys = [_rand((N, h, w, K), 20 + i) for i, (h, w) in enumerate(hws)]
        ref = sum(onn.conv2d_backward(x, torch.zeros((K, 3, 3, C)), dy, 1, 1, 1)[1] for x, dy in zip(xs, dys))
        dw = torch.zeros((K, 3, 3, C), dtype=torch.float32, device=cuda)
        HF.conv2d_wgrad_ml([d.to(cuda).bfloat16() for d in dys], [x.to(cuda).bfloat16() for x in xs], dw, 3, 3, 1, 1, 1)
        _close(dw, ref, 2e-4, f"ring {variant} multi-level")
        # deterministic mode always takes the slab epilogue, whatever the variant says
        prev = HF.DETERMINISTIC
        HF.DETERMINISTIC = True
        try:
            outs = []
            for _ in range(2):
                dw = torch.zeros((K, 3, 3, C), dtype=torch.float32, device=cuda)
                HF.conv2d_wgrad_ml([d.to(cuda).bfloat16() for d in dys], [x.to(cuda).bfloat16() for x in xs], dw, 3, 3, 1, 1, 1)
                outs.append(dw)
        finally:
            HF.DETERMINISTIC = prev
        _close(outs[0], ref, 2e-4, f"ring {variant} deterministic")
        assert torch.equal(outs[0], outs[1])
    finally:
        _C.call("sod_conv_set_wgrad_variant", -1)


@pytest.mark.parametrize("C,groups,stride,hw", [(256, 32, 1, (13, 21)), (512, 32, 2, (14, 18)), (256, 2, 1, (9, 11)), (1024, 32, 1, (7, 9))])
def test_grouped_conv_channel_window_vs_grouped_oracle(cuda, C, groups, stride, hw):
    """Grouped 3x3 (ResNeXt: detectron2 BottleneckBlock(num_groups), configs/ablation_studies/pointset/base_X101.yaml) in CHANNEL-WINDOW
    mode - a 128-wide output tile contracts over the 128 input channels at the same offset (SOD_CONV_CWIN, sod_conv2d_dgrad_cwin,
    SOD_WGRAD_DIAG) - against F.conv2d(groups=) and its autograd: forward + folded FrozenBN bias + ReLU, data gradient with the ReLU mask
    of its input, weight gradient in the reference's (K, C / groups, 3, 3) shape, in default and deterministic mode."""
    import torch.nn.functional as F

    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.layers.nn import HipGroupedConv2d

    N, (H, W) = 2, hw
    Cg = C // groups
    m = HipGroupedConv2d(C, C, 3, stride, 1, 1, groups=groups, bias=False).to(cuda)
    assert m.windowed()
    x = _rand((N, H, W, C), 1).relu()                       # an activation: its own ReLU mask is what the data gradient applies
    w = _rand((C, 3, 3, Cg), 2, 0.05)
    b = torch.randn(C, generator=torch.Generator().manual_seed(3))
    win = m.window_weight(w.to(cuda))
    T = C // 128
    win_t = win.reshape(T, 128, 3, 3, 128).permute(0, 4, 2, 3, 1).reshape(win.shape)
    wk = HF.weight_prep(win.contiguous(), None, True, False)[0]
    wt = HF.weight_prep(win_t.contiguous(), None, True, False)[0]
    xs, ws = x.permute(0, 3, 1, 2).clone().requires_grad_(True), w.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = torch.relu(F.conv2d(xs, ws, b, stride=stride, padding=1, groups=groups))
    xd = x.to(cuda).bfloat16()
    y = HF.conv2d_fwd(xd, wk, b.to(cuda), stride=stride, pad=1, relu=True)
    _close(y.permute(0, 3, 1, 2), ref.detach(), 2 ** -7, "grouped fwd")
    Ho, Wo = ref.shape[2:]
    dy = _rand((N, Ho, Wo, C), 4)
    pre = F.conv2d(xs, ws, b, stride=stride, padding=1, groups=groups)
    gx, gw = torch.autograd.grad(pre, [xs, ws], dy.permute(0, 3, 1, 2))
    dyd = dy.to(cuda).bfloat16()
    dx = HF.conv2d_dgrad(dyd, wt, (H, W), stride, 1, 1, relu_mask=xd)
    _close(dx.permute(0, 3, 1, 2), gx * (xs.detach() > 0), 2 ** -7, "grouped dgrad + mask")
    for det in (False, True):
        prev, HF.DETERMINISTIC = HF.DETERMINISTIC, det
        try:
            dw = torch.zeros((C, 3, 3, 128), dtype=torch.float32, device=cuda)
            HF.conv2d_wgrad(dyd, xd, dw, 3, 3, stride, 1, 1)
            HF.conv2d_wgrad(dyd, xd, dw, 3, 3, stride, 1, 1)      # accumulation semantics
        finally:
            HF.DETERMINISTIC = prev
        _close(m.window_blocks(dw).permute(0, 3, 1, 2), 2 * gw, 2e-4, f"grouped wgrad det={det}")


# the 256x256 8-wave weight-gradient kernel (conv_wgrad256.hip), forced with splits = -1: (N, H, W, C, K, R, stride, pad, dil)
WGRAD256_CASES = [
    (2, 40, 72, 256, 256, 3, 1, 1, 1),     # incremental row path (Wo >= 64), 9 tiles, column and image wraps
    (2, 13, 21, 256, 256, 3, 1, 1, 1),     # division path (Wo < 64), ragged last K-tile
    (1, 9, 70, 512, 256, 1, 1, 0, 1),      # 1x1, two channel tiles
    (2, 16, 24, 256, 512, 1, 2, 0, 1),     # stride-2 1x1 (shortcut), two output-channel tiles
    (1, 10, 66, 256, 256, 3, 1, 2, 2),     # dilation on the incremental path
    (1, 3, 5, 256, 256, 3, 1, 1, 1),       # a single, mostly padded K-tile
    (3, 25, 42, 512, 512, 3, 1, 1, 1),     # res5 conv2 geometry: 36 tiles
    (2, 13, 21, 256, 720, 3, 1, 1, 1),     # RetinaNet cls_score: K = 9 x 80 = 720, the last q-tile holds 208 of its 256 rows
    (2, 9, 11, 256, 264, 1, 1, 0, 1),      # 8 rows in the last q-tile
]


@pytest.mark.parametrize("case", WGRAD256_CASES)
def test_conv_wgrad256(cuda, case):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 1)
    w = _rand((K, R, R, C), 2, 0.05)
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, dil)
    dy = _rand((N, Ho, Wo, K), 3)
    _, dw_ref = onn.conv2d_backward(x, w, dy, st, pad, dil)
    qs = torch.rand(K, generator=torch.Generator().manual_seed(5)) + 0.5
    dyd, xd = dy.to(cuda).bfloat16(), x.to(cuda).bfloat16()
    dw = torch.zeros((K, R, R, C), dtype=torch.float32, device=cuda)
    HF.conv2d_wgrad(dyd, xd, dw, R, R, st, pad, dil, splits=-1)
    _close(dw, dw_ref, 2e-4, f"wgrad256 {case}")
    first = dw.clone()
    HF.conv2d_wgrad(dyd, xd, dw, R, R, st, pad, dil, splits=-1)      # accumulation semantics
    _close(dw, 2 * dw_ref, 2e-4, f"wgrad256 accumulate {case}")
    dw2 = torch.zeros_like(dw)
    HF.conv2d_wgrad(dyd, xd, dw2, R, R, st, pad, dil, splits=-1)     # fixed summation order: bit-identical from run to run
    assert torch.equal(dw2, first)
    dw3 = torch.zeros_like(dw)
    HF.conv2d_wgrad(dyd, xd, dw3, R, R, st, pad, dil, splits=-1, qscale=qs.to(cuda))
    _close(dw3, dw_ref * qs.view(-1, 1, 1, 1), 2e-4, f"wgrad256 qscale {case}")


def test_conv_wgrad256_multilevel(cuda):
    """One launch over three levels that share the weights (the FCOS tower geometry): the virtual pixel index crosses levels and
    switches between the incremental and the division row paths."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 256, 256
    hws = [(20, 68), (10, 34), (5, 17)]
    xs = [_rand((N, h, w, C), 10 + i) for i, (h, w) in enumerate(hws)]
    dys = [_rand((N, h, w, K), 20 + i) for i, (h, w) in enumerate(hws)]
    wz = torch.zeros((K, 3, 3, C))
    ref = sum(onn.conv2d_backward(x, wz, dy, 1, 1, 1)[1] for x, dy in zip(xs, dys))
    dw = torch.zeros((K, 3, 3, C), dtype=torch.float32, device=cuda)
    HF.conv2d_wgrad_ml([d.to(cuda).bfloat16() for d in dys], [x.to(cuda).bfloat16() for x in xs], dw, 3, 3, 1, 1, 1, splits=-1)
    _close(dw, ref, 2e-4, "wgrad256 multi-level")


# the nine-taps-in-one-workgroup weight gradient (conv_wgrad9.hip), forced with splits = -2: (N, H, W, C, K)  [3x3, stride 1, pad 1]
WGRAD9_CASES = [
    (2, 40, 72, 256, 256),      # several K-tiles per block, rows longer than a K-tile, two images
    (2, 13, 21, 256, 256),      # rows shorter than a K-tile: several row / image wraps inside every tile
    (1, 3, 5, 128, 128),        # one tile, mostly padding: a single q-tile, two c-tiles
    (3, 25, 42, 512, 512),      # res5 conv2 geometry: 32 tiles, few K-tiles per block
    (2, 50, 84, 256, 256),      # res4 conv2 / P4 geometry
    (1, 100, 168, 128, 128),    # res3 conv2 geometry: the widest rows of the step (W = 168: seven X chunks of reach)
    (2, 7, 190, 64, 128),       # the widest supported row, one c-tile
    (5, 6, 7, 192, 384),        # C = 3 x 64, K = 3 x 128
    (2, 13, 21, 256, 720),      # RetinaNet cls_score: K = 9 x 80 = 720, the last q-tile holds 80 of its 128 rows
    (1, 9, 11, 64, 136),        # 8 rows in the last q-tile
]


@pytest.mark.parametrize("case", WGRAD9_CASES)
def test_conv_wgrad9(cuda, case):
    """conv_wgrad9.hip vs oracle/nn.py's autograd (F.conv2d backward, fp32): the padded pixel order (zero rows / columns instead of
    border masks), the X ring and its mirror rows, accumulate semantics, the fixed summation order and the FrozenBN scale."""
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = case
    x = _rand((N, H, W, C), 1)
    w = torch.zeros((K, 3, 3, C))
    dy = _rand((N, H, W, K), 3)
    _, dw_ref = onn.conv2d_backward(x, w, dy, 1, 1, 1)
    qs = torch.rand(K, generator=torch.Generator().manual_seed(5)) + 0.5
    dyd, xd = dy.to(cuda).bfloat16(), x.to(cuda).bfloat16()
    dw = torch.zeros((K, 3, 3, C), dtype=torch.float32, device=cuda)
    HF.conv2d_wgrad(dyd, xd, dw, 3, 3, 1, 1, 1, splits=-2)
    _close(dw, dw_ref, 2e-4, f"wgrad9 {case}")
    first = dw.clone()
    HF.conv2d_wgrad(dyd, xd, dw, 3, 3, 1, 1, 1, splits=-2)      # accumulation semantics
    _close(dw, 2 * dw_ref, 2e-4, f"wgrad9 accumulate {case}")
    dw2 = torch.zeros_like(dw)
    HF.conv2d_wgrad(dyd, xd, dw2, 3, 3, 1, 1, 1, splits=-2)     # fixed summation order: bit-identical from run to run
    assert torch.equal(dw2, first)
    dw3 = torch.zeros_like(dw)
    HF.conv2d_wgrad(dyd, xd, dw3, 3, 3, 1, 1, 1, splits=-2, qscale=qs.to(cuda))
    _close(dw3, dw_ref * qs.view(-1, 1, 1, 1), 2e-4, f"wgrad9 qscale {case}")
    # every tap separately against the 128 x 128 kernel (an error confined to one tap's shift would hide in the max-norm above)
    dw4 = torch.zeros_like(dw)
    HF.conv2d_wgrad(dyd, xd, dw4, 3, 3, 1, 1, 1, splits=1)
    for r in range(3):
        for s_ in range(3):
            _close(first[:, r, s_], dw4[:, r, s_].cpu(), 2e-4, f"wgrad9 tap ({r},{s_}) {case}")


def test_conv_wgrad9_multilevel(cuda):
    """One launch over the five FPN levels of the head towers (scaled down): blocks whose K-tile range crosses a level boundary restart
    their rings; the levels' row lengths differ (reach of 1 - 3 X chunks)."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 256, 256
    hws = [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)]
    xs = [_rand((N, h, w, C), 10 + i) for i, (h, w) in enumerate(hws)]
    dys = [_rand((N, h, w, K), 20 + i) for i, (h, w) in enumerate(hws)]
    wz = torch.zeros((K, 3, 3, C))
    ref = sum(onn.conv2d_backward(x, wz, dy, 1, 1, 1)[1] for x, dy in zip(xs, dys))
    dw = torch.zeros((K, 3, 3, C), dtype=torch.float32, device=cuda)
    HF.conv2d_wgrad_ml([d.to(cuda).bfloat16() for d in dys], [x.to(cuda).bfloat16() for x in xs], dw, 3, 3, 1, 1, 1, splits=-2)
    _close(dw, ref, 2e-4, "wgrad9 multi-level")


def test_conv_wgrad9_rejects_unsupported(cuda):
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    x = torch.zeros((1, 8, 200, 128), dtype=torch.bfloat16, device=cuda)      # W = 200: beyond the X ring's reach
    dw = torch.zeros((128, 3, 3, 128), dtype=torch.float32, device=cuda)
    with pytest.raises(_C.SlenderHipError):
        HF.conv2d_wgrad(x, x, dw, 3, 3, 1, 1, 1, splits=-2)
    x = torch.zeros((1, 8, 8, 128), dtype=torch.bfloat16, device=cuda)        # 1x1: not a 3x3
    dw = torch.zeros((128, 1, 1, 128), dtype=torch.float32, device=cuda)
    with pytest.raises(_C.SlenderHipError):
        HF.conv2d_wgrad(x, x, dw, 1, 1, 1, 0, 1, splits=-2)


def test_conv_wgrad256_rejects_unsupported(cuda):
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    dy = torch.zeros((1, 8, 8, 128), dtype=torch.bfloat16, device=cuda)      # K = 128: below one 256-row tile
    x = torch.zeros((1, 8, 8, 256), dtype=torch.bfloat16, device=cuda)
    dw = torch.zeros((128, 1, 1, 256), dtype=torch.float32, device=cuda)
    with pytest.raises(_C.SlenderHipError):
        HF.conv2d_wgrad(dy, x, dw, 1, 1, 1, 0, 1, splits=-1)


@pytest.mark.parametrize("case", [(2, 13, 21, 64, 128, 3, 1, 1, 1), (2, 11, 13, 256, 80, 3, 1, 1, 1), (2, 3, 100, 128, 64, 1, 1, 0, 1)])
def test_conv_wgrad_deterministic_mode(cuda, case):
    """SOD_WGRAD_DETERMINISTIC: the 128x128 kernel's pixel splits meet in slabs summed in a fixed order (no float atomics)."""
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 1)
    w = _rand((K, R, R, C), 2, 0.05)
    Ho, Wo = HF.conv_out_size(H, W, R, R, st, pad, dil)
    dy = _rand((N, Ho, Wo, K), 3)
    _, dw_ref = onn.conv2d_backward(x, w, dy, st, pad, dil)
    prev = HF.DETERMINISTIC
    HF.DETERMINISTIC = True
    try:
        outs = []
        for _ in range(3):
            dw = torch.zeros((K, R, R, C), dtype=torch.float32, device=cuda)
            HF.conv2d_wgrad(dy.to(cuda).bfloat16(), x.to(cuda).bfloat16(), dw, R, R, st, pad, dil)
            outs.append(dw)
    finally:
        HF.DETERMINISTIC = prev
    _close(outs[0], dw_ref, 2e-4, f"deterministic wgrad {case}")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_weight_prep(cuda):
    from slenderobjdet_amd.layers import functional as HF

    w = torch.randn(16, 3, 3, 24, generator=torch.Generator().manual_seed(0))
    sc = torch.rand(16, generator=torch.Generator().manual_seed(1)) + 0.5
    wk, wc = HF.weight_prep(w.to(cuda), sc.to(cuda))
    ref = onn.rb(w * sc.view(-1, 1, 1, 1))
    assert torch.equal(wk.float().cpu(), ref)
    assert torch.equal(wc.float().cpu(), ref.permute(3, 1, 2, 0).contiguous())


def test_conv_rejects_bad_args(cuda):
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    x = torch.zeros((1, 4, 4, 12), dtype=torch.bfloat16, device=cuda)   # C % 8 != 0
    w = torch.zeros((8, 3, 3, 12), dtype=torch.bfloat16, device=cuda)
    with pytest.raises(_C.SlenderHipError):
        HF.conv2d_fwd(x, w, None, stride=1, pad=1)
    with pytest.raises(_C.SlenderHipError):
        HF.conv2d_fwd(x.cpu(), w.cpu(), None, stride=1, pad=1)


def test_conv_multilevel_matches_per_level(cuda):
    """One multi-level launch (shared weights over FPN levels) == the per-level oracle results, fwd / dgrad / wgrad,
    including the concatenated-output form used by the prediction convs."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 64, 128
    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    w = _rand((K, 3, 3, C), 1, 0.05)
    b = torch.randn(K, generator=torch.Generator().manual_seed(2))
    xs = [_rand((N, h, ww, C), 10 + i) for i, (h, ww) in enumerate(hw)]
    dys = [_rand((N, h, ww, K), 20 + i) for i, (h, ww) in enumerate(hw)]
    wk, wt = HF.weight_prep(w.to(cuda))
    xd = [x.to(cuda).bfloat16() for x in xs]
    dyd = [d.to(cuda).bfloat16() for d in dys]
    ys = HF.conv2d_fwd_ml(xd, wk, b.to(cuda), 1, 1, 1, relu=True)
    dxs = HF.conv2d_dgrad_ml(dyd, wt, hw, 1, 1, 1)
    dw = torch.zeros((K, 3, 3, C), device=cuda)
    HF.conv2d_wgrad_ml(dyd, xd, dw, 3, 3, 1, 1, 1)
    dw_ref = torch.zeros(K, 3, 3, C)
    for x, dy, y, dx in zip(xs, dys, ys, dxs):
        _close(y, onn.conv2d(x, w, b, 1, 1, 1, relu=True), 2 ** -7, "ml fwd")
        dx_ref, dwr = onn.conv2d_backward(x, w, dy, 1, 1, 1)
        _close(dx, dx_ref, 2 ** -7, "ml dgrad")
        dw_ref += dwr
    _close(dw, dw_ref, 2e-4, "ml wgrad")
    # concatenated fp32 output + strided gradient input (prediction convs)
    L = sum(h * ww for h, ww in hw)
    offs = [sum(h * ww for h, ww in hw[:i]) for i in range(len(hw))]
    buf = torch.zeros((N, L, K), dtype=torch.float32, device=cuda)
    HF.conv2d_fwd_ml(xd, wk, None, 1, 1, 1, out_f32=True, outs=[buf.view(-1)[o * K:] for o in offs], y_img_stride=L * K)
    ref = torch.cat([onn.conv2d(x, w, None, 1, 1, 1).reshape(N, -1, K) for x in xs], 1)
    _close(buf, ref, 2e-4, "ml concat fwd")
    gbuf = torch.cat([d.reshape(N, -1, K) for d in dys], 1).to(cuda).bfloat16().contiguous()
    gviews = [gbuf.view(-1)[o * K:] for o in offs]
    dxs2 = HF.conv2d_dgrad_ml(gviews, wt, hw, 1, 1, 1, dy_img_stride=L * K, N=N)
    for dx, dx1 in zip(dxs2, dxs):
        assert torch.equal(dx, dx1)
    dw2 = torch.zeros_like(dw)
    HF.conv2d_wgrad_ml(gviews, xd, dw2, 3, 3, 1, 1, 1, dy_img_stride=L * K, K=K)
    _close(dw2, dw_ref, 2e-4, "ml concat wgrad")


@pytest.mark.parametrize("tile", [0, 2])
def test_conv_multilevel_dgrad_accum(cuda, tile):
    """sod_conv2d_dgrad_ml_accum: the data gradients of all levels + a second consumer's gradients of the same tensors in one launch
    (128x128 and 256x256 kernels) == the plain launch's result + accum, rounded once."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 256, 256
    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    w = _rand((K, 3, 3, C), 1, 0.05)
    dys = [_rand((N, h, ww, K), 20 + i) for i, (h, ww) in enumerate(hw)]
    accs = [_rand((N, h, ww, C), 40 + i).to(cuda).bfloat16() for i, (h, ww) in enumerate(hw)]
    _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
    dyd = [d.to(cuda).bfloat16() for d in dys]
    HF.call("sod_conv_set_tile256", tile)
    try:
        got = HF.conv2d_dgrad_ml(dyd, wt, hw, 1, 1, 1, accums=accs)
        plain = HF.conv2d_dgrad_ml(dyd, wt, hw, 1, 1, 1)
    finally:
        HF.call("sod_conv_set_tile256", -1)
    x0 = torch.zeros(1)
    for g, p0, a, dy, (h, ww) in zip(got, plain, accs, dys, hw):
        ref, _ = onn.conv2d_backward(torch.zeros(N, h, ww, C), w, dy, 1, 1, 1)
        _close(g, ref + a.float().cpu(), 2 ** -7, "ml dgrad + accum")
        # against the two-step form: one bf16 rounding instead of two
        two = (p0.float() + a.float()).bfloat16().float()
        assert (g.float() - two).abs().max() <= 2 ** -6 * two.abs().max()
    with pytest.raises(Exception):
        HF.conv2d_dgrad_ml(dyd, wt, hw, 1, 1, 1, accums=accs[:-1])


@pytest.mark.parametrize("K", [8, 40, 80])
def test_conv_wgrad_folded_taps(cuda, K):
    """conv_wgrad_fold.hip: weight gradient of a 3x3 prediction conv with few output channels (FCOS bbox_pred + centerness: 8 padded,
    RetinaNet bbox_pred: 40, class scores: 80) with the taps folded into the tile rows == the un-folded kernel (an explicit split count
    keeps that one) == the oracle; multi-level, dY rows in the concatenated (N, L, K) buffer, a level smaller than one K-step, a level
    whose pixel count is no multiple of 32."""
    from slenderobjdet_amd.layers import functional as HF

    N, C = 2, 256
    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    xs = [_rand((N, h, ww, C), 10 + i) for i, (h, ww) in enumerate(hw)]
    dys = [_rand((N, h, ww, K), 20 + i) for i, (h, ww) in enumerate(hw)]
    xd = [x.to(cuda).bfloat16() for x in xs]
    L = sum(h * ww for h, ww in hw)
    offs = [sum(h * ww for h, ww in hw[:i]) for i in range(len(hw))]
    gbuf = torch.cat([d.reshape(N, -1, K) for d in dys], 1).to(cuda).bfloat16().contiguous()
    gviews = [gbuf.view(-1)[o * K:] for o in offs]
    dw_ref = torch.zeros(K, 3, 3, C)
    for x, dy in zip(xs, dys):
        dw_ref += onn.conv2d_backward(x, torch.zeros(K, 3, 3, C), dy, 1, 1, 1)[1]
    dw = torch.zeros((K, 3, 3, C), device=cuda)
    HF.conv2d_wgrad_ml(gviews, xd, dw, 3, 3, 1, 1, 1, dy_img_stride=L * K, K=K)
    _close(dw, dw_ref, 2e-4, f"folded wgrad K={K}")
    plain = torch.zeros_like(dw)
    HF.conv2d_wgrad_ml(gviews, xd, plain, 3, 3, 1, 1, 1, dy_img_stride=L * K, K=K, splits=3)
    _close(plain, dw_ref, 2e-4, "un-folded wgrad")
    assert (dw - plain).abs().max() <= 1e-4 * plain.abs().max()
    # accumulation into a non-zero gradient, per-output-channel scale
    qs = torch.rand(K, generator=torch.Generator().manual_seed(5)) + 0.5
    dw2 = torch.ones((K, 3, 3, C), device=cuda)
    HF.conv2d_wgrad_ml(gviews, xd, dw2, 3, 3, 1, 1, 1, dy_img_stride=L * K, K=K, qscale=qs.to(cuda))
    _close(dw2 - 1.0, dw_ref * qs.view(-1, 1, 1, 1), 3e-4, "folded wgrad accumulate + scale")


@pytest.mark.parametrize("tile", [0, 2])
def test_conv_multilevel_dgrad_padded_contraction(cuda, tile):
    """sod_conv2d_dgrad_ml_kpitch: dY rows of 72 channels in the concatenated (N, L, 72) buffer (RetinaNet's 720 class scores in small)
    contracted as 128 channels per tap against zero-padded transposed weights, on the 128x128 and on the 256x256 kernel (source pitch !=
    contraction width) == the per-chunk gather path of the plain entry point == the oracle; the rows at the very end of the buffer
    over-read into zero fill."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, K = 2, 256, 72
    hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    w = _rand((K, 3, 3, C), 1, 0.05)
    dys = [_rand((N, h, ww, K), 20 + i) for i, (h, ww) in enumerate(hw)]
    _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
    assert tuple(wt.shape) == (C, 3, 3, K)
    L = sum(h * ww for h, ww in hw)
    offs = [sum(h * ww for h, ww in hw[:i]) for i in range(len(hw))]
    # the buffer sits inside a larger allocation whose other bytes are NaN: no chunk past a pixel's 72 channels may ever be fetched as
    # data (the last pixel's would come from behind the buffer, and NaN x 0 is NaN)
    arena = torch.full((N * L * K + 4096,), float("nan"), device=cuda).bfloat16()
    gbuf = arena[:N * L * K].view(N, L, K)
    gbuf.copy_(torch.cat([d.reshape(N, -1, K) for d in dys], 1).to(cuda).bfloat16())
    gviews = [gbuf.view(-1)[o * K:] for o in offs]
    plain = HF.conv2d_dgrad_ml(gviews, wt, hw, 1, 1, 1, dy_img_stride=L * K, N=N)
    wt_pad = torch.nn.functional.pad(wt, (0, 128 - K))
    HF.call("sod_conv_set_tile256", tile)
    try:
        got = HF.conv2d_dgrad_ml(gviews, wt_pad, hw, 1, 1, 1, dy_img_stride=L * K, N=N, k_pitch=K)
    finally:
        HF.call("sod_conv_set_tile256", -1)
    for g, p0, dy, (h, ww) in zip(got, plain, dys, hw):
        ref, _ = onn.conv2d_backward(torch.zeros(N, h, ww, C), w, dy, 1, 1, 1)
        assert bool(torch.isfinite(g.float()).all())
        _close(g, ref, 2 ** -7, "padded-contraction dgrad")
        assert (g.float() - p0.float()).abs().max() <= 2 ** -7 * p0.float().abs().max()
    with pytest.raises(Exception):      # the padded width must be a multiple of 64
        HF.conv2d_dgrad_ml(gviews, torch.nn.functional.pad(wt, (0, 24)), hw, 1, 1, 1, dy_img_stride=L * K, N=N, k_pitch=K)


# ---- 256x256x64 8-phase kernel (conv_igemm256.hip), forced through sod_conv_set_tile256
T256_CASES = [
    # (N, H, W, C, K, R, stride, pad, dil)
    (2, 13, 21, 64, 256, 3, 1, 1, 1),     # one q-tile, ragged pixel tiles
    (1, 9, 11, 256, 64, 1, 1, 0, 1),      # Nout < 256 (weight rows out of range), T = 4
    (2, 14, 18, 64, 64, 3, 2, 1, 1),      # stride 2 forward
    (3, 25, 42, 256, 512, 3, 1, 1, 1),    # two q-tiles
    (1, 10, 10, 64, 64, 3, 1, 2, 2),      # dilation
    (1, 6, 7, 64, 72, 1, 1, 0, 1),        # a single K-tile (prologue/tail only)
    (1, 12, 10, 128, 256, 5, 1, 2, 1),    # 25 taps
    (2, 13, 21, 64, 720, 3, 1, 1, 1),     # three q-tiles, the last one 19 % empty (RetinaNet's class scores)
]


@pytest.fixture
def tile256(cuda):
    from slenderobjdet_amd import _C

    _C.call("sod_conv_set_tile256", 2)
    yield
    _C.call("sod_conv_set_tile256", -1)


@pytest.mark.parametrize("case", T256_CASES)
@pytest.mark.parametrize("out_f32", [True, False])
def test_conv256_fwd(cuda, tile256, case, out_f32):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 11)
    w = _rand((K, R, R, C), 12, 0.05)
    b = torch.randn(K, generator=torch.Generator().manual_seed(13))
    res = _rand(tuple(onn.conv2d(x, w, b, st, pad, dil).shape), 14)
    ref = onn.conv2d(x, w, b, st, pad, dil, res=res, relu=True)
    y = HF.conv2d_fwd(x.to(cuda).bfloat16(), w.to(cuda).bfloat16(), b.to(cuda), res=res.to(cuda).bfloat16(), stride=st, pad=pad, dil=dil,
                      relu=True, out_f32=out_f32)
    torch.cuda.synchronize()
    _close(y, ref, 2e-4 if out_f32 else 2 ** -7, f"conv256 fwd {case}")


@pytest.mark.parametrize("case", [c for c in T256_CASES if c[6] == 1])
def test_conv256_dgrad(cuda, tile256, case):
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K, R, st, pad, dil = case
    x = _rand((N, H, W, C), 21).requires_grad_(True)
    w = _rand((K, R, R, C), 22, 0.05)
    y = onn.conv2d(x, w, None, st, pad, dil)
    dy = _rand(tuple(y.shape), 23)
    y.backward(dy)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = HF.conv2d_dgrad(dy.to(cuda).bfloat16(), wt.to(cuda).bfloat16(), (H, W), st, pad, dil)
    torch.cuda.synchronize()
    _close(dx, x.grad, 2 ** -7, f"conv256 dgrad {case}")


def test_conv256_split_rounds_and_levels(cuda):
    """Default policy: whole rounds of 256x256 tiles on the 8-phase kernel, the short remainder (starting in the middle of a level)
    on the 128x128 kernel; result must equal the 128x128-only result bit for bit (same bf16 rounding of the same fp32 sums is not
    guaranteed across tilings, so compare against the oracle instead) and cover every pixel exactly once."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    props = torch.cuda.get_device_properties(0)
    cus = props.multi_processor_count
    N, C, K = 2, 128, 256
    # level 0 alone has a little more than two rounds of tiles; level 1 adds a few more
    h0 = 64
    w0 = (2 * cus * 256 + 40 * 256) // (N * h0) + 1
    hw = [(h0, w0), (9, 11)]
    xs = [_rand((N, h, w, C), 31 + i) for i, (h, w) in enumerate(hw)]
    wgt = _rand((K, 3, 3, C), 33, 0.05)
    b = torch.randn(K, generator=torch.Generator().manual_seed(34))
    refs = [onn.conv2d(x, wgt, b, 1, 1, 1, relu=True) for x in xs]
    _C.call("sod_conv_set_tile256", 1)
    try:
        ys = HF.conv2d_fwd_ml([x.to(cuda).bfloat16() for x in xs], wgt.to(cuda).bfloat16(), b.to(cuda), 1, 1, 1, relu=True)
        torch.cuda.synchronize()
    finally:
        _C.call("sod_conv_set_tile256", -1)
    for y, ref, (h, w) in zip(ys, refs, hw):
        _close(y, ref, 2 ** -7, f"split conv256 level {h}x{w}")


def test_conv_prof_event_pairs(cuda):
    """sod_conv_prof_enable / _collect: one hipEvent pair per forward dispatch of this thread, in call order, with the kernel variant."""
    import ctypes

    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    x = _rand((2, 24, 32, 64), 41).to(cuda).bfloat16()
    w1 = _rand((128, 3, 3, 64), 42, 0.05).to(cuda).bfloat16()
    w2 = _rand((8, 3, 3, 64), 43, 0.05).to(cuda).bfloat16()
    _C.call("sod_conv_prof_enable", 1)
    try:
        HF.conv2d_fwd(x, w1, None, stride=1, pad=1)
        HF.conv2d_fwd(x, w2, None, stride=1, pad=1, out_f32=True)
    finally:
        _C.call("sod_conv_prof_enable", 0)
    HF.conv2d_fwd(x, w1, None, stride=1, pad=1)          # not recorded
    ms, var, frac, mode = (ctypes.c_float * 8)(), (ctypes.c_int * 8)(), (ctypes.c_float * 8)(), (ctypes.c_int * 8)()
    n = _C.load().sod_conv_prof_collect(ms, var, frac, mode, 8)
    assert n == 2
    assert all(0.0 < ms[i] < 50.0 for i in range(2)) and mode[0] == 0 and mode[1] == 0 and frac[0] == 1.0
    assert var[0] // 100000 == 128 and var[1] // 100000 == 16          # BQ of the two tile variants
    assert _C.load().sod_conv_prof_collect(ms, var, frac, mode, 8) == 0   # cleared


def test_conv_epilogue_groupnorm_statistics(cuda):
    """sod_conv2d_fwd_ml_gnsum + sod_groupnorm_apply_ml (GroupNorm statistics gathered in the conv epilogue, both the 256x256 and the
    128x128 kernel, levels whose 64-pixel wave ranges straddle image boundaries) against conv -> separate GroupNorm passes: the same
    conv output bit for bit, mean / rstd to fp32 summation-order accuracy, the normalised output to one bf16 ulp; and against the
    oracle (F.conv2d + F.group_norm on the bf16-rounded conv output)."""
    from slenderobjdet_amd.layers import functional as HF

    N, C, G = 3, 256, 32
    hw = [(23, 37), (12, 19), (6, 10), (3, 5)]           # 851 / 228 / 60 / 15 pixels per image: none a multiple of 64
    xs = [_rand((N, h, w, C), 20 + i) for i, (h, w) in enumerate(hw)]
    w = _rand((C, 3, 3, C), 31, scale=(9 * C) ** -0.5)
    bias = torch.randn(C, generator=torch.Generator().manual_seed(32)) * 0.5
    gamma = torch.rand(C, generator=torch.Generator().manual_seed(33)) + 0.5
    beta = torch.randn(C, generator=torch.Generator().manual_seed(34)) * 0.2
    d = lambda t: t.to(cuda).to(torch.bfloat16).contiguous()
    f = lambda t: t.to(cuda).contiguous()
    xd, wd = [d(x) for x in xs], d(w)
    for tile256 in (1, 0):
        HF.call("sod_conv_set_tile256", 2 if tile256 else 0)
        try:
            y1, y2, st = HF.conv_gn_fwd_ml(xd, wd, f(bias), f(gamma), f(beta), G, relu=True)
            r1 = HF.conv2d_fwd_ml(xd, wd, f(bias), 1, 1, 1)
        finally:
            HF.call("sod_conv_set_tile256", 1)
        r2, rst = HF.groupnorm_fwd_ml(r1, f(gamma), f(beta), G, relu=True)
        for l in range(len(hw)):
            assert torch.equal(y1[l], r1[l]), (tile256, l)
            # mean to 1e-5 of the value scale, rstd to 1e-5 relative (different summation order of the same rounded values)
            scale = r1[l].float().abs().max().item()
            assert (st[l, :, :, 0] - rst[l, :, :, 0]).abs().max().item() <= 1e-5 * scale + 1e-6, (tile256, l)
            assert ((st[l, :, :, 1] - rst[l, :, :, 1]).abs() / rst[l, :, :, 1]).max().item() <= 2e-5, (tile256, l)
            dd = (y2[l].float() - r2[l].float()).abs().max().item()
            assert dd <= 2 ** -7 * max(r2[l].float().abs().max().item(), 1e-6), (tile256, l, dd)
            ref = onn.group_norm(onn.rb(onn.conv2d(xs[l], w, bias, pad=1)), gamma, beta, G, relu=True)
            _close(y2[l], ref, 2 ** -6, f"conv+gn level {l} tile256={tile256}")


def test_relu_bit_masks_match_tensor_masks(cuda):
    """sod_conv2d_fwd_bits records "stored output > 0" as one bit per element; sod_conv2d_dgrad_bits applies it.  The forward output
    is bit-identical to sod_conv2d_fwd's, the bits equal (y > 0) packed little-endian per 8 channels, and the data gradient with the
    bit mask (+ accumulate) is bit-identical to the one that re-reads the bf16 tensor as its mask."""
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = 2, 19, 27, 128, 512
    x, w = _rand((N, H, W, C), 41), _rand((K, 1, 1, C), 42, scale=C ** -0.5)
    res = _rand((N, H, W, K), 43)
    bias = torch.randn(K, generator=torch.Generator().manual_seed(44)) * 0.1
    d = lambda t: t.to(cuda).to(torch.bfloat16).contiguous()
    bits = torch.zeros(N * H * W * K // 8, dtype=torch.uint8, device=cuda)
    y = HF.conv2d_fwd(d(x), d(w), bias.to(cuda), d(res), relu=True, relu_bits=bits)
    y0 = HF.conv2d_fwd(d(x), d(w), bias.to(cuda), d(res), relu=True)
    assert torch.equal(y, y0)
    want = ((y.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device=cuda, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert 0.2 < (y > 0).float().mean().item() < 0.8
    # data gradient of a 1x1 conv K -> C' whose input is y: dx = mask(y) * (dy W + accum)
    Cp = 128
    dy, wt = _rand((N, H, W, Cp), 45), _rand((K, 1, 1, Cp), 46, scale=Cp ** -0.5)
    accum = _rand((N, H, W, K), 47)
    a = HF.conv2d_dgrad(d(dy), d(wt), (H, W), accum=d(accum), relu_bits=bits)
    b = HF.conv2d_dgrad(d(dy), d(wt), (H, W), accum=d(accum), relu_mask=y)
    assert torch.equal(a, b)
    assert (a != 0).any()


# conv_ws3.hip (persistent weight-stationary 3x3 kernel, 128 -> 128 channels: conv2 of the res3 bottleneck blocks), forced for every size with
# sod_conv_set_ws3(2): (N, H, W)
WS3_CASES = [(2, 100, 168), (1, 8, 14), (3, 13, 21), (2, 37, 50), (1, 9, 15)]


@pytest.mark.parametrize("case", WS3_CASES)
def test_weight_stationary_3x3_vs_oracle(cuda, case):
    """Forward (bias + ReLU) and data gradient (bf16 ReLU mask tensor of the input) of a 3x3 / stride 1 / pad 1 convolution 128 -> 128 through
    conv_ws3.hip (sod_conv_last_variant() == 7003) vs oracle/nn.py: the 8 x 14 tiling with its dropped halo columns, partial tiles at the right
    and bottom edges, the zero padding staged as out-of-range offsets, several tiles per workgroup and the double-buffered window."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    N, H, W = case
    C = K = 128
    x, w = _rand((N, H, W, C), 81).relu(), _rand((K, 3, 3, C), 82, (9 * C) ** -0.5)
    b = torch.randn(K, generator=torch.Generator().manual_seed(83))
    dy, act = _rand((N, H, W, K), 84), _rand((N, H, W, C), 85)
    d = lambda t: t.to(cuda).bfloat16()
    _C.call("sod_conv_set_ws3", 2)
    try:
        y = HF.conv2d_fwd(d(x), d(w), b.to(cuda), stride=1, pad=1, relu=True)
        assert int(_C.load().sod_conv_last_variant()) == 7003
        y2 = HF.conv2d_fwd(d(x), d(w), None, stride=1, pad=1)
        assert int(_C.load().sod_conv_last_variant()) == 7003
        _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
        dx = HF.conv2d_dgrad(d(dy), wt, (H, W), 1, 1, 1, relu_mask=d(act))
        assert int(_C.load().sod_conv_last_variant()) == 7003
        dx2 = HF.conv2d_dgrad(d(dy), wt, (H, W), 1, 1, 1)
        assert int(_C.load().sod_conv_last_variant()) == 7003
        torch.cuda.synchronize()
        # and bit for bit against the tiled kernel (same MFMA instruction; the K order differs: tap outer there, so NOT identical - compared loosely)
        _C.call("sod_conv_set_ws3", 0)
        y0 = HF.conv2d_fwd(d(x), d(w), b.to(cuda), stride=1, pad=1, relu=True)
        assert int(_C.load().sod_conv_last_variant()) != 7003
    finally:
        _C.call("sod_conv_set_ws3", -1)
    _close(y, onn.conv2d(x, w, b, 1, 1, 1, relu=True), 2 ** -7, f"ws3 fwd bias+relu {case}")
    _close(y2, onn.conv2d(x, w, None, 1, 1, 1), 2 ** -7, f"ws3 fwd plain {case}")
    dx_ref, _ = onn.conv2d_backward(x, w, dy, 1, 1, 1)
    _close(dx, dx_ref * (act > 0), 2 ** -7, f"ws3 dgrad + mask {case}")
    _close(dx2, dx_ref, 2 ** -7, f"ws3 dgrad plain {case}")
    assert (y.float() - y0.float()).abs().max().item() <= 2 ** -7 * y0.float().abs().max().item()


# conv_pw.hip against the ORACLE directly (round-5 review: the persistent kernel had only been compared with its sibling kernel).  Shapes
# the dispatcher routes to variant 7001: the expanding 1x1 convolutions of the bottleneck blocks (128 -> 512, 256 -> 1024), the
# contracting 512 -> 128 of res3, at a ragged pixel count (129 x 131 = 16 899: no multiple of the 32 / 64-pixel tiles).
PW_CASES = [(1, 129, 131, 128, 512), (1, 129, 131, 256, 1024), (1, 129, 131, 512, 128), (2, 100, 84, 512, 2048)]


@pytest.mark.parametrize("case", PW_CASES)
def test_persistent_pointwise_fwd_vs_oracle(cuda, case):
    """Forward C -> K through conv_pw.hip (sod_conv_last_variant() == 7001) vs oracle/nn.py: plain, and with bias + shortcut + ReLU."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = case
    x, w = _rand((N, H, W, C), 61).relu(), _rand((K, 1, 1, C), 62, C ** -0.5)
    b = torch.randn(K, generator=torch.Generator().manual_seed(63))
    res = _rand((N, H, W, K), 64)
    d = lambda t: t.to(cuda).bfloat16()
    y = HF.conv2d_fwd(d(x), d(w), None)
    assert int(_C.load().sod_conv_last_variant()) == 7001
    _close(y, onn.conv2d(x, w), 2 ** -7, f"conv_pw fwd plain {case}")
    bits = torch.zeros(N * H * W * K // 8, dtype=torch.uint8, device=cuda)
    y = HF.conv2d_fwd(d(x), d(w), b.to(cuda), d(res), relu=True, relu_bits=bits)
    assert int(_C.load().sod_conv_last_variant()) == 7001
    ref = onn.conv2d(x, w, b, res=res, relu=True)
    _close(y, ref, 2 ** -7, f"conv_pw fwd bias+res+relu {case}")
    # the 1-bit ReLU mask against the oracle's own decision, away from the undecided band (|pre-activation| below one bf16 ulp of the scale)
    pre = onn.conv2d(x, w, b, res=res)
    got = ((bits.cpu().reshape(-1, 1) >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(pre.shape).bool()
    decided = pre.abs() > 2 ** -6 * pre.abs().max()
    assert torch.equal(got[decided], (pre > 0)[decided])
    assert 0.2 < got.float().mean().item() < 0.8


@pytest.mark.parametrize("case", PW_CASES)
def test_persistent_pointwise_dgrad_vs_oracle(cuda, case):
    """Data gradient of a 1x1 conv K' -> C' whose CONTRACTION width is ``C`` and whose output width is ``K`` (conv1 of a bottleneck block
    for the expanding shapes, conv3 for 512 -> 128) through conv_pw.hip vs oracle/nn.py's autograd: plain, and accumulate + ReLU mask."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    N, H, W, C, K = case            # dy has C channels, dx has K channels: the forward conv was K -> C with weight (C, 1, 1, K)
    xin = _rand((N, H, W, K), 71)
    w = _rand((C, 1, 1, K), 72, K ** -0.5)
    dy = _rand((N, H, W, C), 73)
    dx_ref, _ = onn.conv2d_backward(xin, w, dy)
    _, wt = HF.weight_prep(w.to(cuda), want_krsc=False)
    d = lambda t: t.to(cuda).bfloat16()
    dx = HF.conv2d_dgrad(d(dy), wt, (H, W))
    assert int(_C.load().sod_conv_last_variant()) == 7001
    _close(dx, dx_ref, 2 ** -7, f"conv_pw dgrad plain {case}")
    acc, act = _rand((N, H, W, K), 74), _rand((N, H, W, K), 75)
    bits = ((act.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(1).to(torch.uint8).to(cuda)
    dx = HF.conv2d_dgrad(d(dy), wt, (H, W), accum=d(acc), relu_bits=bits)
    assert int(_C.load().sod_conv_last_variant()) == 7001
    _close(dx, (dx_ref + acc) * (act > 0), 2 ** -7, f"conv_pw dgrad accum+bits {case}")
    dx = HF.conv2d_dgrad(d(dy), wt, (H, W), relu_mask=d(act))
    assert int(_C.load().sod_conv_last_variant()) == 7001
    _close(dx, dx_ref * (act > 0), 2 ** -7, f"conv_pw dgrad mask tensor {case}")


@pytest.mark.parametrize("C,K", [(128, 512), (256, 1024), (512, 2048), (512, 128), (256, 256)])
@pytest.mark.parametrize("shape", [(2, 100, 84), (1, 129, 131)])
def test_persistent_pointwise_kernel_is_bit_identical_to_the_tiled_kernels(cuda, C, K, shape):
    """conv_pw.hip (persistent, weight-stationary: the expanding 1x1 convolutions of the bottleneck blocks and their data gradients)
    against the tiled kernels it replaces - same MFMA instruction, same K order, same epilogue arithmetic, so EQUAL bit for bit: forward
    with bias + shortcut + ReLU + 1-bit mask, plain forward, the data gradient with accumulate + bit mask, without the accumulate
    operand, and under a bf16 mask TENSOR (conv3's data gradient); expanding shapes and the contracting 512 -> 128 of res3; pixel counts
    that are no multiple of the 32 / 64-pixel tiles (ragged last tile, dead prefetches), reversed tile order."""
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    N, H, W = shape
    g = torch.Generator().manual_seed(11)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(cuda)
    x = r(N, H, W, C).relu().bfloat16()
    res = r(N, H, W, K).bfloat16()
    w = r(K, 1, 1, C, sc=C ** -0.5)
    wk, wt = HF.weight_prep(w)              # KRSC (K,1,1,C) for the forward conv C -> K
    bias = r(K)
    # the data gradient of a CONTRACTING conv K -> C (conv1 of a block) expands C -> K channels: transposed weights (K, 1, 1, C) as well
    w1 = r(C, 1, 1, K, sc=K ** -0.5)
    _, w1t = HF.weight_prep(w1)             # CRSK copy of (C,1,1,K): shape (K,1,1,C)
    da = r(N, H, W, C).bfloat16()
    gskip = r(N, H, W, K).bfloat16()

    def run(on, reverse=False):
        _C.call("sod_conv_set_pw", on)
        _C.call("sod_conv_set_reverse", 1 if reverse else 0)
        try:
            bits = torch.zeros(N * H * W * K // 8, dtype=torch.uint8, device=cuda)
            y = HF.conv2d_fwd(x, wk, bias, res, relu=True, relu_bits=bits)
            v1 = int(_C.load().sod_conv_last_variant())
            y2 = HF.conv2d_fwd(x, wk, None, None, relu=False)
            v2 = int(_C.load().sod_conv_last_variant())
            dx = HF.conv2d_dgrad(da, w1t, (H, W), accum=gskip, relu_bits=bits)
            v3 = int(_C.load().sod_conv_last_variant())
            dx2 = HF.conv2d_dgrad(da, w1t, (H, W), relu_bits=bits)
            dx3 = HF.conv2d_dgrad(da, w1t, (H, W))
            dx4 = HF.conv2d_dgrad(da, w1t, (H, W), relu_mask=y)
            v4 = int(_C.load().sod_conv_last_variant())
            torch.cuda.synchronize()
            return (y, bits, y2, dx, dx2, dx3, dx4), (v1, v2, v3, v4)
        finally:
            _C.call("sod_conv_set_pw", -1)
            _C.call("sod_conv_set_reverse", 0)

    ref, vref = run(0)
    assert 7001 not in vref
    for rev in (False, True):
        got, vgot = run(1, rev)
        assert vgot == (7001, 7001, 7001, 7001), vgot            # the persistent kernel really ran
        for name, a, b in zip(("fwd+res+relu", "bits", "fwd plain", "dgrad+accum+bits", "dgrad+bits", "dgrad plain", "dgrad+mask tensor"), got, ref):
            assert torch.equal(a, b), (name, rev, C, K, shape, (a.float() - b.float()).abs().max().item())
    assert 0.2 < (ref[0] > 0).float().mean().item() < 0.8 and (ref[3] != 0).any()
