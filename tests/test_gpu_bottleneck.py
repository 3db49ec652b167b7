"""GPU parity of the fused frozen bottleneck block (csrc/bottleneck_fused.hip, through the C ABI) against
(a) the CPU oracle: the three / four convolutions of detectron2's BottleneckBlock with FrozenBatchNorm2d folded, built from
    oracle/nn.py (F.conv2d, fp32) with the intermediates rounded to bf16 where the HIP side stores them, and
(b) the un-fused product path (one implicit-GEMM launch per convolution) on the same operands.
Tolerance: 2^-7 * max|ref| (the output is bf16: one ulp at the top of the range is 2^-8 relative; the fused kernel adds the
projection shortcut in fp32 where the un-fused path rounds it to bf16 first).
"""
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu


def _stage(cuda, seed=0):
    from bench import make_cfg
    from slenderobjdet_amd.modeling import build_model

    torch.manual_seed(seed)
    model = build_model(make_cfg(50))
    stage = model.backbone.bottom_up.res2
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                  # non-trivial FrozenBN statistics (before the first prepare())
        for blk in stage:
            for m in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut):
                if m is None:
                    continue
                k = m.out_channels
                m.bn_weight.copy_((torch.rand(k, generator=g) * 0.5 + 0.75).to(cuda))
                m.bn_bias.copy_((torch.randn(k, generator=g) * 0.2).to(cuda))
                m.bn_running_mean.copy_((torch.randn(k, generator=g) * 0.1).to(cuda))
                m.bn_running_var.copy_((torch.rand(k, generator=g) * 0.5 + 0.75).to(cuda))
    return stage


def _oracle_block(blk, x):
    """x NHWC fp32 (bf16-representable) -> relu(conv3(conv2(conv1 x)) + shortcut x) with bf16 storage of every stored tensor."""
    def fold(m):
        scale, shift = onn.frozen_bn_fold(m.bn_weight.cpu(), m.bn_bias.cpu(), m.bn_running_mean.cpu(), m.bn_running_var.cpu())
        return onn.rb(m.weight.detach().cpu() * scale.view(-1, 1, 1, 1)), shift

    w1, b1 = fold(blk.conv1)
    w2, b2 = fold(blk.conv2)
    w3, b3 = fold(blk.conv3)
    a = onn.rb(onn.conv2d(x, w1, b1, relu=True))
    b = onn.rb(onn.conv2d(a, w2, b2, pad=1, relu=True))
    if blk.shortcut is not None:
        ws, bs = fold(blk.shortcut)
        sc = onn.conv2d(x, ws, bs)
    else:
        sc = x
    return onn.rb(onn.conv2d(b, w3, b3, res=sc, relu=True))


@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 8, 16), (3, 5, 70), (1, 64, 96)])
def test_fused_frozen_bottleneck_vs_oracle_and_unfused(cuda, shape):
    """Partial tiles on both axes, a single exact tile, a tile row shorter than the 8-row tile, and a many-tile case; both block
    kinds of res2 (projection shortcut 64 -> 256, identity shortcut 256 -> 256) chained as the stage runs them."""
    from slenderobjdet_amd.modeling.backbone import resnet

    n, h, w = shape
    stage = _stage(cuda)
    g = torch.Generator().manual_seed(7)
    x = onn.rb(torch.randn(n, h, w, 64, generator=g))
    xd = x.to(cuda).to(torch.bfloat16)
    with torch.no_grad():
        resnet.BNECK_FUSED = True
        fused = stage(xd)
        resnet.BNECK_FUSED = False
        try:
            unfused = stage(xd)
        finally:
            resnet.BNECK_FUSED = True
    ref = x
    for blk in stage:
        ref = _oracle_block(blk, ref)
    assert tuple(fused.shape) == (n, h, w, 256)
    scale = ref.abs().max().item()
    err_o = (fused.float().cpu() - ref).abs().max().item()
    err_u = (fused.float() - unfused.float()).abs().max().item()
    assert err_o <= 2 ** -7 * scale, (err_o, scale)
    assert err_u <= 2 ** -7 * scale, (err_u, scale)
    # the un-fused product path sits at the same distance from the oracle: the fusion adds no error of its own
    err_uo = (unfused.float().cpu() - ref).abs().max().item()
    assert err_o <= max(2.0 * err_uo, 2 ** -8 * scale), (err_o, err_uo)


def test_fused_frozen_bottleneck_single_blocks(cuda):
    """Each block kind alone through the functional entry point (projection shortcut on 64 channels, identity on 256)."""
    from slenderobjdet_amd.layers import functional as HF

    g = torch.Generator().manual_seed(3)

    def rnd(*s, scale=1.0):
        return onn.rb(torch.randn(*s, generator=g) * scale)

    for cin, proj in ((256, False), (64, True)):
        x = rnd(2, 19, 33, cin)
        w1, w2, w3 = rnd(64, 1, 1, cin, scale=cin ** -0.5), rnd(64, 3, 3, 64, scale=1 / 24.0), rnd(256, 1, 1, 64, scale=0.125)
        wsc = rnd(256, 1, 1, cin, scale=cin ** -0.5) if proj else None
        b1, b2, b3 = torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
        a = onn.rb(onn.conv2d(x, w1, b1, relu=True))
        b = onn.rb(onn.conv2d(a, w2, b2, pad=1, relu=True))
        sc = onn.conv2d(x, wsc) if proj else x
        ref = onn.rb(onn.conv2d(b, w3, b3, res=sc, relu=True))
        d = lambda t: t.to(cuda).to(torch.bfloat16).contiguous()
        f = lambda t: t.to(cuda).contiguous()
        got = HF.bottleneck_frozen_fwd(d(x), d(w1), f(b1), d(w2), f(b2), d(w3), f(b3), d(wsc) if proj else None)
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 2 ** -7 * ref.abs().max().item(), (cin, proj, err, ref.abs().max().item())


def test_fused_frozen_bottleneck_rejects_bad_operands(cuda):
    from slenderobjdet_amd import _C
    from slenderobjdet_amd.layers import functional as HF

    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=cuda)
    b = lambda k: torch.zeros(k, dtype=torch.float32, device=cuda)
    with pytest.raises(_C.SlenderHipError):      # identity shortcut needs 256 input channels
        HF.bottleneck_frozen_fwd(z(1, 8, 16, 64), z(64, 1, 1, 64), b(64), z(64, 3, 3, 64), b(64), z(256, 1, 1, 64), b(256), None)
    with pytest.raises(_C.SlenderHipError):      # projection shortcut only on the 64-channel stem output
        HF.bottleneck_frozen_fwd(z(1, 8, 16, 128), z(64, 1, 1, 128), b(64), z(64, 3, 3, 64), b(64), z(256, 1, 1, 64), b(256), z(256, 1, 1, 128))
    with pytest.raises(_C.SlenderHipError):      # CPU tensors: no fallback
        HF.bottleneck_frozen_fwd(z(1, 8, 16, 256).cpu(), z(64, 1, 1, 256), b(64), z(64, 3, 3, 64), b(64), z(256, 1, 1, 64), b(256), None)
