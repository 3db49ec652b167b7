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


@pytest.mark.parametrize("CN,CW,N,H,W", [(128, 512, 2, 20, 36), (256, 1024, 2, 13, 21), (128, 512, 1, 7, 9), (256, 1024, 3, 50, 84)])
def test_bottleneck_pair_is_bit_identical_to_the_two_launches(cuda, CN, CW, N, H, W):
    """sod_bottleneck_pair (csrc/bneck_pair.hip): the expanding 1x1 convolution with its add operand / nonlinearity and the contracting
    1x1 convolution behind it in one launch, against the two launches it replaces - same MFMA instruction, same k order, same epilogue
    arithmetic, so EQUAL bit for bit, wide tensor, ReLU bits and narrow tensor, in the forward form (conv3 + residual + ReLU -> conv1 +
    ReLU of the next block) and the backward form (conv1's data gradient + identity gradient under the ReLU bits -> conv3's data
    gradient under its input's ReLU mask).  Tile sizes: ragged last tile, fewer tiles than CUs, several tiles per CU."""
    from slenderobjdet_amd.layers import functional as HF

    g = torch.Generator().manual_seed(5)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(cuda)
    xin = r(N, H, W, CN).relu().bfloat16()
    add = r(N, H, W, CW).bfloat16()
    w_e, w_c = r(CW, 1, 1, CN, sc=0.05), r(CN, 1, 1, CW, sc=0.03)
    we, we_t = HF.weight_prep(w_e)          # KRSC (CW,1,1,CN) and CRSK (CN,1,1,CW)
    wc, wc_t = HF.weight_prep(w_c)          # KRSC (CN,1,1,CW) and CRSK (CW,1,1,CN)
    be, bc = r(CW), r(CN)
    assert HF.bottleneck_pair_supported(CN, CW)
    # ---- forward form
    bits_ref = torch.empty(N * H * W * CW // 8, dtype=torch.uint8, device=cuda)
    y_ref = HF.conv2d_fwd(xin, we, be, add, relu=True, relu_bits=bits_ref)
    h_ref = HF.conv2d_fwd(y_ref, wc, bc, relu=True)
    y, bits, h = HF.bottleneck_pair(xin, add, we, be, wc, bc, 0)
    assert torch.equal(y, y_ref) and torch.equal(bits, bits_ref) and torch.equal(h, h_ref)
    # ---- backward form: da (narrow) -> g (wide) = bits ? da x W1 + g_skip : 0 -> db (narrow) = (b > 0) ? g x W3 : 0
    da = r(N, H, W, CN).bfloat16()
    gskip = r(N, H, W, CW).bfloat16()
    b = r(N, H, W, CN).bfloat16()
    g_ref = HF.conv2d_dgrad(da, wc_t, (H, W), accum=gskip, relu_bits=bits_ref)          # conv1: weights w_c (CN out, CW in); its CRSK copy is (CW,1,1,CN)
    db_ref = HF.conv2d_dgrad(g_ref, we_t, (H, W), relu_mask=b)                          # conv3: weights w_e (CW out, CN in); its CRSK copy is (CN,1,1,CW)
    g2, none_bits, db = HF.bottleneck_pair(da, gskip, wc_t, None, we_t, None, 1, bits_in=bits_ref, mask2=b)
    assert none_bits is None
    assert torch.equal(g2, g_ref) and torch.equal(db, db_ref)
