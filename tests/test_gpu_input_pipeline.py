"""GPU parity of the on-device input pipeline: resize (PIL BILINEAR semantics) + flip + normalise + pad + NHWC(8)
bf16 in one launch vs the numpy oracle (oracle/resize.py; reference slender_det/data/utils.py:29-50, fcosv2.py:268-275)."""
import numpy as np
import pytest
import torch

from oracle import resize as orz

pytestmark = pytest.mark.gpu


def test_resize_flip_preprocess_bit_exact(cuda):
    from slenderobjdet_amd.data import DeviceInputPipeline

    rng = np.random.RandomState(1)
    imgs = [rng.randint(0, 256, s).astype(np.uint8) for s in ((48, 64, 3), (75, 50, 3), (33, 90, 3), (64, 64, 3))]
    choices = [(80, 107, False), (96, 64, True), (30, 82, True), (64, 64, False)]      # up-scale, up + flip, down + flip, identity
    mean, std = (103.53, 116.28, 123.675), (57.375, 57.12, 58.395)
    pipe = DeviceInputPipeline(pixel_mean=mean, pixel_std=std, size_divisibility=32)
    boxes = [torch.tensor([[4.0, 5.0, 40.0, 30.0]]), torch.tensor([[10.0, 10.0, 45.0, 70.0]]), torch.zeros((0, 4)), torch.tensor([[0.0, 0.0, 64.0, 64.0]])]
    out, sizes, new_boxes, _ = pipe([torch.from_numpy(i).to(cuda) for i in imgs], [b.to(cuda) for b in boxes], choices=choices)
    ref = orz.pipeline(imgs, choices, mean, std, 32)
    assert tuple(out.shape) == (4, ref.shape[1], ref.shape[2], 8) and sizes == [(c[0], c[1]) for c in choices]
    got = out.float().cpu()
    assert (got[..., 3:] == 0).all()
    want = torch.from_numpy(ref).to(torch.bfloat16).float()          # the kernel rounds (v - mean) / std to bf16 once
    assert torch.equal(got[..., :3], want), float((got[..., :3] - want).abs().max())
    for b, im, c, nb in zip(boxes, imgs, choices, new_boxes):
        assert np.allclose(nb.cpu().numpy(), orz.transform_boxes(b.numpy(), im.shape[0], im.shape[1], c[0], c[1], c[2]).reshape(-1, 4), atol=1e-4)


def test_resize_full_size_and_model_consumes_it(cuda):
    """COCO-sized inputs (480x640 / 427x640 -> shortest edge 800, max 1333): bit-exact against the oracle on a strip of rows, and the
    batch tensor has the layout FCOSV2.preprocess_image produces (the backbone consumes it unchanged)."""
    from bench import make_cfg
    from slenderobjdet_amd.data import DeviceInputPipeline
    from slenderobjdet_amd.modeling import build_model

    rng = np.random.RandomState(2)
    imgs = [rng.randint(0, 256, s).astype(np.uint8) for s in ((480, 640, 3), (427, 640, 3))]
    cfg = make_cfg(18)
    pipe = DeviceInputPipeline(min_sizes=(800,), max_size=1333, flip_prob=0.5, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, seed=3)
    out, sizes, _, choices = pipe([torch.from_numpy(i).to(cuda) for i in imgs])
    assert sizes == [(800, 1067), (800, 1199)] and tuple(out.shape) == (2, 800, 1216, 8)
    for i, (im, c) in enumerate(zip(imgs, choices)):
        r = orz.pil_resize_bilinear(im, c[0], c[1])
        if c[2]:
            r = r[:, ::-1, :]
        strip = slice(390, 410)
        want = torch.from_numpy((r[strip].astype(np.float32) - np.asarray(cfg.MODEL.PIXEL_MEAN, np.float32)) / np.asarray(cfg.MODEL.PIXEL_STD, np.float32))
        assert torch.equal(out[i, strip, : c[1], :3].float().cpu(), want.to(torch.bfloat16).float())
        assert (out[i, :, c[1]:] == 0).all()
    torch.manual_seed(0)
    model = build_model(cfg)
    model.eval()
    with torch.no_grad():
        feats = model.backbone(out)
    assert tuple(feats["p3"].shape) == (2, 100, 152, 256) and torch.isfinite(feats["p3"].float()).all()


def test_fused_stem_matches_unfused(cuda):
    """normalise + conv7x7s2 + FrozenBN + ReLU + max-pool in one kernel on raw uint8 images (csrc/stem_fused.hip) vs the three
    separate launches (preprocess_batch -> conv kernel -> maxpool): same values up to the fp32 summation order of the 147-term dot
    products (one bf16 ulp where a sum sits on a rounding boundary), on images of different sizes incl. the zero padding."""
    from bench import make_cfg
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model

    cfg = make_cfg(18)
    torch.manual_seed(0)
    model = build_model(cfg)
    stem = model.backbone.bottom_up.stem
    with torch.no_grad():                  # non-trivial FrozenBN statistics
        stem.conv1.bn_weight.copy_(torch.rand(64, device=cuda) + 0.5)
        stem.conv1.bn_bias.copy_(torch.randn(64, device=cuda) * 50)
        stem.conv1.bn_running_mean.copy_(torch.randn(64, device=cuda) * 10)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).to(cuda) for h, w in ((97, 130), (128, 61), (64, 64))]
    Hp, Wp = 128, 160
    raw = HF.RawImageBatch(imgs, [(i.shape[1], i.shape[2]) for i in imgs], (Hp, Wp), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    with torch.no_grad():
        fused = stem(raw)
        assert tuple(fused.shape) == (3, Hp // 4, Wp // 4, 64)
        ref = stem(raw.materialize())
    d = (fused.float() - ref.float()).abs()
    scale = ref.float().abs().max().item()
    assert d.max().item() <= 2 ** -7 * scale, (d.max().item(), scale)
    assert (d > 0).float().mean().item() < 0.02                       # almost every element is bit-identical
    # and against the fp32 oracle of the same stem
    from oracle import nn as onn

    x = torch.zeros(3, 3, Hp, Wp)
    for i, im in enumerate(imgs):
        x[i, :, : im.shape[1], : im.shape[2]] = onn.rb((im.cpu().float() - torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)) / torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1))
    c = stem.conv1
    scale_bn = (c.bn_weight * torch.rsqrt(c.bn_running_var + 1e-5)).cpu()
    shift = (c.bn_bias - c.bn_running_mean * c.bn_weight * torch.rsqrt(c.bn_running_var + 1e-5)).cpu()
    w = onn.rb(c.weight.detach().cpu() * scale_bn.view(-1, 1, 1, 1))
    y = torch.relu(torch.nn.functional.conv2d(x, w.permute(0, 3, 1, 2), shift, stride=2, padding=3))
    y = torch.nn.functional.max_pool2d(onn.rb(y), 3, 2, 1).permute(0, 2, 3, 1)
    assert (fused.float().cpu() - y).abs().max().item() <= 2 ** -7 * y.abs().max().item()
