"""GPU parity for the reference's own native operators (BorderAlign, CornerPool) against the oracle."""
import pytest
import torch

from oracle import slender_ops as oso

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims", [(2, 3, 12, 16, 5, 4), (2, 4, 37, 41, 150, 10)])
def test_border_align_fwd_bwd(cuda, dims):
    from slenderobjdet_amd.layers.border_align import BorderAlign

    g = torch.Generator().manual_seed(0)
    B, C, H, W, K, pool = dims
    feat = torch.randn(B, 4 * C, H, W, generator=g)
    xy = torch.rand(B, K, 2, generator=g) * torch.tensor([W * 0.5, H * 0.5])
    wh = torch.rand(B, K, 2, generator=g) * torch.tensor([W * 0.45, H * 0.45]) + 1.0
    boxes = torch.cat([xy, xy + wh], dim=2)
    fr = feat.clone().requires_grad_(True)
    ref = oso.border_align(fr, boxes, pool)
    dout = torch.randn(ref.shape, generator=g)
    (gref,) = torch.autograd.grad(ref, fr, dout)
    fd = feat.to(cuda).requires_grad_(True)
    out = BorderAlign(pool)(fd, boxes.to(cuda))
    assert out.shape == (B, C, K, 4)
    # pool 10: the kernel steps along a border by repeated fp32 addition as the reference's CUDA kernel does (BorderAlign_cuda.cu:117-131),
    # the oracle adds Python floats (float64); ten steps leave ~1e-6 of a pixel between them, times a unit-variance feature gradient
    tol = 1e-5 if pool <= 4 else 1e-4
    assert (out.detach().cpu() - ref.detach()).abs().max() < tol
    out.backward(dout.to(cuda))
    assert (fd.grad.cpu() - gref).abs().max() < tol * max(1.0, gref.abs().max().item())


@pytest.mark.parametrize("mode", ["bottom", "top", "left", "right"])
@pytest.mark.parametrize("shape,ties", [((2, 3, 9, 13), True), ((2, 2, 70, 150), True), ((1, 3, 65, 129), False), ((1, 2, 3, 64), True)])
def test_corner_pool_fwd_bwd(cuda, mode, shape, ties):
    """Lines longer than a wave (the W scans carry the running maximum and its position from one 64-element chunk to the next), lines of
    exactly 64 and 64 + 1 elements, integer values (plenty of ties: the gradient goes to the position torch.cummax reports) and real values."""
    from slenderobjdet_amd.layers.corner_pool import CornerPool

    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 5, shape, generator=g).float() if ties else torch.randn(shape, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = oso.corner_pool(xr, mode)
    dy = torch.randn(ref.shape, generator=g)
    (gref,) = torch.autograd.grad(ref, xr, dy)
    xd = x.to(cuda).requires_grad_(True)
    y = CornerPool(mode)(xd)
    assert torch.equal(y.detach().cpu(), ref.detach())
    y.backward(dy.to(cuda))
    assert (xd.grad.cpu() - gref).abs().max() < 1e-5 * max(1.0, gref.abs().max().item())
