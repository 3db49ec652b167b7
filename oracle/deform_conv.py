"""Oracle deformable convolution v1/v2 (CPU, fp32/fp64, differentiable through autograd).

Restates detectron2's DeformConv / ModulatedDeformConv sampling rule (source absent): offset
channel 2k = dy, 2k+1 = dx of kernel tap k, bilinear sampling, samples outside (-1, H) x (-1, W) contribute zero and
corners outside the image contribute zero.  Pinned forward-only by the reference's KAT (tests/golden/deform_conv_kat.npz);
backward is "parity unpinned" and checked with torch.autograd.gradcheck.
"""
import torch


def _bilinear_gather(x, py, px):
    """x (N,C,H,W); py/px (N,Ho,Wo) float sample coordinates -> (N,C,Ho,Wo) with the DCN zero rules."""
    N, C, H, W = x.shape
    valid = (py > -1) & (px > -1) & (py < H) & (px < W)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    out = 0
    for dy, wy in ((0, 1 - ly), (1, ly)):
        for dx, wx in ((0, 1 - lx), (1, lx)):
            yy, xx = (y0 + dy).long(), (x0 + dx).long()
            ok = valid & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(N, 1, -1).expand(N, C, -1)
            v = torch.gather(x.reshape(N, C, H * W), 2, idx).view(N, C, *py.shape[1:])
            out = out + v * (wy * wx * ok.to(x.dtype)).unsqueeze(1)
    return out


def deform_conv2d(x, offset, weight, bias=None, stride=1, pad=0, dil=1, mask=None, deformable_groups=1, sample_hook=None, groups=1):
    """x (N,C,H,W), offset (N, 2*kh*kw*dg, Ho, Wo), weight (K,C/groups,kh,kw), mask (N, kh*kw*dg, Ho, Wo) or None.  ``groups`` (detectron2's
    DeformConv argument): output channels [j K/g, (j+1) K/g) see input channels [j C/g, (j+1) C/g) only - computed on the block-diagonal
    dense weight, which autograd differentiates back to the grouped one."""
    N, C, H, W = x.shape
    K, _, kh, kw = weight.shape
    if groups > 1:
        Kg, Cg = K // groups, weight.shape[1]
        zero = weight.new_zeros(Kg, Cg, kh, kw)
        weight = torch.cat([torch.cat([weight[j * Kg:(j + 1) * Kg] if jj == j else zero for jj in range(groups)], dim=1) for j in range(groups)], dim=0)
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    dg = deformable_groups
    cg = C // dg
    base_y = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, Ho, 1)
    base_x = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, Wo)
    out = torch.zeros(N, K, Ho, Wo, dtype=x.dtype)
    for g in range(dg):
        xs = x[:, g * cg:(g + 1) * cg]
        for i in range(kh):
            for j in range(kw):
                k = (g * kh + i) * kw + j
                py = base_y + i * dil + offset[:, 2 * k]
                px = base_x + j * dil + offset[:, 2 * k + 1]
                s = _bilinear_gather(xs, py, px)
                if mask is not None:
                    s = s * mask[:, k].unsqueeze(1)
                if sample_hook is not None:      # e.g. bf16 rounding of the gathered columns (what the HIP path stores)
                    s = sample_hook(s)
                out = out + torch.einsum("nchw,kc->nkhw", s, weight[:, g * cg:(g + 1) * cg, i, j])
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
