"""Compare the tiled ROIAlign backward with the plain scatter on adversarial rois (huge, off-image, tiny, rotated)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(plain):
    os.environ["SOD_ROI_PLAIN"] = "1" if plain else "0"
    from slenderobjdet_amd.layers import functional as HF

    g = torch.Generator().manual_seed(0)
    out = {}
    for rotated in (False, True):
        for (H, W, scale) in ((50, 84, 1 / 16), (25, 42, 1 / 32), (200, 336, 1 / 4)):
            R = 300
            cx, cy = torch.rand(R, generator=g) * 1500 - 80, torch.rand(R, generator=g) * 900 - 50
            w = torch.exp(torch.rand(R, generator=g) * 9.0)      # 1 .. 8000 px
            h = torch.exp(torch.rand(R, generator=g) * 9.0)
            bidx = torch.randint(0, 2, (R,), generator=g).float()
            if rotated:
                rois = torch.stack((bidx, cx, cy, w, h, torch.rand(R, generator=g) * 360 - 180), 1)
            else:
                rois = torch.stack((bidx, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), 1)
            dout = torch.randn(R, 7, 7, 64, generator=g) * torch.exp(torch.randn(R, 1, 1, 1, generator=g) * 3)
            dx = HF.roi_align_bwd(dout.cuda(), rois.cuda().contiguous(), (2, H, W, 64), scale, 0, rotated)
            out[(rotated, H)] = dx.cpu()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        torch.save(run(sys.argv[1] == "plain"), f"/tmp/roi_{sys.argv[1]}.pt")
    else:
        for m in ("plain", "tile"):
            subprocess.check_call([sys.executable, __file__, m])
        a, b = torch.load("/tmp/roi_plain.pt"), torch.load("/tmp/roi_tile.pt")
        for k in a:
            fa, fb = bool(torch.isfinite(a[k]).all()), bool(torch.isfinite(b[k]).all())
            err = (a[k] - b[k]).abs().max().item() / max(a[k].abs().max().item(), 1e-30)
            print(k, "finite", fa, fb, "max rel err", err, "max", a[k].abs().max().item())
