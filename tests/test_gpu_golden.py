"""GPU parity against the committed golden vectors of the reference's own Python (tests/golden/), through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


@pytest.mark.parametrize("lt", ["iou", "linear_iou", "giou"])
def test_iou_loss_vs_reference_vectors(cuda, lt):
    from slenderobjdet_amd.layers import functional as HF

    d = _load("iou_loss.npz")
    pred, tgt, w = (torch.tensor(d[k]).to(cuda) for k in ("pred", "target", "weight"))
    s, _ = HF.iou_loss_fwd(pred, tgt, w, lt)
    np.testing.assert_allclose(s.cpu().numpy()[0], d[f"loss_{lt}"], rtol=1e-5)       # north_star: within 1e-3 rel; we hold 1e-5
    g = HF.iou_loss_bwd(pred, tgt, w, lt)
    np.testing.assert_allclose(g.cpu().numpy(), d[f"grad_{lt}"], rtol=1e-4, atol=1e-7)
    s2, _ = HF.iou_loss_fwd(pred, tgt, None, lt)
    np.testing.assert_allclose(s2.cpu().numpy()[0], d[f"loss_noweight_{lt}"], rtol=1e-5)


@pytest.mark.parametrize("radius", [0.0, 1.5])
def test_fcos_assign_vs_reference_vectors(cuda, radius):
    """Labels and regression targets bit-exact against compute_targets_for_locations of the reference."""
    from slenderobjdet_amd.layers import functional as HF

    d = _load("fcos_targets.npz")
    hw = [tuple(int(v) for v in s) for s in d["shapes"]]
    strides = [int(s) for s in d["strides"]]
    boxes = [torch.tensor(d[f"boxes{i}"]) for i in range(3)]
    classes = [torch.tensor(d[f"classes{i}"]).int() for i in range(3)]
    offs = torch.tensor([0] + [len(b) for b in boxes]).cumsum(0).int()
    lab, reg, ctr, stats = HF.fcos_assign(torch.cat(boxes).to(cuda), torch.cat(classes).to(cuda), offs.to(cuda), 3, hw, strides,
                                          [[-1, 64], [64, 128], [128, 256], [256, 512], [512, 100000000]], radius, 80)
    np.testing.assert_array_equal(lab.cpu().numpy(), d[f"labels_r{radius}"])
    np.testing.assert_array_equal(reg.cpu().numpy(), d[f"reg_r{radius}"])
    assert int(stats[0].item()) == int((d[f"labels_r{radius}"] != 80).sum())


def test_centerness_vs_reference_vectors(cuda):
    """ctr targets produced by the assign kernel equal compute_centerness_targets on the reference's vectors."""
    from slenderobjdet_amd.layers import functional as HF

    d = _load("centerness.npz")
    reg = torch.tensor(d["reg"])
    # one location per "box": build boxes around a fixed point so that ltrb == reg
    P = reg.shape[0]
    x, y = 68.0, 68.0     # location (8,8) of a stride-8 level: 8*8+4
    boxes = torch.stack([x - reg[:, 0], y - reg[:, 1], x + reg[:, 2], y + reg[:, 3]], 1)
    offs = torch.arange(P + 1).int()
    lab, rt, ctr, _ = HF.fcos_assign(boxes.to(cuda), torch.zeros(P).int().to(cuda), offs.to(cuda), P, [(16, 16)], [8], [[-1, 1e8]], 0.0, 80)
    idx = 8 * 16 + 8
    np.testing.assert_allclose(rt[:, idx].cpu().numpy(), d["reg"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ctr[:, idx].cpu().numpy(), d["ctr"], rtol=2e-5)
