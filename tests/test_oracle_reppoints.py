"""Pins the RepPoints oracle (oracle/reppoints.py) against golden vectors from the REFERENCE's own Python
(tests/golden/make_golden_reppoints.py).  Runs on CPU (-m "not gpu")."""
import os

import numpy as np
import pytest
import torch

from oracle import reppoints as orp

G = os.path.join(os.path.dirname(__file__), "golden")
MODES = ["points", "nearest_points", "inside"]


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


@pytest.mark.parametrize("mode", MODES)
def test_matchers_bit_exact(mode):
    d = _load("reppoints_matchers.npz")
    centers, strides = orp.center_grid([tuple(x) for x in d["hw"]], list(d["strides"]))
    for i in range(int(d["num_cases"])):
        obj, lab = orp.MATCHERS[mode](centers, strides, torch.tensor(d[f"boxes{i}"]))
        np.testing.assert_array_equal(obj.numpy().astype(np.int8), d[f"{mode}_obj{i}"], err_msg=f"case {i}")
        np.testing.assert_array_equal(lab.numpy(), d[f"{mode}_box{i}"], err_msg=f"case {i}")


def test_points2bbox_matches_reference():
    d = _load("reppoints_losses.npz")
    hw, strides = [tuple(x) for x in d["hw"]], list(d["strides"])
    for key, out in (("oi", "init_boxes"), ("or", "refine_boxes")):
        deltas = [torch.tensor(d[f"{key}{l}"]) for l in range(len(hw))]
        np.testing.assert_array_equal(orp.points2bbox(deltas, hw, strides, [1, 2, 4, 8, 16]).numpy(), d[out])


@pytest.mark.parametrize("mode", MODES)
def test_ground_truth_and_losses_match_reference(mode):
    d = _load("reppoints_losses.npz")
    hw, strides = [tuple(x) for x in d["hw"]], list(d["strides"])
    centers, st = orp.center_grid(hw, strides)
    init_boxes, refine_boxes = torch.tensor(d["init_boxes"]), torch.tensor(d["refine_boxes"])
    gtb = [torch.tensor(d[f"gt_boxes{i}"]) for i in range(2)]
    gtc = [torch.tensor(d[f"gt_classes{i}"]) for i in range(2)]
    sizes = [tuple(int(v) for v in s) for s in d["image_sizes"]]
    obj, ib, cl, rb = orp.get_ground_truth(centers, st, init_boxes, gtb, gtc, sizes, 80, mode)
    np.testing.assert_array_equal(obj.numpy().astype(np.int8), d[f"{mode}_obj"])
    np.testing.assert_array_equal(ib.numpy(), d[f"{mode}_init"])
    np.testing.assert_array_equal(cl.numpy().astype(np.int16), d[f"{mode}_cls"])
    np.testing.assert_array_equal(rb.numpy(), d[f"{mode}_refine"])
    assert (cl == -1).any() and (cl == 80).any() and ((cl >= 0) & (cl < 80)).any()       # all three label kinds are exercised
    lg = torch.tensor(d["logits"]).float().requires_grad_(True)
    b1, b2 = init_boxes.clone().requires_grad_(True), refine_boxes.clone().requires_grad_(True)
    out, nrm = orp.losses(lg, b1, b2, obj, ib, cl, rb, st, 80, 0.25, 2.0, 20)
    got = np.array([float(out[k]) for k in ("loss_cls", "loss_localization_init", "loss_localization_refine")])
    np.testing.assert_allclose(got, d[f"{mode}_losses"], rtol=1e-6)
    np.testing.assert_allclose(nrm, float(d[f"{mode}_normalizer"]), rtol=1e-7)
    gl, g1, g2 = torch.autograd.grad(sum(out.values()), (lg, b1, b2))
    np.testing.assert_allclose(g1.numpy(), d[f"{mode}_grad_init"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(g2.numpy(), d[f"{mode}_grad_refine"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(gl.sum(-1).numpy(), d[f"{mode}_grad_logits_sum"], rtol=1e-4, atol=1e-7)


def test_topk_matcher_matches_reference():
    """slender_det/modeling/matchers/topk_matcher.py (the matcher RPNWNM selects with MODEL.RPN.MATCHER.TYPE = "TopK")."""
    from oracle import detection as od

    d = _load("topk_matcher.npz")
    q = od.pairwise_iou(torch.tensor(d["gt"]), torch.tensor(d["anchors"]))
    np.testing.assert_array_equal(q.numpy(), d["quality"])
    m, lab = od.topk_matcher(q, [0.3, 0.7], [0, -1, 1], 10)
    np.testing.assert_array_equal(m.numpy(), d["matches"])
    np.testing.assert_array_equal(lab.numpy(), d["labels"])
    assert (lab == 1).sum() >= 9 * 10 - 20 and (lab == 0).any()
