"""Pins the CPU oracle against golden vectors produced by the REFERENCE's own Python (tests/golden/make_golden.py).
Runs on CPU (-m "not gpu")."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fcos_targets as ot
from oracle import losses as ol
from oracle.model import OracleFCOS

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


def test_meta_lists_every_fixture():
    meta = json.load(open(os.path.join(G, "meta.json")))
    assert sorted(meta) == sorted(f for f in os.listdir(G) if f.endswith(".npz"))


@pytest.mark.parametrize("lt", ["iou", "linear_iou", "giou"])
def test_iou_loss_matches_reference(lt):
    d = _load("iou_loss.npz")
    pred = torch.tensor(d["pred"], requires_grad=True)
    tgt, w = torch.tensor(d["target"]), torch.tensor(d["weight"])
    loss = ol.iou_loss_ltrb(pred, tgt, w, lt)
    (g,) = torch.autograd.grad(loss, pred)
    np.testing.assert_allclose(loss.detach().numpy(), d[f"loss_{lt}"], rtol=1e-6)
    np.testing.assert_allclose(g.numpy(), d[f"grad_{lt}"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ol.iou_loss_ltrb(pred.detach(), tgt, None, lt).numpy(), d[f"loss_noweight_{lt}"], rtol=1e-6)


def test_centerness_matches_reference():
    d = _load("centerness.npz")
    np.testing.assert_array_equal(ol.centerness_targets(torch.tensor(d["reg"])).numpy(), d["ctr"])


def test_locations_match_reference():
    d = _load("locations.npz")
    locs = ot.locations([tuple(s) for s in d["shapes"]], list(d["strides"]))
    assert sum(len(l) for l in locs) == 22400   # L of SURVEY.md §2.3 at 800x1344
    for i, l in enumerate(locs):
        np.testing.assert_array_equal(l.numpy(), d[f"loc{i}"])


@pytest.mark.parametrize("radius", [0.0, 1.5])
def test_target_assignment_bit_exact(radius):
    d = _load("fcos_targets.npz")
    hw = [tuple(int(v) for v in s) for s in d["shapes"]]
    boxes = [torch.tensor(d[f"boxes{i}"]) for i in range(3)]
    classes = [torch.tensor(d[f"classes{i}"]) for i in range(3)]
    lab, reg = ot.targets_for_batch(hw, [int(s) for s in d["strides"]], boxes, classes, radius, 80)
    np.testing.assert_array_equal(lab.numpy(), d[f"labels_r{radius}"])
    np.testing.assert_array_equal(reg.numpy(), d[f"reg_r{radius}"])
    if radius > 0:
        assert (lab[2] == 80).all(), "first-box-centred-at-x==0 quirk (fcos/utils.py:121-122) must be reproduced"


def _oracle_from_golden(d):
    """Build an OracleFCOS head from the reference's parameter names (cls_tower.{0,3,6,9} conv, {1,4,7,10} GN, ...)."""
    ctr_on_reg, norm_reg, iou_id, radius = [float(v) for v in d["cfg"]]
    ctr_on_reg, norm_reg = bool(ctr_on_reg), bool(norm_reg)
    P = lambda k: torch.tensor(d["param::" + k]).clone().requires_grad_(True)
    p = {}
    for tower, dst in (("cls_tower", "head.cls_tower"), ("bbox_tower", "head.bbox_tower")):
        for i in range(4):
            p[f"{dst}.{i}.conv.weight"], p[f"{dst}.{i}.conv.bias"] = P(f"{tower}.{3 * i}.weight"), P(f"{tower}.{3 * i}.bias")
            p[f"{dst}.{i}.gn.weight"], p[f"{dst}.{i}.gn.bias"] = P(f"{tower}.{3 * i + 1}.weight"), P(f"{tower}.{3 * i + 1}.bias")
    cw, cb, bw, bb = P("cls_logits.weight"), P("cls_logits.bias"), P("bbox_pred.weight"), P("bbox_pred.bias")
    tw, tb = P("centerness.weight"), P("centerness.bias")
    leaves = dict(cw=cw, cb=cb, bw=bw, bb=bb, tw=tw, tb=tb)
    if ctr_on_reg:
        p["head.cls_pred.weight"], p["head.cls_pred.bias"] = cw, cb
        p["head.box_pred.weight"], p["head.box_pred.bias"] = torch.cat([bw, tw]), torch.cat([bb, tb])
    else:
        p["head.cls_pred.weight"], p["head.cls_pred.bias"] = torch.cat([cw, tw]), torch.cat([cb, tb])
        p["head.box_pred.weight"], p["head.box_pred.bias"] = bw, bb
    scales = [P(f"scales.{i}.scale") for i in range(5)]
    p["head.scales"] = torch.cat(scales)
    cfg = dict(num_classes=80, strides=[8, 16, 32, 64, 128], radius=radius, alpha=0.25, gamma=2.0,
               iou_type=["iou", "linear_iou", "giou"][int(iou_id)], norm_reg=norm_reg, ctr_on_reg=ctr_on_reg,
               kc=80 + (0 if ctr_on_reg else 1), num_convs=4)
    return OracleFCOS(p, {}, cfg), leaves, scales


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fcos_head_and_losses_match_reference(tag):
    d = _load(f"fcos_head_losses_{tag}.npz")
    o, leaves, scales = _oracle_from_golden(d)
    feats = [torch.tensor(d[f"feat{i}"]) for i in range(5)]
    cls, box, ctr = o._head(feats)
    # forward of FCOSHead: per-level outputs flattened the way permute_and_concat does
    N = 2
    hw = [f.shape[2:] for f in feats]
    box_ref = torch.cat([torch.tensor(d[f"bbox{i}"]).permute(0, 2, 3, 1).reshape(N, -1, 4) for i in range(5)], 1).reshape(-1, 4)
    ctr_ref = torch.cat([torch.tensor(d[f"ctr{i}"]).permute(0, 2, 3, 1).reshape(N, -1) for i in range(5)], 1).reshape(-1)
    np.testing.assert_allclose(box.detach().numpy(), box_ref.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ctr.detach().numpy(), ctr_ref.numpy(), rtol=2e-5, atol=1e-6)
    L = sum(h * w for h, w in hw)
    lvl1 = torch.tensor(d["logits1"]).permute(0, 2, 3, 1).reshape(N, -1, 80)
    off = hw[0][0] * hw[0][1]
    np.testing.assert_allclose(cls.detach().reshape(N, L, 80)[:, off:off + lvl1.shape[1]].numpy(), lvl1.numpy(), rtol=2e-5, atol=1e-6)
    # targets + losses
    boxes = [torch.tensor(d["gtb0"]), torch.tensor(d["gtb1"])]
    classes = [torch.tensor(d["gtc0"]), torch.tensor(d["gtc1"])]
    labels, reg_t = ot.targets_for_batch([tuple(s) for s in hw], o.c["strides"], boxes, classes, o.c["radius"], 80)
    np.testing.assert_array_equal(labels.numpy(), d["labels"])
    np.testing.assert_array_equal(reg_t.numpy(), d["reg_targets"])
    losses = ol.fcos_losses(labels.reshape(-1), reg_t.reshape(-1, 4), cls, box, ctr, 80, 0.25, 2.0, o.c["iou_type"])
    for k in ("cls_loss", "reg_loss", "centerness_loss"):
        np.testing.assert_allclose(losses[k].detach().numpy(), d["loss::" + k], rtol=1e-5)
    # gradients w.r.t. the reference's parameters
    total = sum(losses.values())
    names = {"cw": "cls_logits.weight", "cb": "cls_logits.bias", "bw": "bbox_pred.weight", "bb": "bbox_pred.bias",
             "tw": "centerness.weight", "tb": "centerness.bias"}
    wanted = list(leaves.values()) + scales + [o.p["head.cls_tower.0.conv.weight"], o.p["head.bbox_tower.3.gn.weight"]]
    grads = torch.autograd.grad(total, wanted)
    for (k, _), g in zip(leaves.items(), grads):
        ref = d["grad::" + names[k]]
        np.testing.assert_allclose(g.numpy(), ref, rtol=2e-4, atol=2e-6 * max(1.0, float(np.abs(ref).max())))
    for i in range(5):
        np.testing.assert_allclose(grads[6 + i].numpy(), d[f"grad::scales.{i}.scale"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(grads[11].numpy(), d["grad::cls_tower.0.weight"], rtol=5e-4, atol=1e-6)
    np.testing.assert_allclose(grads[12].numpy(), d["grad::bbox_tower.10.weight"], rtol=5e-4, atol=1e-6)


def test_deform_conv_kat_zero_offsets_is_plain_conv():
    """The reference's only assert-based test (tests/test_deformable_conv.py:85-87): zero offsets == F.conv2d."""
    d = _load("deform_conv_kat.npz")
    y = torch.nn.functional.conv2d(torch.tensor(d["input"]), torch.tensor(d["weight"]), padding=1)
    np.testing.assert_allclose(y.numpy(), d["y_conv"], atol=1e-5)
    np.testing.assert_allclose(d["y_dconv_zero"], d["y_conv"], atol=1e-5)
    from oracle.deform_conv import deform_conv2d

    for key, off in (("y_dconv_zero", "offsets_2"), ("y_dconv_1", "offsets_1")):
        got = deform_conv2d(torch.tensor(d["input"]), torch.tensor(d[off]), torch.tensor(d["weight"]), stride=1, pad=1)
        np.testing.assert_allclose(got.numpy(), d[key], atol=1e-5)


def test_chaos100_fixture_matches_the_oracle_sources():
    """tests/golden/chaos100.json holds 100-iteration trajectories of oracle/model.py that tests/test_gpu_parity100.py compares the product
    with (only the first six iterations are re-run live there): the file records the sha256 of the sources it was computed from, and a
    change of any of them without `python tests/golden/make_chaos100.py stamp` (re-checks the first iterations) or a regeneration fails."""
    import hashlib
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    chaos = json.load(open(os.path.join(root, "tests", "golden", "chaos100.json")))
    stored = chaos.get("sources_sha256") or {}
    assert stored, "fixture without source hashes"
    changed = [f for f in stored if stored[f] != hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest()]
    assert not changed, f"chaos100.json is stale for {changed}: python tests/golden/make_chaos100.py stamp"
