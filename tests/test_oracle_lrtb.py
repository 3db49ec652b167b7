"""Pins the LRTBHead oracle (oracle/lrtb.py) against golden vectors produced by the REFERENCE's own head run on CPU
(tests/golden/make_golden_reppoints.py).  Runs on CPU (-m "not gpu")."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.lrtb import OracleLRTBHead

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["empty", "sup", "unsup", "topk"])
def test_lrtb_head_losses_and_gradients_match_reference(tag):
    d = {k: v for k, v in np.load(os.path.join(G, f"lrtb_head_{tag}.npz")).items()}
    c = json.loads(str(d["cfg"]))
    cfg = dict(fa=c["fa"], res=c["res"], K=80, gmul=0.1, strides=[8, 16, 32, 64, 128], norm_reg=c["norm_reg"], ctr_on_loc=c["ctr_on_loc"],
               iou_type=c["iou"], slender=c["slender"], radius=c["radius"], w=(1.0, 0.5, 1.0), alpha=0.25, gamma=2.0)
    o = OracleLRTBHead.from_reference_arrays(d, cfg)
    feats = [torch.tensor(d[f"feat{l}"].astype(np.float32)) for l in range(5)]
    gtb = [torch.tensor(d[f"gt_boxes{i}"]) for i in range(2)]
    gtc = [torch.tensor(d[f"gt_classes{i}"]) for i in range(2)]
    out = o.losses(feats, gtb, gtc, topk=bool(c.get("topk")))
    got = np.array([float(out[k].detach()) for k in ("loss_cls", "centerness_loss", "loss_loc_init", "loss_loc_refine")])
    np.testing.assert_allclose(got, d["losses"], rtol=3e-5)
    # gradients of the un-fused reference convs = row blocks of the fused product-style convs
    K = 80
    gc, gb, gi = torch.autograd.grad(sum(out.values()), [o.p["cls_pred.conv.weight"], o.p["box_pred.conv.weight"], o.p["loc_init_out.conv.weight"]])
    np.testing.assert_allclose(gc[:K].numpy(), d["grad:cls_out.weight"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(gb[:4].numpy(), d["grad:loc_refine_out.weight"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(gi.numpy(), d["grad:loc_init_out.weight"], rtol=2e-3, atol=1e-6)
    ctn = gb[4:5] if c["ctr_on_loc"] else gc[K:K + 1]
    np.testing.assert_allclose(ctn.numpy(), d["grad:ctn_out.weight"], rtol=2e-3, atol=1e-6)
    assert float(d["gradnorm:loc_init_out.weight"]) > 0 and float(d["gradnorm:loc_refine_out.weight"]) > 0
