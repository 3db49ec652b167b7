"""Pins the AnchorHead oracle (oracle/anchor_head.py) against golden vectors produced by the REFERENCE's own head run on CPU
(tests/golden/make_golden_reppoints.py).  Runs on CPU (-m "not gpu")."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.anchor_head import OracleAnchorHead

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["none", "sup", "unsup"])
def test_anchor_head_losses_and_gradients_match_reference(tag):
    d = {k: v for k, v in np.load(os.path.join(G, f"anchor_head_{tag}.npz")).items()}
    c = json.loads(str(d["cfg"]))
    sizes = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
    cfg = dict(fa=c["fa"], K=80, A=9, gmul=0.1, strides=[8, 16, 32, 64, 128], sizes=sizes, ratios=[[0.5, 1.0, 2.0]], thresholds=[0.4, 0.5],
               labels=[0, -1, 1], weights=(1.0, 1.0, 1.0, 1.0), box_loss=c["box_loss"], alpha=0.25, gamma=2.0, w=(1.0, 0.5, 1.0))
    o = OracleAnchorHead.from_reference_arrays(d, cfg)
    feats = [torch.tensor(d[f"feat{l}"].astype(np.float32)) for l in range(5)]
    gtb = [torch.tensor(d[f"gt_boxes{i}"]) for i in range(2)]
    gtc = [torch.tensor(d[f"gt_classes{i}"]) for i in range(2)]
    sizes_img = [tuple(int(v) for v in s) for s in d["image_sizes"]]
    out, nrm = o.losses(feats, gtb, gtc, sizes_img)
    got = np.array([float(out[k].detach()) for k in ("loss_cls", "loss_loc_init", "loss_loc_refine")])
    np.testing.assert_allclose(got, d["losses"], rtol=5e-5)
    np.testing.assert_allclose(nrm, float(d["normalizer"]), rtol=1e-6)
    gb, gi = torch.autograd.grad(sum(out.values()), [o.p["bbox_pred.weight"], o.p["loc_init_out.conv.weight"]])
    np.testing.assert_allclose(gb.numpy(), d["grad:loc_refine_out.weight"], rtol=5e-3, atol=1e-6)
    np.testing.assert_allclose(gi.numpy(), d["grad:loc_init_out.weight"], rtol=5e-3, atol=1e-6)
