"""Pins the PointSetHead oracle (oracle/pointset.py) against golden vectors produced by the REFERENCE's own Python
(tests/golden/make_golden_reppoints.py: the reference head built and run on CPU).  Runs on CPU (-m "not gpu")."""
import os

import numpy as np
import pytest
import torch

from oracle.pointset import OraclePointSetHead

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = {"empty": "Empty", "sup": "Supervised Offset", "unsup": "Unsupervised Offset", "partial": "Empty", "moment": "Empty"}
METHODS = {"partial": "partial_minmax", "moment": "moment"}      # TRANSFORM_METHOD of the fixture (default "minmax")


@pytest.mark.parametrize("tag", sorted(CASES))
def test_pointset_head_losses_and_gradients_match_reference(tag):
    d = {k: v for k, v in np.load(os.path.join(G, f"pointset_head_{tag}.npz")).items()}
    o = OraclePointSetHead.from_reference_arrays(d, CASES[tag], bool(d["res_refine"]), METHODS.get(tag, "minmax"))
    feats = [torch.tensor(d[f"feat{l}"].astype(np.float32)) for l in range(5)]
    gtb = [torch.tensor(d[f"gt_boxes{i}"]) for i in range(2)]
    gtc = [torch.tensor(d[f"gt_classes{i}"]) for i in range(2)]
    out = o.losses(feats, gtb, gtc)
    got = np.array([float(out[k].detach()) for k in ("loss_cls", "loss_pts_init", "loss_pts_refine")])
    np.testing.assert_allclose(got, d["losses"], rtol=2e-5)
    names = {"cls_out.weight": "logits.weight", "loc_refine_out.weight": "offsets_refine.weight", "loc_init_out.weight": "loc_init_out.conv.weight"}
    if tag == "moment":
        names["moment_transfer"] = "moment_transfer"
    grads = torch.autograd.grad(sum(out.values()), [o.p[v] for v in names.values()])
    for (ref_name, _), g in zip(names.items(), grads):
        np.testing.assert_allclose(g.numpy(), d["grad:" + ref_name], rtol=2e-3, atol=1e-6, err_msg=ref_name)
        np.testing.assert_allclose(float(g.norm()), float(d["gradnorm:" + ref_name]), rtol=1e-4)
