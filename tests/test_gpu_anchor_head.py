"""GPU parity of AblationMetaArch + AnchorHead against oracle/anchor_head.py, which is pinned to the reference's own
head by tests/test_oracle_anchor_head.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(fa, box_loss):
    from bench import make_cfg

    cfg = make_cfg(18)
    cfg.MODEL.META_ARCHITECTURE = "AblationMetaArch"
    cfg.MODEL.BACKBONE.NAME = "build_retinanet_resnet_fpn_backbone"
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
    m = cfg.MODEL.META_ARCH
    m.NAME, m.FEAT_ADAPTION, m.BBOX_REG_LOSS_TYPE, m.RES_REFINE = "AnchorHead", fa, box_loss, False
    m.IOU_THRESHOLDS, m.IOU_LABELS = [0.4, 0.5], [0, -1, 1]
    return cfg


@pytest.mark.parametrize("fa,box_loss", [("none", "giou"), ("supervised", "smooth_l1"), ("unsupervised", "giou"), ("split", "smooth_l1")])
def test_anchor_head_vs_oracle(cuda, fa, box_loss):
    from oracle import anchor_head as oah
    from oracle import rcnn as orc
    from oracle import reppoints as orp
    from oracle import retinanet as orn
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg(fa, box_loss)
    torch.manual_seed(14)
    model = build_model(cfg)
    model.train()
    head = model.head
    with torch.no_grad():
        head.loc_init_out.conv.weight.mul_(5.0)
        head.bbox_pred.weight.mul_(5.0)
    model.arena.bump()
    opt = build_optimizer(cfg, model)
    data = synthetic_batch(2, 192, 256, 17, device="cuda")
    got = model(data)
    assert set(got) == {"loss_cls", "loss_loc_init", "loss_loc_refine"}
    N, K, A = 2, 80, 9
    with torch.no_grad():
        imgs = model.preprocess_image(data)
        feats = model.backbone(imgs.tensor)
        feats = [feats[f] for f in head.in_features]
        cls_t, box_t, raw = head.run_head(feats)
        cls_buf, box_buf, hw, _ = head.predict(cls_t, box_t)
        init = head.init_boxes(raw, hw).cpu()
    P = cls_buf.shape[1]
    logits, pdel = cls_buf.cpu().view(N, P * A, K), box_buf.cpu()[..., :A * 4].reshape(N, P * A, 4)
    gtb = [d["instances"].gt_boxes.tensor.cpu() for d in data]
    gtc = [d["instances"].gt_classes.cpu() for d in data]
    sizes = [tuple(d["image"].shape[-2:]) for d in data]
    anchors = torch.cat(orc.anchors(hw, head.strides, head.anchor_sizes, head.anchor_ratios))
    # 1. targets: anchor labels and nearest-point init labels bit-exact
    gl, gb = orn.label_anchors(anchors, gtb, gtc, [0.4, 0.5], [0, -1, 1], K)
    lab_h, _, obj_h, init_lab_h = (t.cpu() for t in head.last_targets)
    assert torch.equal(lab_h.long(), gl)
    centers, st = orp.center_grid(hw, head.strides)
    li_ref, obj_ref, lab_ref = oah.init_losses(init, centers, st, gtb, sizes)
    assert torch.equal(obj_h.float(), obj_ref) and torch.equal(init_lab_h[obj_ref > 0], lab_ref[obj_ref > 0]) and (obj_ref > 0).sum() > 0
    # 2. the three losses from the product path's own predictions
    ref, nrm = orn.losses(anchors, logits, pdel, gl, gb, K, 0.25, 2.0, 0.11, (1.0, 1.0, 1.0, 1.0), 100.0, box_reg_loss_type=box_loss)
    exp = {"loss_cls": ref["loss_cls"], "loss_loc_init": li_ref * 0.5, "loss_loc_refine": ref["loss_box_reg"]}
    for k, b in exp.items():
        a, b = float(got[k].detach()), float(b)
        assert abs(a - b) <= 3e-4 * max(abs(b), 1e-3), (k, a, b)
    assert abs(float(head.loss_normalizer) - nrm) < 1e-3
    # 3. head forward vs the oracle head (bf16 storage emulated) on the same features
    o = oah.OracleAnchorHead.from_hip_head(head, emulate_bf16=True)
    with torch.no_grad():
        l2, d2, i2, _ = o.forward([f.float().cpu().permute(0, 3, 1, 2) for f in feats])
    for name, a, b in (("logits", logits, l2), ("deltas", pdel, d2), ("init", init, i2)):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
        assert err < 3e-2, (name, err)
    # 4. gradients and a few steps
    total = sum(got.values())
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    for name, p in head.named_parameters():
        assert torch.isfinite(p.grad).all(), name
    assert head.loc_init_out.conv.weight.grad[:4].abs().sum() > 0 and (head.loc_init_out.conv.weight.grad[4:] == 0).all()
    assert head.bbox_pred.weight.grad[:36].abs().sum() > 0 and (head.bbox_pred.weight.grad[36:] == 0).all()
    for g in opt.param_groups:
        g["lr"] = 0.002
    ls = []
    for _ in range(6):
        losses = model(data)
        t = sum(losses.values())
        opt.zero_grad()
        model.arena.begin_backward(); t.backward(); model.arena.finish_backward()
        opt.step()
        ls.append(float(t.detach()))
    assert all(v == v for v in ls) and ls[-1] < ls[0], ls
    model.eval()
    for d in data:
        d.pop("instances")
    out = model(data)
    assert len(out) == 2 and "instances" in out[0]
