import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_rcnn import _cfg, _data
from slenderobjdet_amd.modeling import build_model
cfg = _cfg(True)
for sz in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
    junk = [torch.full((sz,), float('nan'), device='cuda') for _ in range(24 if sz < (1 << 22) else 3)]
    del junk
torch.manual_seed(0)
model = build_model(cfg); model.train()
data = _data(2, 128, 160, 21, True)
rpn = model.proposal_generator
with torch.no_grad():
    imgs = model.preprocess_image(data)
    feats = model.backbone(imgs.tensor)
    for k, v in feats.items():
        print(k, tuple(v.shape), bool(torch.isfinite(v.float()).all()), float(v.float().abs().max()))
    fl = [feats[f] for f in rpn.in_features]
    lg, dl = rpn.head(fl)
    for a, b in zip(lg, dl):
        print(tuple(a.shape), bool(torch.isfinite(a).all()), float(a.abs().max()), tuple(b.shape), bool(torch.isfinite(b).all()), float(b.abs().max()))
    hw = [(f.shape[1], f.shape[2]) for f in fl]
    anc = rpn.anchor_generator(hw, fl[0].device)
    A, D = 9, 5
    for an, d in zip(anc, dl):
        dd = d[..., :A * D].reshape(-1, D).contiguous()
        p = rpn.box2box_transform.apply_deltas(dd, an.unsqueeze(0).expand(2, -1, -1).reshape(-1, D).contiguous())
        print("props", tuple(p.shape), bool(torch.isfinite(p).all()), float(p.abs().max()))
print("---- training-order replay")
feats = model.backbone(imgs.tensor)
fl = [feats[f] for f in rpn.in_features]
lg, dl = rpn.head(fl)
print("head finite", all(bool(torch.isfinite(x).all()) for x in lg + dl))
anchors = torch.cat(anc).contiguous()
gt = [x["instances"].to("cuda") for x in data]
for g in gt:
    print("gt", g.gt_boxes.tensor)
labels, matched = rpn.label_and_sample_anchors(anchors, gt)
print("labels", [(int((l == 1).sum()), int((l == 0).sum())) for l in labels], "head finite", all(bool(torch.isfinite(x).all()) for x in lg + dl))
gd = rpn.anchor_deltas_for(anchors, matched)
pos = labels == 1
print("gt_deltas finite on pos", bool(torch.isfinite(gd[pos]).all()), "all", bool(torch.isfinite(gd).all()))
from slenderobjdet_amd.modeling.proposal_generator.rpn import _RpnLossFn
out = _RpnLossFn.apply(rpn, labels, gd, *lg, *dl)
print("losses", out, "head finite", all(bool(torch.isfinite(x).all()) for x in lg + dl))
props = rpn.predict_proposals(anc, [x.detach() for x in lg], [x.detach() for x in dl], imgs.image_sizes)
print([len(p) for p in props])
