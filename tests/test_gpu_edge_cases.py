"""Edge cases of the training paths: ragged batches, and a batch in which ONE image has no ground-truth box at all (and, for the dense detectors, the batch
in which NO image has one).  The step must run, give finite losses and gradients, and agree with the oracle where the oracle defines the
case - or fail the way the reference fails (RepPoints)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _strip(data, which):
    from slenderobjdet_amd.structures import Instances

    out = []
    for i, d in enumerate(data):
        d = dict(d)
        if i in which:
            inst = d["instances"]
            e = Instances(inst.image_size)
            bt = type(inst.gt_boxes)
            e.gt_boxes = bt(inst.gt_boxes.tensor[:0])
            e.gt_classes = inst.gt_classes[:0]
            d["instances"] = e
        out.append(d)
    return out


def _cfg(arch):
    if arch == "fcos":
        from bench import make_cfg
        return make_cfg(18)
    if arch == "retinanet":
        from test_gpu_retinanet import _cfg as c
        return c()
    if arch == "reppoints":
        from test_gpu_reppoints import _cfg as c
        return c()
    from test_gpu_rcnn import _cfg as c
    return c(arch == "rrcnn")


@pytest.mark.parametrize("arch", ["fcos", "retinanet", "reppoints", "rcnn", "rrcnn"])
@pytest.mark.parametrize("which", [(1,), (0, 1)], ids=["one_empty", "all_empty"])
def test_training_step_with_images_without_boxes(cuda, arch, which):
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = _cfg(arch)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    if arch in ("rcnn", "rrcnn"):
        from test_gpu_rcnn import _data
        data = _data(2, 96, 128, 21, arch == "rrcnn")
    else:
        data = synthetic_batch(2, 192, 256, 7, device="cuda")
    data = _strip(data, which)
    assert sum(len(d["instances"]) for d in data) == (0 if len(which) == 2 else len(data[0]["instances"]))
    if arch == "reppoints":     # the reference's point matcher refuses the case (matchers/rep_matcher.py:38-39): same error, same message
        with pytest.raises(ValueError, match="No gt or bboxes"):
            model(data)
        return
    losses = model(data)
    total = sum(losses.values())
    assert all(torch.isfinite(v).all() for v in losses.values()), {k: float(v) for k, v in losses.items()}
    opt.zero_grad()
    model.arena.begin_backward(); total.backward(); model.arena.finish_backward()
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
    opt.step()
    if arch == "fcos":      # the oracle defines the case (fcosv2.py:116-147: normalisers clamped to 1, empty selections sum to 0)
        from oracle.model import OracleFCOS
        torch.manual_seed(0)
        m2 = build_model(cfg)
        m2.train()
        ref = OracleFCOS.from_hip_model(m2, emulate_bf16=True).losses([{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data])
        got = m2(data)
        for k, v in ref.items():
            a, b = float(got[k].detach()), float(v)
            assert abs(a - b) <= 2e-3 * max(abs(b), 1e-3), (k, a, b)
        if len(which) == 2:
            assert float(got["reg_loss"].detach()) == 0.0 and float(got["centerness_loss"].detach()) == 0.0


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_ragged_batch_matches_oracle(cuda, precision):
    """Images of DIFFERENT sizes in one batch (ImageList.from_tensors pads to the largest, rounded up to the size divisibility; the
    targets use the padded grid, fcosv2.py:63-102): FCOS losses against the oracle, 1e-3 for the bf16 product and 2e-5 in the fp32 mode."""
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF
    from slenderobjdet_amd.modeling import build_model

    prev = HF.set_precision(precision)
    try:
        cfg = make_cfg(18)
        torch.manual_seed(0)
        model = build_model(cfg)
        model.train()
        data = [synthetic_batch(1, 192, 256, 3, device="cuda")[0], synthetic_batch(1, 150, 203, 4, device="cuda")[0], synthetic_batch(1, 97, 256, 5, device="cuda")[0]]
        assert len({tuple(d["image"].shape[-2:]) for d in data}) == 3
        got = model(data)
        ref = OracleFCOS.from_hip_model(model, emulate_bf16=precision == "bf16").losses(
            [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data])
        tol = 1e-3 if precision == "bf16" else 2e-5
        for k, v in ref.items():
            a, b = float(got[k].detach()), float(v.detach())
            assert abs(a - b) <= tol * max(abs(b), 1e-3), (k, a, b)
    finally:
        HF.set_precision(prev)


def test_operands_of_2_gib_are_refused_not_wrapped(cuda):
    """Maximum sizes: the kernels address their operands with 32-bit byte offsets, so an operand of 2 GiB or more must be REFUSED
    (SOD_ESIZE) by the C ABI before anything is launched - never computed with wrapped offsets.  The shapes are only claimed
    (``x_shape``), the refusal comes from the size check of sod_conv2d_fwd."""
    from slenderobjdet_amd._C import SlenderHipError
    from slenderobjdet_amd.layers import functional as HF

    x = torch.zeros(1, 8, 8, 64, device=cuda, dtype=torch.bfloat16)
    w = torch.zeros(8, 1, 1, 64, device=cuda, dtype=torch.bfloat16)
    ok = HF.conv2d_fwd(x, w)                                   # the same call at its real size works
    assert tuple(ok.shape) == (1, 8, 8, 8)
    with pytest.raises(SlenderHipError, match="SOD_ESIZE"):
        HF.conv2d_fwd(x, w, x_shape=(256, 256, 256, 64))      # 2.1 GB of claimed input
    torch.cuda.synchronize()
