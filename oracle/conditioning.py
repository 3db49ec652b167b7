"""TEST INFRASTRUCTURE (like everything under oracle/): a well-conditioned test problem for whole-model gradient comparisons.

A random-init ResNet whose FrozenBN layers are the identity (what ``build_model`` gives without a checkpoint) doubles the activation
variance in every residual block - x256 in standard deviation over the 16 blocks of R50 - and its weight gradients are then dominated by
a few huge pre-activations: ONE ReLU decision that falls differently under another fp32 summation order moves a deep weight gradient by
1e-2 of its norm, in the CPU oracle against float64 as much as in the HIP kernels.  The reference never trains
in that regime: it starts from an ImageNet checkpoint whose FrozenBatchNorm2d statistics normalise every convolution's output
(detectron2 ResNet).  No checkpoint can be fetched here, so ``calibrate_frozen_bn`` manufactures the property a
checkpoint has: it walks the backbone once in float64 on the CPU with the model's own random weights and sets every FrozenBN's
running mean / variance to the statistics of its convolution's output ON THE GIVEN BATCH (weight 1 - ``gamma_last`` for the last norm
of a residual branch -, bias 0).  Activations are O(1) everywhere afterwards, a single ReLU flip changes a weight gradient by ~1e-7 of
its norm, and two fp32 implementations can be held to 1e-3 per tensor again instead of to a cap (tests/test_gpu_f32_mode.py,
tests/test_gpu_parity100.py).  The buffers are written into the product model in place; oracles built from it afterwards see them."""
import torch
import torch.nn.functional as F

from .model import OracleFCOS


class _Calibrator(OracleFCOS):
    """OracleFCOS whose FrozenBN convolutions fix their statistics from the activations that reach them (forward order)."""

    modules = None
    gamma_last = 0.5
    log = None

    def _set(self, name, raw):
        m = self.modules[name]
        mean = raw.mean((0, 2, 3))
        var = raw.var((0, 2, 3), unbiased=False).clamp_min(1e-8)
        last = name.endswith(".conv3") or (name.endswith(".conv2") and not self.c["bottleneck"])
        gamma = self.gamma_last if last else 1.0
        with torch.no_grad():
            m.bn_running_mean.copy_(mean.to(m.bn_running_mean))
            m.bn_running_var.copy_(var.to(m.bn_running_var))
            m.bn_weight.fill_(gamma)
            m.bn_bias.zero_()
        # exactly what OracleFCOS._collect derives from the buffers just written (fp32 buffers, eps 1e-5)
        scale = m.bn_weight.float().cpu() * torch.rsqrt(m.bn_running_var.float().cpu() + 1e-5)
        shift = m.bn_bias.float().cpu() - m.bn_running_mean.float().cpu() * scale
        self.b[name + ".scale"], self.b[name + ".shift"] = scale.to(raw.dtype), shift.to(raw.dtype)
        if self.log is not None:
            self.log.append((name, float(raw.std()), float(mean.abs().max())))

    def _conv(self, name, x, stride=1, pad=0, relu=False, res=None, out_f32=False):
        if name + ".scale" in self.b:
            raw = F.conv2d(x, self.p[name + ".weight"], None, stride=stride, padding=pad, groups=self.c.get("groups", {}).get(name, 1))
            self._set(name, raw)
        return super()._conv(name, x, stride, pad, relu, res, out_f32)

    def _dcn(self, name, x, om, relu=False):
        if name + ".scale" in self.b:
            self.b[name + ".scale"] = torch.ones_like(self.b[name + ".scale"])
            self.b[name + ".shift"] = torch.zeros_like(self.b[name + ".shift"])
            self._set(name, super()._dcn(name, x, om, relu=False))
        return super()._dcn(name, x, om, relu)


@torch.no_grad()
def calibrate_frozen_bn(model, batched_inputs, gamma_last=0.5, log=None):
    """Sets the FrozenBN buffers of ``model``'s bottom-up ResNet (any meta-architecture of this package whose backbone the oracle can
    walk) from ONE float64 pass over ``batched_inputs``.  Returns the number of norms set."""
    cpu = [{"image": d["image"].cpu()} for d in batched_inputs]
    cal = _Calibrator.from_hip_model(model) if hasattr(model, "head") and hasattr(model.head, "scales") else _Calibrator(*_backbone_only(model))
    cal.double()
    cal.modules = dict(model.named_modules())
    cal.gamma_last = gamma_last
    cal.log = log
    before = len(cal.b)
    x = cal.preprocess(cpu)
    cal._bottom_up(x[0] if isinstance(x, (tuple, list)) else x)
    return before // 2


def _backbone_only(model):
    """(params, buffers, cfg) for models without an FCOS head (RetinaNet, RepPoints, R-CNN): only the bottom-up part is walked."""
    params, buffers, dcn, groups = OracleFCOS._collect(model)
    return params, buffers, OracleFCOS._backbone_cfg(model, params, dcn, groups), False


def gradient_envelope(params, loss_fn, tau=2e-5):
    """``loss_fn()`` -> dict of losses on an oracle whose trainable tensors are ``params`` (name -> tensor).  Runs ONE forward pass
    inside a ReluBand context and three backward passes; returns (losses, g, spread, info):
    g[name] = the natural gradient, spread[name] = ||g_on - g|| + ||g_off - g|| (what switching every undecided ReLU unit on / off does
    to that tensor), info = {"undecided": count, "units": total}.  A comparison against another correct implementation of the same
    network may be off by up to ~spread on a tensor, and by rounding only (1e-5 ... 1e-4 of the norm) where spread is ~0."""
    from .nn import ReluBand

    names, tensors = list(params.keys()), list(params.values())
    ReluBand.active = {"tau": float(tau), "mode": 0, "count": 0, "units": 0}
    try:
        losses = loss_fn()
        total = sum(losses.values())
        out = {}
        for mode in (0, 1, -1):
            ReluBand.active["mode"] = mode
            out[mode] = torch.autograd.grad(total, tensors, retain_graph=mode != -1)
        info = {"undecided": ReluBand.active["count"], "units": ReluBand.active["units"], "tau": float(tau)}
    finally:
        ReluBand.active = None
    g = dict(zip(names, out[0]))
    spread = {n: float((a - c).norm() + (b - c).norm()) for n, a, b, c in zip(names, out[1], out[-1], out[0])}
    return {k: float(v.detach()) for k, v in losses.items()}, g, spread, info


class ProductReluTap:
    """Records the ReLU decisions of the HIP model's fp32 validation mode during one forward pass (layers/functional_f32.RELU_TAP) and
    turns them into the ``{(position, i): mask}`` dictionary ``oracle.nn.ForcedMasks`` takes.  Positions are module names: a fused
    conv + ReLU is identified by the address of the weight operand it was launched with, GroupNorm + ReLU by its gamma, the one bare
    ReLU of the FPN (LastLevelP6P7: p7 = conv(relu(p6)), fpn.py:94-115) by ``bare_relu_keys`` in call order.

        with ProductReluTap() as tap:
            losses = model(data); ... backward ...
        masks, unmatched = tap.masks_for(model)
        st = ForcedMasks.begin(masks); ...oracle forward / backward...; ForcedMasks.end()
    """

    def __init__(self, bare_relu_keys=("backbone.top_block.p7:in",)):
        self.rec = []
        self.bare = list(bare_relu_keys)

    def __enter__(self):
        # both precisions of the product report through the same observer signature: the fp32 validation mode (functional_f32) and the
        # bf16 product path (functional; round 5) - whichever runs inside the context is recorded
        from slenderobjdet_amd.layers import functional as HF
        from slenderobjdet_amd.layers import functional_f32 as F32

        self._mods = (F32, HF)
        tap = lambda kind, key, y: self.rec.append((kind, int(key), (y.detach() > 0).cpu()))      # noqa: E731
        for m in self._mods:
            m.RELU_TAP = tap
        return self

    def __exit__(self, *exc):
        for m in self._mods:
            m.RELU_TAP = None

    def masks_for(self, model):
        conv_names, gn_names = {}, {}
        for name, m in model.named_modules():
            w = getattr(m, "w_bf16", None)          # the compute copy a convolution is launched with (fp32 in this mode)
            if torch.is_tensor(w):
                conv_names[w.data_ptr()] = name
            if type(m).__name__ == "HipGroupNorm":
                gn_names[m.weight.data_ptr()] = name
        masks, count, unmatched, bare = {}, {}, [], 0
        for kind, key, y in self.rec:
            if kind == "conv":
                name = conv_names.get(key)
            elif kind == "gn":
                name = gn_names.get(key)
            else:
                name = self.bare[bare] if bare < len(self.bare) else None
                bare += 1
            if name is None:
                unmatched.append((kind, key, tuple(y.shape)))
                continue
            i = count.get(name, 0)
            count[name] = i + 1
            masks[(name, i)] = y.permute(0, 3, 1, 2).contiguous() if y.dim() == 4 else y      # NHWC -> the oracle's NCHW
        return masks, unmatched
