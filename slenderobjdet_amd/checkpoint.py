"""Checkpoint interchange with the reference's ``DetectionCheckpointer`` (slender_det/checkpoint/detection_checkpoint.py, a thin
subclass of detectron2's; train_net.py:30,149-152).

The native ``state_dict`` differs from a reference / detectron2 one in layout, not in content:

* conv weights are stored ``(K, R, S, C)`` (the layout the HIP kernels read) instead of ``(K, C, R, S)``;
* ``FrozenBatchNorm2d`` lives as ``bn_weight / bn_bias / bn_running_mean / bn_running_var`` buffers on the conv (folded into the
  bf16 compute copy) instead of a ``.norm`` sub-module;
* the FCOS head fuses ``cls_logits`` (+ ``centerness``) into ``cls_pred`` and ``bbox_pred`` (+ ``centerness``) into ``box_pred``,
  both zero-padded to a multiple of 8 output channels, its towers are ``[conv, gn]`` units instead of a flat ``nn.Sequential`` and
  the five ``Scale`` modules are one 5-vector (fcos.py:476-582);
* fully connected layers run as 1x1 convolutions over ``(R, 1, 1, C*H*W)`` rows flattened HWC, so the first FC after ROI pooling has
  its input dimension permuted from CHW.

``reference_to_native`` / ``native_to_reference`` convert both ways; ``load_file`` reads ``.pth`` and detectron2 ``.pkl`` files
(including the Caffe2-named ImageNet backbones, detectron2/checkpoint/c2_model_loading.py [upstream knowledge, SURVEY.md C.9]).
Nothing is matched silently: every key that could not be placed is returned in the report and logged by the checkpointer.
"""
import logging
import pickle
import re

import numpy as np
import torch

NATIVE_FORMAT = "slenderobjdet_amd-native-1"
_BN = {"weight": "bn_weight", "bias": "bn_bias", "running_mean": "bn_running_mean", "running_var": "bn_running_var"}
logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------ file formats
def _c2_to_d2_names(keys):
    """Caffe2 blob names of the ImageNet-pretrained ResNets (MSRA R-50/R-101.pkl) -> detectron2 module names (backbone.bottom_up.*)."""
    out = {}
    for k in keys:
        n = k
        n = re.sub(r"^conv1_w$", "stem.conv1.weight", n)
        n = re.sub(r"^res_conv1_bn_", "stem.conv1.norm.", n)
        n = re.sub(r"^res(\d)_(\d+)_branch1_w$", r"res\1.\2.shortcut.weight", n)
        n = re.sub(r"^res(\d)_(\d+)_branch1_bn_", r"res\1.\2.shortcut.norm.", n)
        for c2, d2 in (("a", "1"), ("b", "2"), ("c", "3")):
            n = re.sub(rf"^res(\d)_(\d+)_branch2{c2}_w$", rf"res\1.\2.conv{d2}.weight", n)
            n = re.sub(rf"^res(\d)_(\d+)_branch2{c2}_bn_", rf"res\1.\2.conv{d2}.norm.", n)
        n = re.sub(r"\.norm\.s$", ".norm.weight", n)
        n = re.sub(r"\.norm\.b$", ".norm.bias", n)
        n = re.sub(r"\.norm\.rm$", ".norm.running_mean", n)
        n = re.sub(r"\.norm\.riv$", ".norm.running_var", n)
        if n != k and not n.startswith("fc1000"):
            out[k] = "backbone.bottom_up." + n
    return out


def load_file(path):
    """-> (state dict of torch tensors, meta dict).  ``meta['native']`` tells whether the file was written by this package."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        sd = data["model"] if isinstance(data, dict) and "model" in data else (data.get("blobs", data) if isinstance(data, dict) else data)
        sd = {k: v for k, v in sd.items() if not k.endswith("_momentum")}
        c2 = _c2_to_d2_names(sd.keys())
        if c2:     # Caffe2 names: keep only what maps onto the backbone
            sd = {c2[k]: v for k, v in sd.items() if k in c2}
        sd = {k: torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v for k, v in sd.items()}
        return sd, {"native": False, "raw": data if isinstance(data, dict) else {}}
    data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data else data
    return sd, {"native": isinstance(data, dict) and data.get("__format__") == NATIVE_FORMAT, "raw": data if isinstance(data, dict) else {}}


# ------------------------------------------------------------------------------------------------ conversion
def _conv_modules(model):
    from .layers.deform_conv import DeformConv
    from .layers.nn import HipConv2d

    return {name: m for name, m in model.named_modules() if isinstance(m, (HipConv2d, DeformConv))}


def _fit(t, shape):
    """Zero-pads a tensor up to ``shape`` (channel padding of the fused / padded prediction convs); None if it does not fit."""
    if tuple(t.shape) == tuple(shape):
        return t
    if t.dim() != len(shape) or any(a > b for a, b in zip(t.shape, shape)):
        return None
    out = torch.zeros(shape, dtype=t.dtype)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


def _krsc(w):      # (K,C,R,S) -> (K,R,S,C)
    return w.permute(0, 2, 3, 1).contiguous()


def _kcrs(w):
    return w.permute(0, 3, 1, 2).contiguous()


def _fcos_head_rules(model):
    """native key -> function(ref_sd) for the FCOS head (reference names: fcos.py:486-547)."""
    head = getattr(model, "head", None)
    if head is None or not hasattr(head, "cls_pred") or not hasattr(head, "box_pred") or not hasattr(head, "scales"):
        return {}
    rules = {}
    ctr_on_reg = head.centerness_on_reg

    def cat_rows(names, pad_to, conv_w):
        def fn(sd):
            parts = [sd[n] for n in names]
            t = torch.cat(parts, 0)
            if conv_w:
                t = _krsc(t)
            shape = (pad_to,) + tuple(t.shape[1:])
            return _fit(t, shape)
        fn.uses = names
        return fn

    cls_src = ["head.cls_logits"] + ([] if ctr_on_reg else ["head.centerness"])
    box_src = ["head.bbox_pred"] + (["head.centerness"] if ctr_on_reg else [])
    rules["head.cls_pred.weight"] = cat_rows([n + ".weight" for n in cls_src], head.kc_pad, True)
    rules["head.cls_pred.bias"] = cat_rows([n + ".bias" for n in cls_src], head.kc_pad, False)
    rules["head.box_pred.weight"] = cat_rows([n + ".weight" for n in box_src], 8, True)
    rules["head.box_pred.bias"] = cat_rows([n + ".bias" for n in box_src], 8, False)

    nlev = head.scales.numel()

    def scales(sd):
        return torch.stack([sd[f"head.scales.{i}.scale"].reshape(()) for i in range(nlev)])
    scales.uses = [f"head.scales.{i}.scale" for i in range(nlev)]
    rules["head.scales"] = scales
    for tower in ("cls_tower", "bbox_tower"):
        for i in range(len(getattr(head, tower))):
            for part, off, conv in (("conv", 0, True), ("gn", 1, False)):
                for leaf in ("weight", "bias"):
                    src = f"head.{tower}.{3 * i + off}.{leaf}"

                    def fn(sd, src=src, conv=conv, leaf=leaf):
                        t = sd[src]
                        return _krsc(t) if (conv and leaf == "weight") else t
                    fn.uses = [src]
                    rules[f"head.{tower}.{i}.{part}.{leaf}"] = fn
    return rules


# module attributes named differently here and in detectron2 (native prefix, reference prefix)
_PREFIX_ALIASES = (("proposal_generator.head.", "proposal_generator.rpn_head."),)


def _ref_name(name):
    for nat, ref in _PREFIX_ALIASES:
        if name.startswith(nat):
            return ref + name[len(nat):]
    # detectron2's FastRCNNConvFCHead registers its layers as fc1, fc2, ... (add_module("fc{k+1}")); natively ``fcs`` is a ModuleList
    m = re.match(r"^(.*\.box_head)\.fcs\.(\d+)\.(.*)$", name)
    if m:
        return f"{m.group(1)}.fc{int(m.group(2)) + 1}.{m.group(3)}"
    return name


def _unit_tower_map(model):
    """native key prefix -> reference key prefix for towers built from [conv (+ GroupNorm) + ReLU] UNITS.  The reference keeps such a
    tower as a flat ``nn.Sequential`` (RetinaNetHead: ``cls_subnet.{2i}`` = conv, ``{2i+1}`` = ReLU, retina_rotated.py:418-430; the
    ablation heads with GN: ``{3i}`` conv, ``{3i+1}`` GroupNorm) or as a Sequential of per-layer Sequentials (RepPointsDetector:
    ``cls_conv.{i}.0`` conv, ``cls_conv.{i}.1`` GroupNorm, rpd.py:191-204); natively unit i is ``<tower>.{i}.conv`` / ``<tower>.{i}.gn``."""
    from torch import nn

    from .layers.nn import ConvGnRelu, ConvML, ConvReluML

    out = {}
    for name, mod in model.named_modules():
        if isinstance(mod, ConvML):          # a plain shared conv: the reference has the nn.Conv2d itself under this name
            out[f"{name}.conv"] = name
            continue
        if not isinstance(mod, nn.ModuleList) or len(mod) == 0 or not all(isinstance(u, (ConvGnRelu, ConvReluML)) for u in mod):
            continue
        nested = getattr(mod, "ckpt_nested", False) or name.rsplit(".", 1)[-1] in ("cls_conv", "reg_conv")
        for i, u in enumerate(mod):
            stride = 3 if isinstance(u, ConvGnRelu) else 2
            out[f"{name}.{i}.conv"] = f"{name}.{i}.0" if nested else f"{name}.{stride * i}"
            if isinstance(u, ConvGnRelu):
                out[f"{name}.{i}.gn"] = f"{name}.{i}.1" if nested else f"{name}.{stride * i + 1}"
    return out


def reference_to_native(ref_sd, model):
    """Maps a reference / detectron2 state dict onto ``model``'s native keys.
    Returns (native state dict, report) with report = {"missing": native keys nothing was found for, "unexpected": reference keys
    that were not used, "shape_mismatch": [(key, ref shape, native shape)]}."""
    native = model.state_dict()
    convs = _conv_modules(model)
    rules = _fcos_head_rules(model)
    towers = {} if rules else _unit_tower_map(model)
    out, used, mismatch = {}, set(), []
    for key, cur in native.items():
        if key in rules:
            fn = rules[key]
            if all(u in ref_sd for u in fn.uses):
                t = fn(ref_sd)
                if t is not None and tuple(t.shape) == tuple(cur.shape):
                    out[key] = t.to(cur.dtype)
                    used.update(fn.uses)
                else:
                    mismatch.append((key, [tuple(ref_sd[u].shape) for u in fn.uses], tuple(cur.shape)))
            continue
        mod, _, leaf = key.rpartition(".")
        src = key
        if mod in convs and leaf in _BN.values():
            src = mod + ".norm." + {v: k for k, v in _BN.items()}[leaf]
        elif mod in towers:
            src = towers[mod] + "." + leaf
        src = _ref_name(src)
        if src not in ref_sd:
            continue
        t = ref_sd[src]
        if mod in convs and leaf == "weight":
            K, R, S, C = cur.shape
            if t.dim() == 4:
                t = _krsc(t)
            elif t.dim() == 2 and R == 1 and S == 1:          # nn.Linear stored as a 1x1 conv; first FC after ROI pooling: CHW -> HWC
                hw = getattr(convs[mod], "fc_input_chw", None)
                if hw is not None:
                    c, h, w = hw
                    t = t.reshape(t.shape[0], c, h, w).permute(0, 2, 3, 1).reshape(t.shape[0], -1)
                t = t.reshape(t.shape[0], 1, 1, t.shape[1])
        t = _fit(t, tuple(cur.shape)) if torch.is_tensor(t) else None
        if t is None:
            mismatch.append((key, tuple(ref_sd[src].shape), tuple(cur.shape)))
            continue
        out[key] = t.to(cur.dtype)
        used.add(src)
    report = {"missing": [k for k in native if k not in out], "unexpected": [k for k in ref_sd if k not in used], "shape_mismatch": mismatch}
    return out, report


def native_to_reference(model):
    """The reference's key names / layouts for ``model``'s current weights (so that a reference DetectionCheckpointer can load what
    this package trained).  Padded rows are dropped, fused prediction convs split, FrozenBN buffers move back under ``.norm``."""
    sd = model.state_dict()
    convs = _conv_modules(model)
    head = getattr(model, "head", None)
    fcos = head is not None and hasattr(head, "cls_pred") and hasattr(head, "box_pred") and hasattr(head, "scales")
    towers = {} if fcos else _unit_tower_map(model)
    out = {}
    for key, t in sd.items():
        t = t.detach().cpu()
        mod, _, leaf = key.rpartition(".")
        rows = getattr(convs.get(mod), "ckpt_rows", None)
        if rows is not None and leaf in ("weight", "bias"):      # prediction convs padded to a multiple of 8 output channels: drop the pad
            t = t[:rows]
        if mod in towers:
            out[_ref_name(towers[mod] + "." + leaf)] = _kcrs(t) if (leaf == "weight" and t.dim() == 4) else t.clone()
            continue
        if fcos and key.startswith("head."):
            if key == "head.scales":
                for i in range(t.numel()):
                    out[f"head.scales.{i}.scale"] = t[i].reshape(1).clone()
                continue
            m = re.match(r"head\.(cls_tower|bbox_tower)\.(\d+)\.(conv|gn)\.(weight|bias)$", key)
            if m:
                idx = 3 * int(m.group(2)) + (0 if m.group(3) == "conv" else 1)
                out[f"head.{m.group(1)}.{idx}.{m.group(4)}"] = _kcrs(t) if (m.group(3) == "conv" and m.group(4) == "weight") else t.clone()
                continue
            if mod in ("head.cls_pred", "head.box_pred"):
                w = _kcrs(t) if leaf == "weight" else t
                K = model.num_classes
                if mod == "head.cls_pred":
                    out[f"head.cls_logits.{leaf}"] = w[:K].clone()
                    if not head.centerness_on_reg:
                        out[f"head.centerness.{leaf}"] = w[K:K + 1].clone()
                else:
                    out[f"head.bbox_pred.{leaf}"] = w[:4].clone()
                    if head.centerness_on_reg:
                        out[f"head.centerness.{leaf}"] = w[4:5].clone()
                continue
        if mod in convs:
            if leaf in _BN.values():
                out[mod + ".norm." + {v: k for k, v in _BN.items()}[leaf]] = t.clone()
                continue
            if leaf == "weight":
                conv = convs[mod]
                hw = getattr(conv, "fc_input_chw", None)
                if getattr(conv, "is_linear", False) or hw is not None:
                    w2 = t.reshape(t.shape[0], -1)
                    if hw is not None:
                        c, h, w = hw
                        w2 = w2.reshape(t.shape[0], h, w, c).permute(0, 3, 1, 2).reshape(t.shape[0], -1)
                    out[key] = w2.clone()
                else:
                    out[key] = _kcrs(t)
                continue
        out[key] = t.clone()
    return {_ref_name(k): v for k, v in out.items()}


def load_into(model, path, strict=False, allow_missing=()):
    """Loads ``path`` (native or reference format) into ``model``.  Returns the report dict; raises if NOTHING of the file matched
    (a silently random-initialised model is the failure this guards against), if the file is a whole-model checkpoint (it holds tensors
    outside ``backbone.``) and leaves a trainable tensor outside the backbone without a value - unless its name starts with one of the
    ``allow_missing`` prefixes (``"*"`` accepts every such tensor with a warning, detectron2's DetectionCheckpointer behaviour:
    fine-tuning from a detector trained with another head or NUM_CLASSES; the trainer forwards MODEL.WEIGHTS_ALLOW_MISSING) -, or,
    with ``strict``, on any incompatibility.  Every check runs BEFORE the model is touched: a refused file leaves it as it was."""
    sd, meta = load_file(path)
    if meta["native"]:
        own = model.state_dict()
        mism = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in sd.items() if k in own and hasattr(v, "shape") and tuple(v.shape) != tuple(own[k].shape)]
        native = {k: v for k, v in sd.items() if k in own and not any(k == m[0] for m in mism)}
        report = {"missing": [k for k in own if k not in native], "unexpected": [k for k in sd if k not in own], "shape_mismatch": mism}
    else:
        native, report = reference_to_native(sd, model)
    matched = len(native)
    if matched == 0:
        raise RuntimeError(f"checkpoint {path}: no tensor matches this model ({len(sd)} tensors in the file)")
    if any(not k.startswith("backbone.") for k in sd):
        # a detector checkpoint, not an ImageNet backbone file: a head parameter nothing was found for would train from its random
        # initialisation behind a log line (the padded / re-laid-out heads are exactly where a name or shape rule can be missing)
        params = {n for n, _ in model.named_parameters()}
        accept_all = "*" in tuple(allow_missing)
        lost = [k for k in list(report["missing"]) + [m[0] for m in report["shape_mismatch"]]
                if k in params and not k.startswith("backbone.") and not any(k.startswith(a) for a in allow_missing)]
        if lost and accept_all:
            logger.warning("checkpoint %s: %d head parameter(s) keep their initialisation (allow_missing '*'): %s%s", path, len(lost),
                           ", ".join(lost[:12]), " ..." if len(lost) > 12 else "")
        elif lost:
            raise RuntimeError(f"checkpoint {path}: {len(lost)} head parameter(s) of the model have no counterpart in the file "
                               f"(pass allow_missing=(prefix, ...) or MODEL.WEIGHTS_ALLOW_MISSING to accept): {lost[:12]}{' ...' if len(lost) > 12 else ''}")
    if strict and any(report[k] for k in report):
        raise RuntimeError(f"checkpoint {path} is incompatible with the model: {report}")
    model.load_state_dict(native, strict=False)
    for kind in ("shape_mismatch", "missing", "unexpected"):
        if report[kind]:
            logger.warning("checkpoint %s: %d %s key(s): %s%s", path, len(report[kind]), kind.replace("_", " "),
                           ", ".join(str(k) for k in report[kind][:12]), " ..." if len(report[kind]) > 12 else "")
    if getattr(model, "arena", None) is not None:
        model.arena.bump()
    return report, meta
