"""A self-contained yacs-compatible ``CfgNode`` (yacs/detectron2 are not available in this image).

Mirrors the surface the reference uses (slender_det/config.py:1-220, train_net.py:145-154): attribute access,
``merge_from_file`` with ``_BASE_`` inheritance, ``merge_from_list``, ``freeze``/``defrost``, ``clone``, ``dump``.
YAMLs may contain ``!!python/object/apply:eval`` (configs/retina/Base-RetinaNet.yaml:8), which is evaluated like
detectron2's unsafe loader does.
"""
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class _EvalLoader(yaml.SafeLoader):
    pass


def _apply_eval(loader, node):
    args = loader.construct_sequence(node, deep=True)
    return eval(args[0])  # noqa: S307 - same behaviour as the reference's config loading


def _py_tuple(loader, node):
    return tuple(loader.construct_sequence(node, deep=True))


_EvalLoader.add_constructor("tag:yaml.org,2002:python/object/apply:eval", _apply_eval)
_EvalLoader.add_constructor("tag:yaml.org,2002:python/tuple", _py_tuple)


def _literal(v):
    """Decode a CLI override the way yacs does (python literal, else the raw string)."""
    if not isinstance(v, str):
        return v
    import ast

    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access ------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def __setitem__(self, key, value):
        if self.__dict__.get(CfgNode.IMMUTABLE, False):
            raise AttributeError(f"Attempted to set {key}, but CfgNode is immutable")
        super().__setitem__(key, value)

    # freezing ----------------------------------------------------------------------------
    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._immutable(flag)

    def freeze(self):
        self._immutable(True)

    def defrost(self):
        self._immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    # merging -----------------------------------------------------------------------------
    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.load(f, Loader=_EvalLoader) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, cfg_filename, allow_unsafe=True):
        loaded = CfgNode.load_yaml_with_base(cfg_filename)
        self.merge_from_other_cfg(loaded)

    def merge_from_other_cfg(self, other):
        _merge_into(other, self, [])

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2:
            raise AssertionError(f"Override list has odd length: {cfg_list}; it must be a list of pairs")
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent key: {full_key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent key: {full_key}")
            node[parts[-1]] = _coerce(_literal(v), node[parts[-1]], full_key)

    def dump(self, **kwargs):
        def to_dict(n):
            return {k: to_dict(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}

        return yaml.safe_dump(to_dict(self), **kwargs)

    def __repr__(self):
        return self.dump()

    __str__ = __repr__


def _merge_dict(src, dst):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k])
        else:
            dst[k] = v


def _coerce(new, old, key):
    """yacs-style type reconciliation (tuple<->list, int->float)."""
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, CfgNode) and isinstance(new, dict):
        return CfgNode(new)
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")


def _merge_into(src, dst, path):
    for k, v in src.items():
        full = ".".join(path + [k])
        if k not in dst:
            raise KeyError(f"Non-existent config key: {full}")
        if isinstance(dst[k], CfgNode) and isinstance(v, dict):
            _merge_into(v, dst[k], path + [k])
        else:
            v = copy.deepcopy(v)
            if isinstance(v, str) and not isinstance(dst[k], str):
                v = _literal(v)   # yacs decodes '("coco_2017_train",)' style strings
            dst[k] = _coerce(v, dst[k], full)
