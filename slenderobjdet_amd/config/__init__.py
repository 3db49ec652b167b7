"""``slender_det.config`` surface (reference: slender_det/config.py:213-220)."""
from .cfgnode import CfgNode
from .defaults import DEFAULTS

_C = CfgNode(DEFAULTS)


def get_cfg() -> CfgNode:
    """Returns the shared global node, exactly like the reference (config.py:213-220 returns ``_C``, not a clone).
    Use ``get_cfg().clone()`` for an independent copy."""
    return _C


def fresh_cfg() -> CfgNode:
    """An independent default tree (not part of the reference surface; handy for tests and bench)."""
    return CfgNode(DEFAULTS)


__all__ = ["CfgNode", "get_cfg", "fresh_cfg"]
