from collections import namedtuple


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    """detectron2.layers.ShapeSpec (used at fcosv2.py:58-60, backbone/fpn.py:95)."""

    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)
