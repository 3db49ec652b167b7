"""detectron2.modeling.box_regression (source absent): Box2BoxTransform on XYXY boxes and Box2BoxTransformRotated on
(cx, cy, w, h, angle) boxes, both as one HIP kernel per call (``sod_box2box_get_deltas`` / ``sod_box2box_apply_deltas``)."""
import math

from ..layers import functional as HF

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    box_dim = 4

    def __init__(self, weights, scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = tuple(float(w) for w in weights)
        assert len(self.weights) == self.box_dim, f"{type(self).__name__} needs {self.box_dim} weights, got {self.weights}"
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes, target_boxes):
        return HF.box2box_get_deltas(src_boxes.float().contiguous(), target_boxes.float().contiguous(), self.weights)

    def apply_deltas(self, deltas, boxes):
        """deltas (N, k*box_dim) class-specific, boxes (N, box_dim) -> (N, k*box_dim)."""
        k = deltas.shape[1] // self.box_dim
        return HF.box2box_apply_deltas(deltas.float().contiguous(), boxes.float().contiguous(), self.weights, self.scale_clamp, k)


class Box2BoxTransformRotated(Box2BoxTransform):
    box_dim = 5
