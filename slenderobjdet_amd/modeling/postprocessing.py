"""detectron2.modeling.postprocessing.detector_postprocess (call site fcosv2.py:262): rescale boxes to the requested
output resolution, clip, drop empty ones."""
from ..structures import Instances


def detector_postprocess(results, output_height, output_width):
    scale_x = output_width / results.image_size[1]
    scale_y = output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    boxes = results.pred_boxes if results.has("pred_boxes") else results.proposal_boxes
    boxes.scale(scale_x, scale_y)
    boxes.clip(results.image_size)
    return results[boxes.nonempty()]
