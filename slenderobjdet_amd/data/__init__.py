from .synthetic import SyntheticCocoBatches, synthetic_batch
from .transforms import DeviceInputPipeline, pil_bilinear_coeffs, resize_shortest_edge_size, transform_boxes
