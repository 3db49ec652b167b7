from .roi_heads import (ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY, FastRCNNConvFCHead, FastRCNNOutputLayers, ProposalVisibleHead, ROIPooler, RROIHeads,
                        StandardROIHeads, build_roi_heads, fast_rcnn_inference_single_image)
