"""CPU oracle: a plain-PyTorch (fp32, CPU) restatement of the reference's algorithm for the FCOS training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product path (``slenderobjdet_amd``) never does and fails loudly without its HIP
library.

Pinning status (see DESIGN.md §"Oracle"):
  * functions that restate code present under /root/reference (``slender_det/layers/iou_loss.py``,
    ``slender_det/modeling/meta_arch/fcos/utils.py``, ``fcos/fcosv2.py`` FCOSHead / losses, ``tests/
    test_deformable_conv.py`` helpers) are PINNED: ``tests/golden/make_golden.py`` imported the reference files
    in the build container (under stub modules for detectron2/fvcore, which are absent everywhere) and wrote
    the input/output vectors committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks them.
  * functions that restate third-party arithmetic whose source is NOT in the reference tree
    (fvcore ``sigmoid_focal_loss``/``giou_loss``/``smooth_l1_loss``, detectron2 @ 8bc84a2ff8a0b5787ec ResNet / FPN /
    FrozenBatchNorm2d / Matcher / Box2BoxTransform / DeformConv backward / ROIAlign / NMS, torchvision nms)
    are "parity unpinned" against upstream: they follow the published formulas and are
    cross-checked against independent torch built-ins (F.binary_cross_entropy_with_logits, F.conv2d,
    F.group_norm, torch.autograd.gradcheck) in ``tests/test_oracle_crosscheck.py``.
"""
