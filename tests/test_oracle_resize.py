"""CPU: the resize oracle's pinned properties and its agreement with the product's host-side filter tables."""
import numpy as np

from oracle import resize as orz


def test_resize_oracle_properties():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (11, 14, 3)).astype(np.uint8)
    assert np.array_equal(orz.pil_resize_bilinear(img, 11, 14), img)                      # identity
    const = np.full((9, 13, 3), 77, np.uint8)
    assert (orz.pil_resize_bilinear(const, 23, 5) == 77).all()                             # partition of unity survives quantisation
    # 2x down-scaling: support 2, triangle weights (1, 3, 3, 1) / 8 -> a column pattern of period 2 (0, 200) becomes 100 inside
    pat = np.zeros((4, 16, 3), np.uint8)
    pat[:, 1::2] = 200
    half = orz.pil_resize_bilinear(pat, 4, 8)
    assert (half[:, 1:-1] == 100).all(), half[0, :, 0]
    assert orz.resize_shortest_edge(480, 640, 800, 1333) == (800, 1067) and orz.resize_shortest_edge(300, 1200, 800, 1333) == (333, 1333)


def test_host_tables_match_the_oracle_filter():
    from slenderobjdet_amd.data.transforms import pil_bilinear_coeffs, resize_shortest_edge_size, transform_boxes

    import torch

    for in_size, out_size in ((480, 800), (1333, 800), (37, 37), (500, 333), (7, 20)):
        b, k = pil_bilinear_coeffs(in_size, out_size)
        ref = orz._coeffs_1d(in_size, out_size)
        for x, (x0, ks) in enumerate(ref):
            assert b[x, 0] == x0 and b[x, 1] == len(ks) and list(k[x, : len(ks)]) == ks and (k[x, len(ks):] == 0).all()
    for h, w, s in ((480, 640, 800), (300, 1200, 800), (1000, 700, 640)):
        assert resize_shortest_edge_size(h, w, s, 1333) == orz.resize_shortest_edge(h, w, s, 1333)
    boxes = np.array([[10.0, 20.0, 200.0, 100.0], [0.0, 0.0, 640.0, 480.0]], np.float32)
    for flip in (False, True):
        got = transform_boxes(torch.tensor(boxes), 480, 640, 800, 1067, flip).numpy()
        assert np.allclose(got, orz.transform_boxes(boxes, 480, 640, 800, 1067, flip), atol=1e-4)
