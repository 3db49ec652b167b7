"""Fused stem (uint8 -> normalise -> conv7x7s2 -> BN -> ReLU -> max-pool, one kernel) vs the three separate launches: python tools/bench_stem.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_cfg  # noqa: E402
from slenderobjdet_amd.layers import functional as HF  # noqa: E402
from slenderobjdet_amd.modeling import build_model  # noqa: E402

cfg = make_cfg(50)
model = build_model(cfg)
stem = model.backbone.bottom_up.stem
imgs = [torch.randint(0, 256, (3, 800, 1333), dtype=torch.uint8, device="cuda") for _ in range(16)]
sizes = [(800, 1333)] * 16


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def fused():
    return stem(HF.RawImageBatch(imgs, sizes, (800, 1344), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD))


def unfused():
    return stem(HF.RawImageBatch(imgs, sizes, (800, 1344), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD).materialize())


with torch.no_grad():
    for _ in range(3):
        print("fused %.3f ms   unfused (preprocess + conv + pool) %.3f ms" % (timeit(fused), timeit(unfused)), flush=True)
