"""BorderAlign / CornerPool (the reference's own native operators): time per call of the library's kernels, and of a
side build of an older version of csrc/slender_ops.hip when tools/micro/_old_slender_ops.so exists (same C ABI), at CornerNet / BorderDet
sizes.  python tools/bench_f3.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slenderobjdet_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong


def libs():
    out = [("library", _C.load())]
    old = os.path.join(ROOT, "tools", "micro", "_old_slender_ops.so")
    if os.path.exists(old):
        out.append(("previous", ctypes.CDLL(old)))
    for _, lib in out:
        lib.sod_border_align_fwd.argtypes = [P, P, P, I, I, I, I, I, I, P]
        lib.sod_border_align_bwd.argtypes = [P, P, P, P, I, I, I, I, I, I, P]
        lib.sod_corner_pool_fwd.argtypes = [P, P, L, I, I, I, P]
        lib.sod_corner_pool_bwd.argtypes = [P, P, P, L, I, I, I, I, P]
    return out


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    torch.manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    # CornerNet: 16 x 128 channels x 128 x 128 fp32 per pooling branch
    N, C, H, W = 16, 128, 128, 128
    x = torch.randn(N, C, H, W, device=dev)
    dy = torch.randn_like(x)
    mb = x.numel() * 4 / 1e6
    results = {}
    for name, lib in libs():
        for mode, mname in enumerate(("bottom", "top", "right", "left")):
            y, dx = torch.empty_like(x), torch.zeros_like(x)
            f = timeit(lambda: lib.sod_corner_pool_fwd(p(x), p(y), N * C, H, W, mode, st))
            b = timeit(lambda: lib.sod_corner_pool_bwd(p(x), p(dy), p(dx), N * C, H, W, mode, 0, st))
            results[(name, mname)] = (y, f, b)
            print(f"corner_pool {mname:6s} {N}x{C}x{H}x{W} ({mb:.0f} MB)  {name:8s}: fwd {f:8.1f} us ({2 * mb / f:.2f} TB/s)  bwd {b:8.1f} us", flush=True)
    for mname in ("bottom", "top", "right", "left"):
        if ("previous", mname) in results:
            assert torch.equal(results[("library", mname)][0], results[("previous", mname)][0]), mname
    # BorderDet: one box per location of a 100 x 136 level, 4 x 128 border channels, pool 10
    B, C, H, W, pool = 4, 128, 100, 136, 10
    K = H * W
    feat = torch.randn(B, 4 * C, H, W, device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
    ctr = torch.stack((xs, ys), -1).reshape(1, K, 2)
    wh = torch.rand(B, K, 2, device=dev) * 30 + 2
    boxes = torch.cat(((ctr - wh / 2).clamp(min=0), torch.minimum(ctr + wh / 2, torch.tensor([W - 1.0, H - 1.0], device=dev))), -1).contiguous()
    dout = torch.randn(B, C, K, 4, device=dev)
    outs = {}
    for name, lib in libs():
        out, df = torch.empty(B, C, K, 4, device=dev), torch.zeros_like(feat)
        f = timeit(lambda: lib.sod_border_align_fwd(p(feat), p(boxes), p(out), B, C, K, H, W, pool, st))
        b = timeit(lambda: lib.sod_border_align_bwd(p(dout), p(feat), p(boxes), p(df), B, C, K, H, W, pool, st))
        outs[name] = out
        print(f"border_align {B}x{4 * C}x{H}x{W}, {K} boxes, pool {pool}  {name:8s}: fwd {f:8.1f} us  bwd {b:8.1f} us", flush=True)
    if "previous" in outs:
        assert torch.equal(outs["library"], outs["previous"])
        print("outputs equal bit for bit")


if __name__ == "__main__":
    main()
