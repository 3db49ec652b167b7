"""Do the GroupNorm backward passes run BESIDE a 256 x 256 convolution kernel (same CUs), or only between its workgroups?  Times, on the
head-tower tensors (five levels, batch 16, 256 channels): the tower data gradient alone, the GroupNorm backward alone, both in one stream,
and both in two streams.  python tools/bench_corun.py   (SOD_HIP_LIB=<other.so> for another build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slenderobjdet_amd.layers import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, C, G = 16, 256, 32
levels = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
xs = [torch.randn(N, h, w, C, device=dev).bfloat16() for h, w in levels]
dys = [torch.randn(N, h, w, C, device=dev).bfloat16() * 0.01 for h, w in levels]
dy2 = [torch.randn(N, h, w, C, device=dev).bfloat16() * 0.01 for h, w in levels]
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
dgamma, dbeta, dxsum = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
w = torch.randn(C, 3, 3, C, device=dev) * 0.02
_, wt = HF.weight_prep(w)
_, stats = HF.groupnorm_fwd_ml(xs, gamma, beta, G, relu=True)
s2 = torch.cuda.Stream(dev)


def conv():
    HF.conv2d_dgrad_ml(dy2, wt, levels, 1, 1, 1)


def gn():
    HF.groupnorm_bwd_ml(dys, xs, gamma, beta, stats, G, dgamma, dbeta, relu=True, dxsum=dxsum)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3


def both_serial():
    conv(); gn()


def both_two_streams():
    main = torch.cuda.current_stream(dev)
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        gn()
    conv()
    main.wait_stream(s2)


def chains():      # four data gradients on the main stream, four GroupNorm backward passes beside them (the convolution enqueued first)
    main = torch.cuda.current_stream(dev)
    s2.wait_stream(main)
    for _ in range(4):
        conv()
    with torch.cuda.stream(s2):
        for _ in range(4):
            gn()
    main.wait_stream(s2)


tch = timed(chains, 3)
tc, tg, ts, t2 = timed(conv), timed(gn), timed(both_serial), timed(both_two_streams)
print(f"4 x dgrad beside 4 x GroupNorm backward: {tch:.1f} us (4 x dgrad alone {4 * tc:.1f}, 4 x GN alone {4 * tg:.1f})", flush=True)
print(f"tower dgrad {tc:.1f} us | GroupNorm backward {tg:.1f} us | one stream {ts:.1f} us | two streams {t2:.1f} us (hidden: {tc + tg - t2:.1f} us of {tg:.1f})", flush=True)

# where in the window do the GroupNorm passes complete?  (events after every launch of the chains experiment)
main = torch.cuda.current_stream(dev)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t0.record()
s2.wait_stream(main)
ce, ge = [], []
for _ in range(4):
    conv()
    e = torch.cuda.Event(enable_timing=True); e.record(); ce.append(e)
with torch.cuda.stream(s2):
    for _ in range(4):
        gn()
        e = torch.cuda.Event(enable_timing=True); e.record(s2); ge.append(e)
main.wait_stream(s2)
torch.cuda.synchronize()
print("dgrad launches end at (us):", " ".join(f"{t0.elapsed_time(e) * 1e3:.0f}" for e in ce), "| GroupNorm backward launches end at:",
      " ".join(f"{t0.elapsed_time(e) * 1e3:.0f}" for e in ge), flush=True)
