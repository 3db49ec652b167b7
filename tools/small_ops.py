"""Which Python lines issue the small device operations (copies, fills, elementwise ATen kernels) of one training step.
A TorchDispatchMode logs every ATen call that reaches the device together with the innermost frame of this repo.
Usage (GPU box): python tools/small_ops.py [arch]"""
import collections
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, ".")
import bench  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.as_strided", "aten.slice", "aten.select", "aten.detach", "aten.alias", "aten.t.", "aten.permute",
        "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten.empty", "aten.transpose", "aten.unbind", "aten.split", "aten.narrow",
        "aten._local_scalar_dense", "aten.is_", "aten.lift_fresh", "aten.new_empty", "aten.set_", "aten.resize_", "aten.record_stream")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.per = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            on_dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            dev_kw = (kwargs or {}).get("device")
            if on_dev or (dev_kw is not None and "cuda" in str(dev_kw)):
                where = "?"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if ("slenderobjdet_amd" in fr.filename or fr.filename.endswith("bench.py")) and "small_ops" not in fr.filename:
                        where = f"{fr.filename.split('slenderobjdet_amd/')[-1]}:{fr.lineno} {fr.name}"
                        break
                self.per[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "fcos"
    from slenderobjdet_amd.data import SyntheticCocoBatches
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    cfg = bench.make_cfg(50, arch)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    if arch in ("retinanet", "rrcnn"):
        bench.damp_residual_branches(model)
    opt = build_optimizer(cfg, model)
    loader = SyntheticCocoBatches(16, 800, 1333, rank=0, device=torch.device("cuda"), pool=2, rotated=arch == "rrcnn")
    it = iter(loader)
    for _ in range(3):
        bench.train_step(model, opt, next(it))
    torch.cuda.synchronize()
    log = Log()
    with log:
        for _ in range(2):
            bench.train_step(model, opt, next(it))
        torch.cuda.synchronize()
    for (name, where), n in log.per.most_common(80):
        print(f"{n / 2:7.1f} /step  {name:34s} {where}")
    print("total device ATen calls per step:", sum(log.per.values()) / 2)


if __name__ == "__main__":
    main()
