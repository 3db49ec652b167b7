"""How far do two correct fp32 implementations of the SAME 100-iteration FCOS R18 run drift apart?  CPU only: the fp32 oracle
(oracle/model.py = the reference's CPU path restated) against the same oracle in float64 and against itself with another summation
order (one thread instead of all, which changes oneDNN's blocking / reduction order).  Same initial weights, data and schedule as
tests/test_gpu_parity100.py.  Writes tests/golden/chaos100.json (about 15 minutes on 8 cores); ``make_chaos100.py emu`` adds the trajectory of
the bf16-storage-emulating oracle, so that the GPU test does not spend a minute of the GPU box on two 100-iteration CPU runs (it re-runs
the first iterations of both oracles live and checks them against the file).  The file records the sha256 of the sources the
trajectories are a function of; the test refuses a fixture whose sources have changed: ``make_chaos100.py stamp`` re-runs the first
iterations of both oracles against the file and re-stamps it (the change did not touch the arithmetic), otherwise regenerate."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


SOURCES = ("oracle/model.py", "oracle/nn.py", "oracle/losses.py", "oracle/fcos_targets.py", "slenderobjdet_amd/data/synthetic.py")


def source_hashes():
    """sha256 of the files the stored trajectories are a function of (tests/test_gpu_parity100.py compares them with the fixture's)."""
    import hashlib

    return {f: hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest() for f in SOURCES}


def main(iters=100):
    from bench import make_cfg
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_lr_scheduler, build_optimizer

    cfg = make_cfg(18)
    cfg.SOLVER.IMS_PER_BATCH = 2
    cfg.MODEL.DEVICE = "cpu"
    pool = [synthetic_batch(2, 512, 512, 100 + i, device="cpu") for i in range(4)]

    def build():
        torch.manual_seed(7)
        model = build_model(cfg)
        return model

    model = build()
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], cfg.SOLVER.BASE_LR)
    sched = build_lr_scheduler(cfg, opt)
    lrs = []
    for _ in range(iters):
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()

    def run(tag, threads, double, emu=False):
        torch.set_num_threads(threads)
        oracle = OracleFCOS.from_hip_model(build(), emulate_bf16=emu)
        if double:
            oracle.double()
        state, out = {}, []
        for it in range(iters):
            ref = oracle.losses(pool[it % len(pool)])
            total = sum(ref.values())
            grads = dict(zip(oracle.trainable().keys(), torch.autograd.grad(total, list(oracle.trainable().values()))))
            oracle.sgd_step(grads, state, lrs[it], cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM)
            out.append(float(total))
            if it % 10 == 9:
                print(tag, it + 1, out[-1], flush=True)
        return out

    ncpu = os.cpu_count() or 1
    path = os.path.join(ROOT, "tests", "golden", "chaos100.json")
    if "stamp" in sys.argv[1:]:    # the sources changed but the arithmetic did not: re-run the first iterations of both oracles, then re-stamp
        res = json.load(open(path))
        assert res["lrs"] == lrs
        iters = 4
        live32, live_emu = run("f32/all", ncpu, False), run("emu/all", ncpu, False, emu=True)
        d32 = max(abs(a - b) for a, b in zip(live32, res["f32_all_threads"]))
        demu = max(abs(a - b) for a, b in zip(live_emu, res["emu_all_threads"]))
        print("first iterations against the fixture: f32", d32, "emu", demu)
        assert d32 <= 2e-5 and demu <= 1e-3, "the oracle's arithmetic changed: regenerate the fixture (no arguments, then `emu`)"
        res["sources_sha256"] = source_hashes()
        json.dump(res, open(path, "w"))
        return
    if "emu" in sys.argv[1:]:      # add the bf16-storage-emulating oracle's trajectory (the `emu` run of tests/test_gpu_parity100.py) to the file
        res = json.load(open(path))
        assert res["lrs"] == lrs
        res["emu_all_threads"] = run("emu/all", ncpu, False, emu=True)
        res["sources_sha256"] = source_hashes()
        json.dump(res, open(path, "w"))
        return
    res = {"lrs": lrs, "f32_all_threads": run("f32/all", ncpu, False), "f32_one_thread": run("f32/1", 1, False), "f64": run("f64", ncpu, True),
           "sources_sha256": source_hashes()}
    json.dump(res, open(os.path.join(ROOT, "tests", "golden", "chaos100.json"), "w"))
    a, b, c = res["f32_all_threads"], res["f32_one_thread"], res["f64"]
    for i in list(range(9, iters, 10)) + [iters - 1]:
        print(f"it {i + 1:3d}  |f32 - f32'| {abs(a[i] - b[i]):.2e}  |f32 - f64| {abs(a[i] - c[i]):.2e}  |f32' - f64| {abs(b[i] - c[i]):.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100)
