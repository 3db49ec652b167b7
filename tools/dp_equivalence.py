"""One data-parallel training step of FCOS on a FIXED set of images, split over the ranks that run it.

    python tools/dp_equivalence.py --out DIR [--depth 18] [--images 4] [--height 256] [--width 320]            # one process, all images
    python -m torch.distributed.run --nproc-per-node 2 ... tools/dp_equivalence.py --out DIR ...                # rank r takes its share

Every rank builds the model from the same seed, takes images [r * n / world, (r + 1) * n / world) of the same seeded batch, runs ONE
step (forward, backward with the bucketed gradient all-reduce, fused SGD with 1 / world) and writes ``rank{r}.pt`` = {loss: this
rank's three losses, params: the updated flat parameter arena}.  tests/test_gpu_model.py::test_two_ranks_equal_one_process_on_the_union
compares the two launches: the property the folded [num_pos, sum centerness] all-reduce (fcos/utils.py:10-19, fcosv2.py:115-118,
132-133) and the gradient SUM x 1 / world (train_net.py:185-195 -> DistributedDataParallel) exist for.
SOD_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo (the one-GPU test box); otherwise one rank per GPU over RCCL."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    share = os.environ.get("SOD_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    from bench import make_cfg, train_step
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_optimizer

    assert args.images % world == 0
    cfg = make_cfg(args.depth)
    cfg.SOLVER.BASE_LR = args.lr
    torch.manual_seed(1)              # the SAME initial parameters on every rank and in the one-process run
    model = build_model(cfg)
    model.train()
    if world > 1:
        dist.broadcast(model.arena.params, src=0)
        model.arena.bump()
    opt = build_optimizer(cfg, model)
    for g in opt.param_groups:
        g["lr"] = args.lr
    opt.grad_scale = 1.0 / world
    data = synthetic_batch(args.images, args.height, args.width, 4242, device="cuda")
    per = args.images // world
    mine = data[rank * per:(rank + 1) * per]
    losses = model(mine)
    total = sum(losses.values())
    opt.zero_grad()
    model.arena.begin_backward()
    total.backward()
    model.arena.finish_backward()
    opt.step()
    torch.cuda.synchronize()
    os.makedirs(args.out, exist_ok=True)
    torch.save({"loss": {k: float(v.detach()) for k, v in losses.items()}, "params": model.arena.params.detach().float().cpu(),
                "world": world}, os.path.join(args.out, f"rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
