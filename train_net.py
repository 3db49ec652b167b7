#!/usr/bin/env python
"""Training entry point with the reference's CLI (train_net.py:145-195 of wanzysky/SlenderObjDet):

    python train_net.py --config-file configs/fcos/fcos_R_50_FPN_1x.yaml --num-gpus 8 SOLVER.MAX_ITER 100

Evaluation (--eval-only) needs datasets and the COCO evaluator, which are out of scope for this package."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from slenderobjdet_amd.config import get_cfg
from slenderobjdet_amd.engine import BaseTrainer, default_argument_parser, default_setup, launch


class Trainer(BaseTrainer):
    pass


def setup(args):
    cfg = get_cfg()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    default_setup(cfg, args)
    return cfg


def main(args):
    cfg = setup(args)
    if args.eval_only:
        raise NotImplementedError("--eval-only: dataset evaluation is outside the training hot path this package covers")
    trainer = Trainer(cfg)
    trainer.resume_or_load(resume=args.resume)
    return trainer.train()


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    launch(main, args.num_gpus, num_machines=args.num_machines, machine_rank=args.machine_rank, dist_url=args.dist_url, args=(args,))
