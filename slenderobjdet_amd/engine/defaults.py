"""default_argument_parser / default_setup / trainers (reference: train_net.py:145-195, slender_det/engine/defaults.py:22-178;
detectron2 DefaultTrainer / SimpleTrainer semantics from SURVEY.md C.16).

The trainer owns the MI355X-specific step: backward is bracketed by ``arena.begin_backward()/finish_backward()`` (bucketed RCCL
all-reduce of the flat gradient arena overlapped with backward) instead of wrapping the model in DistributedDataParallel.
"""
import argparse
import logging
import os
import sys

import torch

from ..utils import comm
from . import hooks


def default_argument_parser(epilog=None):
    parser = argparse.ArgumentParser(epilog=epilog, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    parser.add_argument("--resume", action="store_true", help="whether to attempt to resume from the checkpoint directory")
    parser.add_argument("--eval-only", action="store_true", help="perform evaluation only")
    parser.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine*")
    parser.add_argument("--num-machines", type=int, default=1, help="total number of machines")
    parser.add_argument("--machine-rank", type=int, default=0, help="the rank of this machine (unique per machine)")
    port = 2 ** 15 + 2 ** 14 + hash(os.getuid() if sys.platform != "win32" else 1) % 2 ** 14
    parser.add_argument("--dist-url", default=f"tcp://127.0.0.1:{port}", help="initialization URL for pytorch distributed backend")
    parser.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return parser


def setup_logger(output=None, distributed_rank=0, name="slender_det", abbrev_name=None):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if not logger.handlers and distributed_rank == 0:
        ch = logging.StreamHandler(stream=sys.stdout)
        ch.setFormatter(logging.Formatter("[%(asctime)s] %(name)s %(levelname)s: %(message)s", datefmt="%m/%d %H:%M:%S"))
        logger.addHandler(ch)
        if output:
            os.makedirs(output, exist_ok=True)
            fh = logging.FileHandler(os.path.join(output, "log.txt"))
            fh.setFormatter(ch.formatter)
            logger.addHandler(fh)
    return logger


def seed_all_rng(seed=None):
    import random

    import numpy as np

    if seed is None:
        seed = int.from_bytes(os.urandom(2), "big") + os.getpid()
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def default_setup(cfg, args):
    """slender_det/engine/defaults.py:22-71: logger, config dump, per-rank seed ``SEED + rank``."""
    output_dir = cfg.OUTPUT_DIR
    if comm.is_main_process() and output_dir:
        os.makedirs(output_dir, exist_ok=True)
    rank = comm.get_rank()
    logger = setup_logger(output_dir, distributed_rank=rank, name="slender_det")
    logger.info("Rank of current process: {}. World size: {}".format(rank, comm.get_world_size()))
    logger.info("Command line arguments: " + str(args))
    if comm.is_main_process() and output_dir:
        path = os.path.join(output_dir, "config.yaml")
        with open(path, "w") as f:
            f.write(cfg.dump())
        logger.info("Full config saved to {}".format(path))
    seed_all_rng(None if cfg.SEED < 0 else cfg.SEED + rank)


class DefaultTrainer:
    """Training loop with the DefaultTrainer/SimpleTrainer contract: classmethod builders, hooks, ``train()``."""

    def __init__(self, cfg):
        logger = logging.getLogger("slender_det")
        self.cfg = cfg
        self.model = self.build_model(cfg)
        self.model.train()
        self.optimizer = self.build_optimizer(cfg, self.model)
        self.data_loader = self.build_train_loader(cfg)
        self._data_loader_iter = iter(self.data_loader)
        world = comm.get_world_size()
        arena = getattr(self.model, "arena", None)
        if world > 1:
            if arena is None:
                raise RuntimeError("multi-GPU training needs the flat parameter arena (MODEL.DEVICE must be a GPU)")
            torch.distributed.broadcast(arena.params, src=0)     # DDP semantics: identical replicas
            arena.bump()
            if hasattr(self.optimizer, "grad_scale"):
                self.optimizer.grad_scale = 1.0 / world
        self.scheduler = self.build_lr_scheduler(cfg, self.optimizer)
        self.checkpointer = _Checkpointer(self.model, cfg.OUTPUT_DIR, cfg=cfg, optimizer=self.optimizer, scheduler=self.scheduler)
        self.start_iter, self.max_iter, self.iter = 0, cfg.SOLVER.MAX_ITER, 0
        self.storage = {}
        self._hooks = []
        self.register_hooks(self.build_hooks())
        logger.info("Trainer ready: %d trainable parameters", sum(p.numel() for p in self.model.parameters() if p.requires_grad))

    # ---- hooks -------------------------------------------------------------------------------
    def register_hooks(self, hks):
        for h in [h for h in hks if h is not None]:
            h.trainer = self
            self._hooks.append(h)

    def build_hooks(self):
        ret = [hooks.IterationTimer(), hooks.LRScheduler(self.optimizer, self.scheduler)]
        if comm.is_main_process():
            ret.append(hooks.PeriodicCheckpointer(self.checkpointer, self.cfg.SOLVER.CHECKPOINT_PERIOD))
            ret.append(hooks.PeriodicWriter(20, logging.getLogger("slender_det").info))
        return ret

    # ---- builders (overridable classmethods, as in the reference) ----------------------------
    @classmethod
    def build_model(cls, cfg):
        from ..modeling import build_model

        return build_model(cfg)

    @classmethod
    def build_optimizer(cls, cfg, model):
        from ..solver import build_optimizer

        return build_optimizer(cfg, model)

    @classmethod
    def build_lr_scheduler(cls, cfg, optimizer):
        from ..solver import build_lr_scheduler

        return build_lr_scheduler(cfg, optimizer)

    @classmethod
    def build_train_loader(cls, cfg):
        """Dataset loading is out of scope: synthetic COCO-shaped batches with the reference's
        ``batched_inputs`` contract, ``IMS_PER_BATCH / world`` images per rank."""
        from ..data import SyntheticCocoBatches

        per_rank = max(cfg.SOLVER.IMS_PER_BATCH // comm.get_world_size(), 1)
        side = cfg.INPUT.MIN_SIZE_TRAIN[-1] if isinstance(cfg.INPUT.MIN_SIZE_TRAIN, (tuple, list)) else cfg.INPUT.MIN_SIZE_TRAIN
        return SyntheticCocoBatches(per_rank, side, cfg.INPUT.MAX_SIZE_TRAIN, rank=comm.get_rank(),
                                    base_seed=1234 if cfg.SEED < 0 else cfg.SEED, device=cfg.MODEL.DEVICE, pool=4)

    # ---- loop ------------------------------------------------------------------------------------
    def resume_or_load(self, resume=True):
        self.start_iter = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        self.iter = self.start_iter

    def run_step(self):
        # one batch of look-ahead: between this step's forward and backward the model runs the NEXT batch's preprocess + frozen
        # backbone prefix on a side stream (meta-arch ``prefetch``; bench.py's train_step does the same)
        data = self._next_data if getattr(self, "_next_data", None) is not None else next(self._data_loader_iter)
        self._next_data = None
        loss_dict = self.model(data)
        if hasattr(self.model, "prefetch") and self.iter + 1 < self.max_iter:
            try:
                self._next_data = next(self._data_loader_iter)
            except StopIteration:
                self._next_data = None
            if self._next_data is not None:
                self.model.prefetch(self._next_data)
        losses = sum(loss_dict.values())
        self.optimizer.zero_grad()
        arena = getattr(self.model, "arena", None)
        if arena is not None:
            arena.begin_backward()
        losses.backward()
        if arena is not None:
            arena.finish_backward()
        self.optimizer.step()
        if (self.iter + 1) % 20 == 0 or self.iter == self.max_iter - 1:     # metrics only when they are written (avoids a host sync per step)
            red = comm.reduce_dict({k: v.detach() for k, v in loss_dict.items()})
            if comm.is_main_process():
                vals = {k: float(v) for k, v in red.items()}
                if not all(v == v and abs(v) != float("inf") for v in vals.values()):
                    raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {vals}")
                self.storage["losses"], self.storage["total_loss"] = vals, sum(vals.values())

    def train(self):
        for h in self._hooks:
            h.before_train()
        try:
            for self.iter in range(self.start_iter, self.max_iter):
                for h in self._hooks:
                    h.before_step()
                self.run_step()
                for h in self._hooks:
                    h.after_step()
        finally:
            for h in self._hooks:
                h.after_train()
        return self.storage


class BaseTrainer(DefaultTrainer):
    """slender_det/engine/defaults.py:74-178: the reference's thin wrapper (its checkpointer swap and evaluation hooks touch
    I/O and datasets, which are out of scope; the builders route to this package's build_model / build_optimizer)."""


class _Checkpointer:
    """DetectionCheckpointer stand-in (slender_det/checkpoint/detection_checkpoint.py): rank-0 torch.save of model / optimizer /
    scheduler state; loads native checkpoints and - through slenderobjdet_amd.checkpoint - reference / detectron2 ``.pth`` / ``.pkl``
    files (layout conversion, incompatible keys logged).  Remote paths (detectron2://, https://) need the reference's PathManager
    and a network, neither of which this package has: they raise instead of silently training from random weights."""

    def __init__(self, model, save_dir="", cfg=None, **checkpointables):
        # cfg: the trainer's config, for MODEL.WEIGHTS_ALLOW_MISSING (key prefixes a checkpoint may lack: DetectionCheckpointer's
        # warn-and-continue for fine-tuning from a detector with another head)
        self.model, self.save_dir, self.cfg, self.checkpointables = model, save_dir, cfg, checkpointables

    def save(self, name, **extra):
        if not self.save_dir or not comm.is_main_process():
            return
        from ..checkpoint import NATIVE_FORMAT

        os.makedirs(self.save_dir, exist_ok=True)
        data = {"model": self.model.state_dict(), "__format__": NATIVE_FORMAT,
                **{k: v.state_dict() for k, v in self.checkpointables.items() if hasattr(v, "state_dict")}, **extra}
        torch.save(data, os.path.join(self.save_dir, name + ".pth"))
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(name + ".pth")

    def save_reference_format(self, name):
        """Writes ``<name>.pth`` with the REFERENCE's key names / layouts, loadable by its DetectionCheckpointer."""
        if not self.save_dir or not comm.is_main_process():
            return
        from ..checkpoint import native_to_reference

        os.makedirs(self.save_dir, exist_ok=True)
        torch.save({"model": native_to_reference(self.model)}, os.path.join(self.save_dir, name + ".pth"))

    def resume_or_load(self, path, resume=True):
        import logging

        from ..checkpoint import load_into

        log = logging.getLogger(__name__)
        last = os.path.join(self.save_dir or ".", "last_checkpoint")
        if resume and os.path.exists(last):
            path = os.path.join(self.save_dir, open(last).read().strip())
        if not path:
            log.info("no checkpoint given (MODEL.WEIGHTS is empty): training starts from the random initialisation")
            return 0
        if "://" in path:
            raise FileNotFoundError(f"MODEL.WEIGHTS = {path!r}: remote checkpoints are not supported (no network / PathManager); "
                                    "download the file and pass its local path")
        if not os.path.exists(path):
            raise FileNotFoundError(f"MODEL.WEIGHTS = {path!r} does not exist")
        allow = tuple(getattr(getattr(self.cfg, "MODEL", None), "WEIGHTS_ALLOW_MISSING", ()) or ()) if self.cfg is not None else ()
        report, meta = load_into(self.model, path, allow_missing=allow)
        data = meta["raw"]
        if resume and meta["native"]:
            for k, v in self.checkpointables.items():
                if k in data and hasattr(v, "load_state_dict"):
                    v.load_state_dict(data[k])
            return int(data.get("iteration", -1)) + 1
        return 0
