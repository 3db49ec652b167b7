"""``detectron2.utils.comm`` surface used by the reference (train_net.py:23, engine/defaults.py:7, fcos/utils.py:10-19):
rank / world size helpers, barrier, dict reduction.  One process per GPU; backend "nccl" is RCCL on ROCm."""
import functools
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist() else 0


def get_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", 0)) if is_dist() else 0


def is_main_process() -> bool:
    return get_rank() == 0


# A process group of ONE rank normally takes the single-process shortcuts (no collective is issued).  ``FORCE_COLLECTIVES`` (bench.py
# --rccl-rehearsal, tests) makes the data-parallel code issue them anyway: on a one-GPU box that runs the real RCCL calls - the
# asynchronous normaliser all-reduce, the bucketed gradient all-reduce from the autograd thread, their stream hand-overs - which a
# two-ranks-on-one-GPU gloo test cannot.
FORCE_COLLECTIVES = False

# HIP hardware queues of a data-parallel rank.  The ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4).  A one-GPU step uses exactly four streams (main, weight gradients, box tower, frozen-prefix look-ahead); a data-parallel
# rank adds the bucket stream of layers/arena.py and c10d's own RCCL stream, which then SHARE hardware queues with compute streams: the
# bucket stream's "wait until every compute stream has produced this bucket" barrier sits in the main stream's queue, and the main stream
# stalls until the weight-gradient stream has drained - at every bucket.  Measured in the one-GPU rehearsal (bench.py --rccl-rehearsal,
# FCOS R50, ms per step): 4 queues 25.73, 5: 26.75, 6: 24.99 (the plain step: 24.71), 7: 30.1, 8: 30.3, 16: 30.6.
# Six = one queue per stream; more is pathological on this runtime, so the value is set exactly, before the HIP runtime starts.
HW_QUEUES_DATA_PARALLEL = 6


HIP_STARTED_BEFORE_PREPARE = False


def _hip_runtime_started():
    """True when the HIP runtime of this process has (probably) read its environment already: torch has initialised the device, or a
    profiler's preloaded tool library is mapped (rocprofv3 initialises HIP before Python starts, so a value set here comes too late and
    the run uses whatever the SHELL exported - set GPU_MAX_HW_QUEUES in front of ``rocprofv3``, tools/profile_*.sh do)."""
    import torch

    if torch.cuda.is_initialized():
        return True
    try:
        with open("/proc/self/maps") as f:
            maps = f.read()
    except OSError:
        return False
    return any(t in maps for t in ("librocprofiler-sdk-tool", "librocprofiler64", "libroctracer64"))


def prepare_rank_env(world_size, rehearsal=False):
    """Call BEFORE the first HIP call of the process (``import torch`` and ``torch.cuda.device_count()`` are fine): settings the HIP
    runtime reads once at start-up.  An explicit GPU_MAX_HW_QUEUES in the environment wins."""
    global HIP_STARTED_BEFORE_PREPARE
    HIP_STARTED_BEFORE_PREPARE = _hip_runtime_started() and "GPU_MAX_HW_QUEUES" not in os.environ
    if world_size > 1 or rehearsal:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(HW_QUEUES_DATA_PARALLEL))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform (RCCL / tensor sharing across ranks)


def collectives_active() -> bool:
    return get_world_size() > 1 or (FORCE_COLLECTIVES and is_dist())


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def get_num_gpus():
    """slender_det/modeling/meta_arch/fcos/utils.py:10-11 reads WORLD_SIZE from the environment."""
    return int(os.environ["WORLD_SIZE"]) if "WORLD_SIZE" in os.environ else 1


def reduce_sum(tensor):
    """fcos/utils.py:14-19 (all_reduce SUM of a clone when world > 1)."""
    if get_world_size() <= 1:
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def reduce_dict(input_dict, average=True):
    """Reduce scalar tensors of a dict to rank 0 (d2 SimpleTrainer._write_metrics)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return {k: v for k, v in zip(names, values)}


@functools.lru_cache()
def _gloo_group():
    return dist.new_group(backend="gloo") if dist.get_backend() == "nccl" else dist.group.WORLD


def gather(data, dst=0):
    """Gather picklable objects on ``dst`` (evaluation only; slender_det/evaluation/coco_evaluation.py:83)."""
    if get_world_size() == 1:
        return [data]
    out = [None] * get_world_size() if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst, group=_gloo_group())
    return out if get_rank() == dst else []
