"""detectron2.engine.launch: one process per GPU, RCCL ("nccl") process group, then ``main_func``."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _auto_dist_url():
    """Rendezvous for the ranks of ONE machine without guessing a TCP port: a FileStore in a fresh temporary directory.  (detectron2
    binds a probe socket to port 0, closes it and passes the number on; whatever takes the port in between makes every rank fail with
    EADDRINUSE - seen on the GPU box.)"""
    import tempfile

    return "file://" + os.path.join(tempfile.mkdtemp(prefix="sod_rdzv_"), "store")


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=()):
    world_size = num_machines * num_gpus_per_machine
    if world_size > 1:
        own_dir = None
        if dist_url == "auto" or dist_url is None:
            assert num_machines == 1, "dist_url=auto not supported in multi-machine jobs."
            dist_url = _auto_dist_url()
            own_dir = os.path.dirname(dist_url[len("file://"):])
        try:
            mp.spawn(_distributed_worker, nprocs=num_gpus_per_machine,
                     args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args), daemon=False)
        finally:
            if own_dir:
                import shutil
                shutil.rmtree(own_dir, ignore_errors=True)
    else:
        main_func(*args)


def _distributed_worker(local_rank, main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args):
    use_gpu = torch.cuda.device_count() >= num_gpus_per_machine   # device_count() does not initialise the GPU
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    os.environ["WORLD_SIZE"] = str(world_size)      # fcos/utils.py:10-11 reads it
    os.environ["RANK"] = str(global_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    from ..utils.comm import prepare_rank_env
    prepare_rank_env(world_size)                    # before the first HIP call of this rank (set_device below)
    if use_gpu:
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl" if use_gpu else "gloo", init_method=dist_url, world_size=world_size, rank=global_rank)
    try:
        dist.barrier()
        main_func(*args)
    finally:
        dist.destroy_process_group()
