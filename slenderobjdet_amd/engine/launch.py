"""detectron2.engine.launch: one process per GPU, RCCL ("nccl") process group, then ``main_func``."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _find_free_port():
    import socket

    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=()):
    world_size = num_machines * num_gpus_per_machine
    if world_size > 1:
        if dist_url == "auto" or dist_url is None:
            assert num_machines == 1, "dist_url=auto not supported in multi-machine jobs."
            dist_url = f"tcp://127.0.0.1:{_find_free_port()}"
        mp.spawn(_distributed_worker, nprocs=num_gpus_per_machine,
                 args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args), daemon=False)
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
