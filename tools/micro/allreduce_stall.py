"""What a c10d all-reduce costs the OTHER streams of the process (one-rank RCCL group on a one-GPU box).

A compute stream runs a chain of short kernels; every `every` kernels a collective is issued on a communication stream that nothing waits
for.  Prints the compute chain's time without collectives, with dist.all_reduce(async_op=True), and with ncclAllReduce called directly on
the communication stream (ctypes on the librccl.so torch ships).  Usage: python tools/micro/allreduce_stall.py"""
import ctypes
import os
import sys
import time

import torch
import torch.distributed as dist



class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def direct_comm():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    uid = UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib, comm


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
    lib, comm = direct_comm()
    x = torch.ones(16 << 20, device=dev)            # 64 MB: ~30 us per pass
    side_x = torch.ones(16 << 20, device=dev)
    bucket = torch.ones(8 << 20, device=dev)        # 32 MB "gradient bucket"
    comm_stream, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    n, every = 400, 40

    def chain(kind, main_stream):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        handles = []
        with torch.cuda.stream(main_stream):
            e0.record()
            for i in range(n):
                x.mul_(1.0)
                with torch.cuda.stream(side):           # a second busy compute queue, as the weight-gradient stream is
                    side_x.mul_(1.0)
                if kind != "none" and i % every == every - 1:
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(comm_stream):
                        comm_stream.wait_event(ev)
                        if kind == "c10d":
                            handles.append(dist.all_reduce(bucket, async_op=True))
                        elif kind == "direct":
                            rc = lib.ncclAllReduce(bucket.data_ptr(), bucket.data_ptr(), bucket.numel(), 7, 0, comm, ctypes.c_void_p(comm_stream.cuda_stream))
                            assert rc == 0, rc
                        elif kind == "events_only":
                            pass
            e1.record()
        t_host = time.perf_counter()
        for h in handles:
            h.wait()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    for main_name, ms in (("default stream", torch.cuda.default_stream(dev)), ("pool stream", torch.cuda.Stream(dev))):
        for kind in ("none", "events_only", "c10d", "direct", "none", "c10d", "direct"):
            chain(kind, ms)
            t = min(chain(kind, ms) for _ in range(3))
            print(f"main = {main_name:14s}  collectives = {kind:12s}: {t:8.3f} ms for {n} kernels ({n // every} collectives)", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
