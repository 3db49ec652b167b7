"""world_size-2 data-parallel gradient exchange on CPU (gloo): the arena's bucketed all-reduce, launched from the
backward-side ``mark_ready`` notifications, must leave the SUM of the per-rank gradients on every rank; and the folded
(num_pos, sum_ctr) normaliser all-reduce must match the reference's two reduce_sum calls (fcos/utils.py:14-19)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, store_file, out, bucket_dtype="fp32", collective="all_reduce"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    # a FileStore: no TCP port to guess (bind-to-0-then-close is a race)
    dist.init_process_group("gloo", init_method="file://" + store_file, rank=rank, world_size=world)
    try:
        from slenderobjdet_amd.layers.arena import ParamArena
        from slenderobjdet_amd.utils import comm

        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 100), torch.nn.Linear(100, 50), torch.nn.Linear(50, 10))
        arena = ParamArena(model, bucket_mb=0.01, bucket_dtype=bucket_dtype, collective=collective)   # several buckets
        assert arena.collective == collective
        assert len(arena.buckets) >= 3
        params = [p for p in model.parameters()]
        # forward "uses": last layer used twice (like the head weights shared by FPN levels)
        for p in params:
            arena.note_use(p)
        for p in params[-2:]:
            arena.note_use(p)
        arena.zero_grad()
        arena.begin_backward()
        g = torch.Generator().manual_seed(100 + rank)
        expect = {}
        for use in range(2):
            for p in reversed(params):
                if use == 1 and not any(p is q for q in params[-2:]):
                    continue
                contrib = torch.randn(p.shape, generator=g)
                arena.grad_view(p).add_(contrib)
                expect[id(p)] = expect.get(id(p), 0) + contrib
                arena.mark_ready(p)
        arena.finish_backward()
        # every rank now holds the sum over ranks; rebuild the expectation with an explicit all_reduce
        ok = True
        for p in params:
            e = expect[id(p)].clone()
            if bucket_dtype == "bf16":      # the wire carries the per-rank gradient rounded to bf16; the reduced values are what every rank gets
                e = e.to(torch.bfloat16)
                dist.all_reduce(e)
                ok = ok and torch.allclose(arena.grad_view(p), e.float(), rtol=2e-2, atol=2e-2)
            else:
                dist.all_reduce(e)
                ok = ok and torch.allclose(arena.grad_view(p), e, atol=1e-6)
                if world == 2:          # a + b is one rounding whichever rank forms it: reduce-scatter + all-gather == all-reduce, bit for bit
                    ok = ok and torch.equal(arena.grad_view(p), e)
        # replicas must hold bit-identical gradients (same reduced values on every rank), whatever the wire format
        mine = arena.grads.clone()
        other = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        ok = ok and all(torch.equal(o, other[0]) for o in other)
        # normaliser exchange: one 2-element all-reduce == two scalar reduce_sum calls
        stats = torch.tensor([3.0 + rank, 1.5 * (rank + 1)])
        folded = stats.clone()
        dist.all_reduce(folded)
        a = comm.reduce_sum(stats[0:1].clone())
        b = comm.reduce_sum(stats[1:2].clone())
        ok = ok and torch.allclose(folded, torch.cat([a, b])) and comm.get_world_size() == 2 and comm.get_num_gpus() == 2
        out.put((rank, bool(ok)))
    except Exception as e:   # surface the failure instead of letting the parent time out
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["all_reduce", "rs_ag"])
@pytest.mark.parametrize("bucket_dtype", ["fp32", "bf16"])
def test_arena_bucketed_allreduce_world2(bucket_dtype, collective, tmp_path):
    """``collective`` = the shape of the exchange (SURVEY.md section 8 e): one all-reduce per bucket, or reduce-scatter + all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path / "store"), q, bucket_dtype, collective)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)]
