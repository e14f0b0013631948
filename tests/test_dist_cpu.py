"""world_size-2 gloo tests of the data-parallel plumbing (hip-ad_amd/dist.py) on CPU: flat gradient
views, one-collective mean all-reduce (fp32 and bf16 wire format), parameter broadcast, max-over-ranks
timing -- the N>1 path of bench.py without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hipad_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must equalise
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    D.broadcast_parameters(model)
    ref = [p.detach().clone() for p in model.parameters()]
    gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
    dist.all_gather(gathered, ref[0])
    same_params = all(torch.equal(g, gathered[0]) for g in gathered)
    for comm in (None, torch.bfloat16):
        fg = D.FlatGrads(list(model.parameters()), comm_dtype=comm)
        fg.zero()
        x = torch.full((3, 8), float(rank + 1))
        model(x).sum().backward()
        fg.check_views()
        local = fg.flat.clone()
        fg.all_reduce_mean()
        # expected: mean over ranks of the local gradients
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        expect = torch.stack(both).mean(0)
        tol = 1e-6 if comm is None else 2e-2
        ok = torch.allclose(fg.flat, expect, rtol=tol, atol=tol * float(expect.abs().max()))
        views_ok = all(p.grad.data_ptr() >= fg.flat.data_ptr() for p in model.parameters())
        q.put((rank, str(comm), bool(ok), bool(views_ok), bool(same_params)))
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, "max", t == float(world), True, True))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = [q.get(timeout=5) for _ in range(world * 3)]
    assert all(ok and views and same for _, _, ok, views, same in results), results


def test_single_process_is_a_noop():
    from hipad_amd import dist as D
    m = torch.nn.Linear(4, 4)
    fg = D.FlatGrads(list(m.parameters()))
    m(torch.ones(2, 4)).sum().backward()
    before = fg.flat.clone()
    fg.all_reduce_mean()
    assert torch.equal(before, fg.flat)
    assert D.max_over_ranks(2.5, torch.device("cpu")) == 2.5
