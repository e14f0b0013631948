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


def _worker_segments_and_counts(rank, world, port, q):
    """(a) the two-segment all-reduce of FlatGrads, (b) the three-phase positive-count exchange around the losses:
    every rank evaluates the reference loss on ITS sample with counts averaged over the ranks; the mean of the ranks'
    loss terms must equal the single-process value on the concatenated batch (what the reference's DDP + reduce_mean
    computes, sparse_onedecoder.py:1134, 1190, 1292)."""
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    sys.path.insert(0, here)
    from hipad_amd import dist as D
    from hipad_amd.compat import count_exchange
    D.init_from_env("gloo")
    # (a) segments
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    params = list(model.parameters())
    fg = D.FlatGrads(params)
    fg.set_split(params[2])
    model(torch.full((3, 8), float(rank + 1))).sum().backward()
    fg.check_views()
    local = fg.flat.clone()
    fg.all_reduce_mean(segment="early")
    mid = fg.flat.clone()
    fg.all_reduce_mean(segment="late")
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    expect = torch.stack(both).mean(0)
    seg_ok = bool(torch.allclose(fg.flat, expect, rtol=1e-6, atol=1e-7)
                  and torch.equal(mid[fg.split:], local[fg.split:]) and torch.allclose(mid[:fg.split], expect[:fg.split], rtol=1e-6, atol=1e-7))
    # (b) losses with exchanged counts
    import loss_case as LC
    from test_losses import build_criterion
    crit = build_criterion()
    outs = LC.head_outputs()
    data = LC.ground_truth()

    def take(obj, i):
        if isinstance(obj, torch.Tensor):
            return obj[i:i + 1]
        if isinstance(obj, dict):
            return {k: take(v, i) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [None if v is None else (take(v, i)) for v in obj]
        return obj

    mine = [take(o, rank) for o in outs]
    dmine = {k: (v[rank:rank + 1] if isinstance(v, (list, torch.Tensor)) else v) for k, v in data.items()}
    count_exchange.begin("collect")
    crit.positive_counts(*mine, dmine)
    count_exchange.begin("direct")
    count_exchange.all_reduce()
    count_exchange.begin("use")
    losses = crit.loss(*mine, dmine)
    count_exchange.begin("direct")
    keys = sorted(losses)
    vec = torch.stack([losses[k].detach().double() for k in keys])
    dist.all_reduce(vec)
    vec /= world
    if rank == 0:
        ref_crit = build_criterion()                      # single process, both samples: local counts, no collective
        count_exchange.begin("collect")
        ref_crit.positive_counts(*outs, data)
        count_exchange.begin("use")
        whole = ref_crit.loss(*outs, data)
        count_exchange.begin("direct")
        ref = torch.stack([whole[k].detach().double() for k in keys])
        # terms normalised by a positive count are sums / count: the rank mean with a shared count equals the whole-batch
        # value; terms that are plain means over the batch (ego, plan) agree as well
        rel = ((vec - ref).abs() / ref.abs().clamp_min(1e-9))
        q.put((rank, "loss", bool((rel < 2e-5).all()), {k: float(r) for k, r in zip(keys, rel) if r >= 2e-5}))
    q.put((rank, "segments", seg_ok, {}))
    dist.destroy_process_group()


def test_segment_allreduce_and_count_exchange_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments_and_counts, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    results = [q.get(timeout=5) for _ in range(world + 1)]
    assert all(ok for _, _, ok, _ in results), results


def test_reduce_mean_refuses_to_run_inside_a_capture_and_count_exchange_phases():
    from hipad_amd.compat import CountExchange
    ex = CountExchange()
    with pytest.raises(RuntimeError):
        ex.begin("use")                      # nothing collected yet
    ex.begin("collect")
    a = torch.tensor([3.0, 5.0])
    assert ex.exchange(a) is a and ex.filled == 2
    ex.begin("use")
    assert torch.equal(ex.exchange(torch.zeros(2)), a)   # single process: the collected values come back
    ex.begin("direct")


def _worker_replay_schedule(rank, world, port, q):
    """GraphedTrainStep._replay at world 2 with the backward split at the pyramid cut: the real control flow (and the
    real TrainStep.reduce_early / reduce_late / exchange_counts) around stand-in graphs that do a toy model's work on
    the host -- [F | counts | L: decoder backward | early all-reduce || E: encoder backward | late all-reduce | B]."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hipad_amd import dist as D
    from hipad_amd.compat import count_exchange
    from hipad_amd.frame import GraphedTrainStep, TrainStep
    D.init_from_env("gloo")
    torch.manual_seed(3)
    decoder, encoder = torch.nn.Linear(8, 4), torch.nn.Linear(6, 8)      # flat layout: [decoder | encoder]
    params = list(decoder.parameters()) + list(encoder.parameters())
    fg = D.FlatGrads(params)
    fg.set_split(params[2])
    order, state = [], {}
    inner = object.__new__(TrainStep)
    inner.grads, inner.world, inner._comm_stream = fg, world, None
    plain_all_reduce = fg.all_reduce_mean
    fg.all_reduce_mean = lambda group=None, segment=None: (order.append("reduce_" + str(segment)), plain_all_reduce(group, segment))[1]

    class Graph:
        def __init__(self, name, fn):
            self.name, self.fn = name, fn

        def replay(self):
            order.append(self.name)
            self.fn()

    def forward():
        x = torch.full((3, 6), float(rank + 1))
        # the "pyramid levels": a VIEW of the encoder's output, as the detector's reshaped levels are -- capturing the
        # gradient at a node's own output makes torch 2.10 release that node's buffers in the first backward
        state["cut"] = encoder(x).view(3, 8)
        state["loss"] = decoder(state["cut"]).square().sum()
        count_exchange.begin("collect")
        count_exchange.exchange(torch.tensor([float(rank + 1)]))
        count_exchange.begin("direct")

    def decoder_backward():
        count_exchange.begin("use")
        state["count"] = float(count_exchange.exchange(torch.zeros(1)))
        count_exchange.begin("direct")
        torch.autograd.backward([state["loss"]], inputs=list(decoder.parameters()) + [state["cut"]])

    def encoder_backward():
        state["early_at_e"] = fg.flat[:fg.split].clone()        # must already be the cross-rank mean
        torch.autograd.backward([state["cut"]], [state["cut"].grad])

    g = object.__new__(GraphedTrainStep)
    g.inner, g.world = inner, world
    g.graph_f, g.graph_l = Graph("F", forward), Graph("L", decoder_backward)
    g.graph_e, g.graph_b = Graph("E", encoder_backward), Graph("B", lambda: state.__setitem__("final", fg.flat.clone()))
    plain_exchange = inner.exchange_counts
    inner.exchange_counts = lambda: (order.append("counts"), plain_exchange())[1]
    # reference: local gradient of this rank, then the mean over the ranks
    forward()
    state["loss"].backward()
    local = fg.flat.clone()
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    expect = torch.stack(both).mean(0)
    fg.zero()
    g._replay()
    ok_order = order == ["F", "counts", "L", "reduce_early", "E", "reduce_late", "B"]
    ok_vals = bool(torch.allclose(state["final"], expect, rtol=1e-6, atol=1e-7))
    ok_early = bool(torch.allclose(state["early_at_e"], expect[:fg.split], rtol=1e-6, atol=1e-7))
    ok_count = abs(state["count"] - (1 + 2) / 2) < 1e-6           # mean of the ranks' counts reached graph L
    q.put((rank, ok_order and ok_vals and ok_early and ok_count, dict(order=order, vals=ok_vals, early=ok_early, count=state["count"])))
    dist.destroy_process_group()


def test_replay_schedule_overlaps_the_early_segment_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_replay_schedule, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = [q.get(timeout=5) for _ in range(world)]
    assert all(ok for _, ok, _ in results), results
