"""The captured training step (hipGraph replay) computes what the eagerly launched step computes.

Regression test for the ROCm 7.2 graph packet-capture defect (hip-ad_amd/runtime_env.py): with it, replays returned
gradients of norm 1e27..1e37 that clipping turned into a "working" step.  All stochastic parts (dropout, sampling-weight
keep mask, GridMask) and the weight update (lr 0) are switched off so the two runs see the same arithmetic; what remains is bf16 / atomic-order
noise and MIOpen's per-process solver choice."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def quiet(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "attention_dropout"):
            m.attention_dropout = 0.0
        if hasattr(m, "attn_drop") and isinstance(getattr(m, "attn_drop"), float):
            m.attn_drop = 0.0
    model.use_grid_mask = False


def run(mode, steps):
    from hipad_amd.frame import GraphedTrainStep, SyntheticFrames, TrainStep, build_detector
    torch.manual_seed(5)
    model, cfg = build_detector(stage=2, plan_queries=480)
    model.train()
    quiet(model)
    # frozen weights: with updates on, gradient-sign flips of near-zero gradients make two EAGER runs drift 10 % apart
    # within five frames (tools/graph_vs_eager.py); what is left evolving is the temporal instance bank
    cfg["optimizer"] = dict(cfg["optimizer"], lr=0.0, weight_decay=0.0)
    frames = SyntheticFrames(seed=3)
    trace = []
    if mode == "eager":
        step = TrainStep(model, cfg)
        for _ in range(steps):
            loss = step(*frames.next())
            trace.append((float(loss), float(step.grad_norm)))
    else:
        step = GraphedTrainStep(model, cfg, frames)     # runs frames 0..4 itself (3 eager, 1 side-stream eager, 1 replay)
        for _ in range(steps - 5):
            loss = step()
            trace.append((float(loss), float(step.inner.grad_norm)))
    return trace


def test_replayed_step_tracks_eager_step():
    warnings.filterwarnings("ignore")
    eager = run("eager", 8)[5:]      # frames 5, 6, 7
    graph = run("graph", 8)          # the same three frames, replayed
    assert len(eager) == len(graph) == 3
    for (le, ge), (lg, gg) in zip(eager, graph):
        assert all(map(lambda v: v == v and abs(v) < 1e6, (le, ge, lg, gg))), (eager, graph)
        assert abs(le - lg) <= 0.03 * abs(le), (eager, graph)      # two eager runs: within 1.1 %
        assert abs(ge - gg) <= 0.30 * abs(ge), (eager, graph)      # two eager runs: within 14 %
