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


DECODER_TAGS = ("topk", "motion_class")


class Choices:
    """Recorder / replayer of the decoder's discrete choices (hipad_amd.compat.discrete_choice) frame by frame.  Replayed
    choices are served from STATIC device buffers filled before the frame starts, so a captured step reads them too."""

    def __init__(self, recorded=None, tags=DECODER_TAGS):
        self.frames = [] if recorded is None else recorded
        self.replay = recorded is not None
        self.tags = tags                     # None: every tag (the loss path's assignment / order / mode / gate choices too)
        self.bufs, self.i, self.k = {}, 0, -1

    def begin_frame(self, k):
        self.k, self.i = k, 0
        if not self.replay:
            while len(self.frames) <= k:
                self.frames.append([])
            self.frames[k] = []
            return
        for i, (tag, t) in enumerate(self.frames[k]):
            key = (i, tag, tuple(t.shape))
            if key not in self.bufs:
                self.bufs[key] = torch.empty_like(t)
            self.bufs[key].copy_(t)

    def __call__(self, tag, choice):
        if self.tags is not None and tag not in self.tags:
            return choice
        if not self.replay:
            self.frames[self.k].append((tag, choice.clone()))
            return choice
        rtag, t = self.frames[self.k][self.i]
        assert rtag == tag and t.shape == choice.shape, (self.k, self.i, rtag, tag)
        buf = self.bufs[(self.i, tag, tuple(t.shape))]
        self.i += 1
        return buf


def pinned_run(mode, steps, recorded, scope="all", first=5, encoder_dtype=None):
    """Frames 5 .. steps-1 of the eager or the replayed step with the decoder's discrete choices recorded (``recorded`` None)
    or replayed: per frame the loss, the pre-clip gradient norm and a copy of the flat gradient (+ the segment split)."""
    from hipad_amd import compat as CR, functional as HF
    from hipad_amd.frame import GraphedTrainStep, SyntheticFrames, TrainStep, build_detector
    torch.manual_seed(5)
    model, cfg = build_detector(stage=2, plan_queries=480, encoder_dtype=encoder_dtype)
    model.train()
    quiet(model)
    model.head.onedecoder_head.with_instance_id = False    # as in the captured step (track ids: a top-k of their own)
    cfg["optimizer"] = dict(cfg["optimizer"], lr=0.0, weight_decay=0.0)
    frames = SyntheticFrames(seed=3)
    # scope "all": every discrete choice incl. the target assignment is pinned (the objective then runs in its torch-op
    # formulation, which hosts those hooks); "decoder": only the decoder's own choices, the fused objective stays on
    choices = Choices(recorded, None if scope == "all" else DECODER_TAGS)
    plain_next = frames.next

    def next_frame():
        choices.begin_frame(frames.step)
        return plain_next()
    frames.next = next_frame
    identity, scope_before = CR.discrete_choice[0], CR.discrete_scope[0]
    CR.discrete_choice[0], CR.discrete_scope[0] = choices, scope
    names = {id(p): n for n, p in model.named_parameters()}
    out = []
    try:
        if mode == "eager":
            step = TrainStep(model, cfg)
            for k in range(steps):
                img, data = frames.next()
                step.part_forward(img, data, keep_levels=True)
                step.exchange_counts()
                loss = step.part_loss_backward()
                step.part_backward_encoder()
                step.grads.check_views()
                flat = step.grads.flat.clone()
                step.update()
                HF.advance_dropout_clock(img.device)
                if k >= first:
                    out.append(dict(loss=float(loss), norm=float(step.grad_norm), flat=flat, split=step.grads.split))
        else:
            g = GraphedTrainStep(model, cfg, frames)     # frames 0..4 (eager warm-up, capture, first replay)
            for k in range(5, steps):
                g._feed(*frames.next())
                g.graph_f.replay()
                if g.graph_l is not None:          # the multi-rank schedule (HIPAD_SPLIT_FORWARD=1 forces it on one rank)
                    g.inner.exchange_counts()
                    g.graph_l.replay()
                if g.graph_e is not None:
                    g.inner.reduce_early()
                    g.graph_e.replay()
                    g.inner.reduce_late()
                flat = g.inner.grads.flat.clone()
                g.graph_b.replay()
                out.append(dict(loss=float(g.loss), norm=float(g.inner.grad_norm), flat=flat, split=g.inner.grads.split))
    finally:
        CR.discrete_choice[0], CR.discrete_scope[0] = identity, scope_before
    grads = step.grads if mode == "eager" else g.inner.grads
    layout = [(names[id(p)], off, p.numel()) for p, off in zip(grads.params, grads.offsets)]
    for o in out:
        o["layout"] = layout
    return out, choices.frames


def segment_distances(a, b):
    """Relative L2 distance of the two flat gradients: (decoder + depth heads segment, FPN + backbone segment)."""
    s = a["split"]
    return tuple(float((a["flat"][lo:hi] - b["flat"][lo:hi]).norm() / a["flat"][lo:hi].norm().clamp_min(1e-30))
                 for lo, hi in ((0, s), (s, a["flat"].numel())))


def test_replayed_step_tracks_eager_step():
    """The flat gradient of the replayed (hipGraph) step against the eagerly launched step, frame by frame (frames 5..9:
    temporal caches warm, graphs replayed), decoder segment and encoder segment separately, with the decoder's discrete
    choices (temporal top-k, motion-mode class) of the eager run replayed so that both runs follow the same branches.

    What made this test impossible to hold tightly in round 2 (two EAGER runs 30-80 % apart in the gradient, loss equal to
    3 digits) was not the chaos of a random-init net alone: the BatchNorm statistics were summed with fp32 atomics, the
    sums moved in their last bit from run to run, bf16 activations flipped roundings, the pyramid ended 4e-3 apart
    (tools/diag_forward_determinism.py) and the decoder's gradient is ~100x that sensitive (measured on CPU in plain
    fp32 torch ops: tools/diag_grad_conditioning_cpu.py, profiles/r03_gradient_conditioning_cpu.txt).  The statistics are
    accumulated in 64-bit fixed point now (order-independent): the forward is bitwise reproducible, what is left between
    two runs is the summation order of the BACKWARD's float atomics -- measured 0.8-1.5e-2 (decoder) / 1.0-1.7e-2 (encoder)
    between two eager runs as well as between eager and replayed (profiles/r03_eager_vs_replay_pinned.txt).  A replay defect
    of the ROCm 7.2 kind (garbage of norm 1e27 in some gradient segment) or a missing / doubled segment is orders of
    magnitude above the bound; the bound is 2-3x the measured noise floor."""
    warnings.filterwarnings("ignore")
    from hipad_amd import functional as HF
    HF.LIBRARY_CALLS.clear()
    pinned_run("eager", 2, None, "decoder", 0)        # throw-away: MIOpen's find phase runs other solvers on first calls
    eager, choices = pinned_run("eager", 10, None, "decoder")
    again, _ = pinned_run("eager", 10, choices, "decoder")
    graph, _ = pinned_run("graph", 10, choices, "decoder")
    # the multi-rank schedule on one rank: forward | losses + decoder backward | encoder backward as three graphs (the
    # decoder segment's all-reduce travels beside the third one when there are ranks)
    import os
    os.environ["HIPAD_SPLIT_FORWARD"] = "1"
    try:
        split, _ = pinned_run("graph", 10, choices, "decoder")
    finally:
        del os.environ["HIPAD_SPLIT_FORWARD"]
    # no Linear / LayerNorm of the decoder took a torch / library path on the way (silent fallbacks)
    assert not HF.LIBRARY_CALLS, dict(HF.LIBRARY_CALLS)
    assert len(eager) == len(again) == len(graph) == 5
    floor = [segment_distances(a, b) for a, b in zip(eager, again)]
    replay = [segment_distances(a, b) for a, b in zip(eager, graph)]
    print("eager vs eager (noise floor):", floor, "\neager vs replayed:", replay)
    replay_split = [segment_distances(a, b) for a, b in zip(eager, split)]
    print("eager vs replayed (three-graph schedule):", replay_split)
    for k, (e, g) in enumerate(zip(eager, split)):
        assert abs(e["loss"] - g["loss"]) <= 1e-4 * abs(e["loss"]), (k, e["loss"], g["loss"])
        # measured: like the two-graph schedule on four frames of five, 4.9e-2 / 1.9e-2 on the frame whose gradient norm
        # is largest (its eager-vs-eager floor is the largest as well, 1.5e-2): the two-part backward sums in another order
        assert replay_split[k][0] <= 8e-2 and replay_split[k][1] <= 4e-2, (k, "split schedule", replay_split, floor)
    for k, (e, g) in enumerate(zip(eager, graph)):
        assert all(map(lambda v: v == v and abs(v) < 1e6, (e["loss"], e["norm"], g["loss"], g["norm"]))), (k, e["loss"], g["loss"])
        assert abs(e["loss"] - g["loss"]) <= 1e-4 * abs(e["loss"]), (k, e["loss"], g["loss"])
        assert abs(e["norm"] - g["norm"]) <= 0.02 * e["norm"], (k, e["norm"], g["norm"])
        assert replay[k][0] <= 3e-2, (k, "decoder segment", replay, floor)
        assert replay[k][1] <= 4e-2, (k, "encoder segment", replay, floor)
        assert floor[k][0] <= 3e-2 and floor[k][1] <= 4e-2, (k, "two eager runs", floor)


def test_two_part_backward_equals_one_backward():
    """The eager step's backward runs in two parts (losses + decoder down to the pyramid levels, then the encoder) so that
    the decoder's gradient segment can be all-reduced while the encoder's backward runs.  On ONE forward (graph retained)
    the flat gradient of the two-part path equals the one of a single loss.backward() up to the order of float atomics
    (two separate forwards of a random-init model differ by ~1 %: near-ties in the target assignment)."""
    from hipad_amd.frame import SyntheticFrames, TrainStep, build_detector
    warnings.filterwarnings("ignore")
    torch.manual_seed(11)
    model, cfg = build_detector(stage=2, plan_queries=480)
    model.train()
    quiet(model)
    step = TrainStep(model, cfg)
    img, data = SyntheticFrames(seed=2).next()
    step.part_forward(img, data, keep_levels=True)
    levels = [t for t in step._cut if t.requires_grad]
    assert len(levels) == 4
    loss = step._objective()
    g = step.grads
    g.before_backward()
    loss.backward(retain_graph=True)
    g.after_backward()
    g.check_views()
    one = g.flat.clone()
    g.flat.zero_()
    g.before_backward()
    torch.autograd.backward([loss], inputs=step._early_params + levels, retain_graph=True)
    g.after_backward("early")
    late_before = g.flat[g.split:].clone()
    torch.autograd.backward(levels, [t.grad for t in levels])
    g.after_backward("late")
    g.check_views()
    two = g.flat
    assert 0 < g.split < one.numel()
    assert float(late_before.abs().max()) == 0.0           # nothing of the encoder's segment exists after part one
    # decoder segment: same arithmetic, only the order of float atomics differs.  Encoder segment: the gradient handed
    # over at the cut point is a bf16 tensor (the encoder runs under bf16 autocast) summed from its consumers in a
    # different order -- bf16 rounding (eps 4e-3) of the hand-off, measured 3.8e-3
    for (lo, hi), tol in (((0, g.split), 1e-3), ((g.split, one.numel()), 1e-2)):
        rel = float((two[lo:hi] - one[lo:hi]).norm() / one[lo:hi].norm())
        assert rel < tol, (lo, hi, rel)
