"""One training frame of the whole hot path on synthetic data: multi-view images -> ResNet/FPN ->
unified decoder -> objective -> backward -> optimiser step.  Shared by bench.py, smoke() and tests.

The objective is the reference's: every det / map / motion / ego / plan loss term with Hungarian target
assignment on the device (projects/mmdet3d_plugin/models/criterion.py; reference sparse_onedecoder.py:1094-1579)
plus the dense-depth loss.  (``OBJECTIVE = "surrogate"`` -- mean square of every head output -- remains as a
debugging aid that skips the target assignment.)

The step exists in two forms: ``TrainStep`` launches it eagerly in three parts (forward | losses + decoder
backward | encoder backward) and overlaps the all-reduce of the decoder's gradient segment with the encoder's
backward; ``GraphedTrainStep`` replays it from hipGraphs (the collectives sit between the graphs).
"""
import numpy as np
import torch

from . import synthetic as syn


def build_detector(stage=2, input_hw=(256, 704), plan_queries=None, device="cuda", with_cp=False, backbone_depth=None,
                   encoder_dtype=None):
    import projects.mmdet3d_plugin.models  # noqa: F401  registers the modules
    # MIOpen exhaustive find on the first call of every convolution shape (during the eager warm-up frames):
    # the step is 5.4 ms faster than with the default quick find (63.1 -> 57.7 ms).  Immediate mode
    # (torch.backends.miopen.immediate) is NOT an alternative: without a populated find-db its ranking picked
    # CK weight-gradient solvers of 1.2 ms and 38 ms per call (step 189 ms) on a fresh machine.
    torch.backends.cudnn.benchmark = True
    from hipad_amd.compat import DETECTORS, build_from_cfg
    from projects.configs._hipad_b2d_common import hipad_b2d
    cfg = hipad_b2d(stage=stage, input_shape=(input_hw[1], input_hw[0]))
    model_cfg = cfg["model"]
    model_cfg["img_backbone"]["pretrained"] = None  # no network / checkpoints here: random init
    model_cfg["img_backbone"]["with_cp"] = with_cp
    if backbone_depth is not None:      # BASELINE.json config 5: ResNet101
        model_cfg["img_backbone"]["depth"] = backbone_depth
    if encoder_dtype is not None:       # BASELINE.json config 2: fp32 encoder
        model_cfg["encoder_dtype"] = encoder_dtype
    if plan_queries == 48 and stage == 2:
        # BASELINE.json words the plan set as 6x8 = 48 queries: keep one anchor group of the ten
        od = model_cfg["head"]["onedecoder_head"]
        keep = od["plan_anchor_refer"]
        for key in ("plan_instance_bank", "plan_refine_layer", "plan_decoder"):
            od[key]["anchor_types"] = [keep]
        od["plan_instance_bank"]["anchor_paths"] = {keep: od["plan_instance_bank"]["anchor_paths"][keep]}
    model = build_from_cfg(model_cfg, DETECTORS)
    model.init_weights()
    return model.to(device), cfg


class SyntheticFrames:
    """Endless stream of synthetic frames with consistent ego motion (SURVEY.md section 8d inputs)."""

    MAX_DET, MAX_MAP = 32, 16  # capacity of the padded ground-truth tensors

    def __init__(self, bs=1, input_hw=(256, 704), device="cuda", seed=0):
        self.bs, self.hw, self.device = bs, input_hw, device
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        pm, wh = syn.projection_mats(input_hw, bs=bs)
        self.projection_mat = torch.from_numpy(pm).to(device)
        self.image_wh = torch.from_numpy(wh).to(device)
        self.step = 0
        # a small pool of pre-generated images resident in HBM (the timed region starts with inputs on device)
        self.pool = [torch.randn(bs, 6, 3, input_hw[0], input_hw[1], generator=self.gen).to(device) for _ in range(4)]
        cmd = torch.zeros(bs, 6)
        cmd[:, 3] = 1
        self.cmd = cmd.to(device)
        self.target = (torch.rand(bs, 2, generator=self.gen) * 60 - 30).to(device)
        # ground truth for the loss path: a small pool of pre-generated, padded sets resident in HBM
        from projects.mmdet3d_plugin.models.criterion import pad_ground_truth
        self.gt_pool = []
        for k in range(4):
            raw = syn.ground_truth(bs=bs, seed=seed * 16 + k, input_hw=input_hw)
            raw = {key: ([t.to(device) for t in v] if isinstance(v, list) else v.to(device)) for key, v in raw.items()}
            dense = {key: v for key, v in raw.items() if not key.startswith(("gt_bboxes", "gt_labels", "gt_agent", "gt_map"))}
            dense["gt_padded"] = pad_ground_truth(raw, max_det=self.MAX_DET, max_map=self.MAX_MAP)
            self.gt_pool.append(dense)

    def next(self):
        k = self.step
        T = syn.ego_motion(k)
        Tinv = np.linalg.inv(T)
        data = dict(projection_mat=self.projection_mat, image_wh=self.image_wh,
                    timestamp=torch.full((self.bs,), 0.5 * k, dtype=torch.float64, device=self.device),
                    img_metas=[dict(T_global=T, T_global_inv=Tinv) for _ in range(self.bs)],
                    gt_ego_fut_cmd=self.cmd, target_point=self.target)
        data.update(self.gt_pool[k % len(self.gt_pool)])
        self.step += 1
        return self.pool[k % len(self.pool)], data


def surrogate_objective(model_outs, depths=None):
    det, mp, ego, plan, motion, _ = model_outs
    terms = []
    for out in (det, mp, ego, plan, motion):
        for key in ("classification", "prediction", "quality", "status"):
            for t in out.get(key, []) or []:
                if t is not None:
                    terms.append(t.float().square().mean())
    for d in depths or []:
        terms.append(d.float().clamp(max=60.0).mean() * 1e-3)
    return torch.stack(terms).sum()


class TrainStep:
    """forward + objective + backward + gradient all-reduce + clip + AdamW step for one frame batch
    (optimiser settings of the reference: AdamW lr 2e-4 wd 1e-3, backbone lr x0.5, grad-clip 25,
    projects/configs/hipad_b2d_stage2.py:629-641; data parallel as apis/mmdet_train.py:97-102)."""

    def __init__(self, model, cfg, comm_dtype=None, capturable=False):
        import torch.distributed as dist
        from .dist import broadcast_parameters
        from .optim import FlatAdamW
        self.model = model
        broadcast_parameters(model)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        opt = cfg["optimizer"]
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        bb = [p for n, p in named if n.startswith("img_backbone")]
        # flat layout = [decoder + depth heads | FPN | backbone]: the first segment's gradients are complete when the
        # decoder's backward has run, so its all-reduce can travel while the encoder's backward computes the rest
        early = [p for n, p in named if not n.startswith(("img_backbone", "img_neck"))]
        neck = [p for n, p in named if n.startswith("img_neck")]
        rest = early + neck
        mult = opt["paramwise_cfg"]["custom_keys"]["img_backbone"]["lr_mult"]
        self.max_norm = cfg["optimizer_config"]["grad_clip"]["max_norm"]
        # parameters, gradients and AdamW moments in flat buffers; clip + step = two launches
        # lr_config (linear warm-up + cosine annealing per iteration) is evaluated inside the optimiser kernel
        runner = cfg.get("runner") or {}
        import importlib
        enc = importlib.import_module("projects.mmdet3d_plugin.models.image_encoder")
        convs = [m for m in model.modules() if isinstance(m, enc.Conv2d) and m.weight.requires_grad]
        shadow_convs = bool(convs) and getattr(model, "encoder_dtype", None) == torch.bfloat16 and convs[0].weight.is_cuda
        self.opt = FlatAdamW([(rest, opt["lr"]), (bb, opt["lr"] * mult)], weight_decay=opt["weight_decay"],
                             max_norm=self.max_norm, comm_dtype=comm_dtype, lr_config=cfg.get("lr_config"),
                             max_iters=int(runner.get("max_iters", 0)), bf16_shadow=shadow_convs)
        if shadow_convs:
            # the encoder's convolutions read the bf16 copy of their weight that the AdamW kernel keeps current
            for m in convs:
                m.weight._hipad_bf16 = self.opt.shadow_of(m.weight)
        self.params = self.opt.params
        self.grads = self.opt.grads
        late = neck + bb
        if late:
            self.grads.set_split(late[0])
        self._early_params = early
        self._loosened = False
        # ids of parameters whose gradients OUR kernels write in place are collected during this step's first backward;
        # ids left from an earlier model (tests build many) could be reused by Python for this model's parameters
        from . import functional as _HF
        _HF.INPLACE_PARAMS.clear()
        self._comm_stream = None

    @property
    def grad_norm(self):
        """Total gradient norm before clipping of the last update (device scalar)."""
        return self.opt.grad_norm

    # ---- the step in three parts (each capturable as one hipGraph; the collectives sit BETWEEN them) ----------------
    def part_forward(self, img, data, keep_levels=False):
        """Encoder + decoder forward; with more than one rank also the target assignment and the positive counts the
        losses normalise by (compat.CountExchange "collect": they are all-reduced before part_loss_backward).
        ``keep_levels``: remember the pyramid levels as the cut point of a two-part backward (eager steps only)."""
        from .compat import count_exchange
        det = self.model
        self._data = data
        self._cut = None
        if OBJECTIVE == "surrogate":
            feature_maps, self._depths = det.extract_feat(img, True, data)
            self._outs = det.head(img, feature_maps, data)
            return
        feature_maps, self._depths = det.extract_feat(img, True, data)
        if keep_levels:
            self._cut = getattr(feature_maps[0], "_hipad_levels", None)
        with torch.autocast("cuda", dtype=DECODER_DTYPE, enabled=DECODER_DTYPE != torch.float32):
            self._outs = det.head(img, feature_maps, data)
        if self.world > 1:
            count_exchange.begin("collect")
            try:
                det.head.onedecoder_head.positive_counts(*self._outs, data)
            finally:
                count_exchange.begin("direct")

    def exchange_counts(self):
        from .compat import count_exchange
        if self.world > 1 and OBJECTIVE != "surrogate":
            count_exchange.all_reduce()

    def _objective(self):
        """Sum of the loss terms of the kept forward outputs (positive counts from the exchange when there are ranks)."""
        from .compat import count_exchange
        det, data = self.model, self._data
        if OBJECTIVE == "surrogate":
            return surrogate_objective(self._outs, self._depths)
        if self.world > 1:
            count_exchange.begin("use")
        try:
            losses = det.head.loss(self._outs, data)
        finally:
            count_exchange.begin("direct")
        return _sum_losses(losses, det.depth_branch.loss(self._depths, data["gt_depth"])
                           if (self._depths is not None and "gt_depth" in data) else None)

    def part_loss_backward(self, whole=False):
        """Losses (normalised by the exchanged counts) + the backward down to the pyramid levels: every gradient of
        the first flat segment is complete afterwards.  ``whole=True`` (or no kept cut point) runs the entire backward in
        one call, as the captured steps do."""
        from . import functional as HF
        from .compat import count_exchange
        loss = self._objective()
        self.grads.before_backward()
        levels = [] if whole else [t for t in (self._cut or []) if t.requires_grad]
        self._levels, self._cut = levels, None
        if levels:
            torch.autograd.backward([loss], inputs=self._early_params + levels)
            self.grads.after_backward("early")
        else:  # frozen encoder: everything is in the first part
            loss.backward()
            self.grads.after_backward()
        self._outs = self._depths = None
        return loss

    def part_backward_encoder(self):
        """The rest of the backward: FPN + backbone, from the gradients of the pyramid levels."""
        from . import functional as HF
        levels = self._levels
        if levels:
            torch.autograd.backward(levels, [t.grad for t in levels])
            for t in levels:
                t.grad = None
            self.grads.after_backward("late")
        self._levels = None
        if not self._loosened:  # after the first backward it is known which gradients our kernels write in place
            self.grads.loosen(HF.INPLACE_PARAMS)
            self._loosened = True

    def forward_backward(self, img, data):
        self.part_forward(img, data, keep_levels=True)
        self.exchange_counts()
        loss = self.part_loss_backward()
        self.part_backward_encoder()
        return loss

    # ---- gradient all-reduce overlapped with the encoder's backward --------------------------------------------------
    def reduce_early(self):
        """Start the all-reduce of the first segment on the communication stream (call after part_loss_backward)."""
        if self.world == 1:
            return
        if not self.grads.flat.is_cuda:       # host tensors (gloo tests): no streams, the collective runs in place
            self.grads.all_reduce_mean(segment="early")
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            self.grads.all_reduce_mean(segment="early")

    def reduce_late(self):
        """All-reduce the second segment and join the communication stream (call after part_backward_encoder)."""
        if self.world == 1:
            return
        if not self.grads.flat.is_cuda:
            self.grads.all_reduce_mean(segment="late")
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            self.grads.all_reduce_mean(segment="late")
        torch.cuda.current_stream().wait_stream(self._comm_stream)

    def update(self):
        self.opt.step(zero_grad=True)  # the kernel clears each gradient once it has been applied

    def __call__(self, img, data):
        from . import functional as HF
        self.part_forward(img, data, keep_levels=True)
        self.exchange_counts()
        loss = self.part_loss_backward()  # gradients are zero here: update() clears what it applied
        if img.is_cuda:
            self.grads.check_views("early")       # before the segment's all-reduce starts (it runs on a side stream)
            self.reduce_early()
            self.part_backward_encoder()
            self.grads.check_views("late")
            self.reduce_late()
        else:
            self.part_backward_encoder()
            self.grads.check_views()
            self.grads.all_reduce_mean()
        self.update()
        HF.advance_dropout_clock(img.device)
        return loss


class GraphedTrainStep:
    """The same step replayed from hipGraphs: a frame is ~17 000 small kernels, and launching them
    eagerly costs more host time than the GPU needs to run them.

    Everything the step reads that changes from frame to frame lives in static device buffers filled
    OUTSIDE the graph (images, timestamp, ego-motion transform, GridMask parameters; the dropout clock
    and the instance-bank caches are device state updated in place INSIDE it), so the captured work has
    no host dependency.  The collectives sit between the graphs (see the capture code and ``_replay``): with several
    ranks the positive counts of the losses are all-reduced between the forward graph and the loss + backward graph
    (compat.CountExchange), the flat gradient between that one and the clip + AdamW graph.
    Tracking-id bookkeeping (InstanceBank.get_instance_id: data-dependent shapes, inference only) is
    left out of the captured step.
    """

    debug_hook = None  # tools/diag_stale.py: called with "before_capture" / "after_capture"

    def __init__(self, model, cfg, frames, comm_dtype=None, warm_frames=3):
        import torch.distributed as dist
        from . import runtime_env
        if not runtime_env.graph_replay_is_safe():
            raise RuntimeError("captured training steps refused: %s (see hipad_amd/runtime_env.py)" % runtime_env.why_unsafe())
        self.model, self.frames = model, frames
        self.inner = TrainStep(model, cfg, comm_dtype=comm_dtype, capturable=True)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        import os
        self.split_forward = self.world > 1 or os.environ.get("HIPAD_SPLIT_FORWARD") == "1"
        # split_backward: the backward as TWO graphs cut at the pyramid levels (losses + decoder | FPN + backbone), so that
        # the decoder segment's all-reduce can travel on a side stream while the encoder's backward replays
        self.split_backward = self.split_forward and os.environ.get("HIPAD_SPLIT_BACKWARD", "1") == "1"
        dec = model.head.onedecoder_head
        dec.with_instance_id = False
        if model.use_grid_mask:
            model.grid_mask.external_randomize = True
        dev = frames.device
        bs = frames.bs
        img, data = frames.next()
        self.img = torch.empty_like(img)
        self.ts = torch.zeros(bs, dtype=torch.float64, device=dev)
        self.T = torch.zeros(bs, 4, 4, dtype=torch.float32, device=dev)
        # pinned staging ring: the host may run several frames ahead of the GPU, so a staging buffer is
        # not rewritten until RING further frames have been enqueued behind its asynchronous copy
        self._ring = [(torch.zeros(bs, 4, 4, dtype=torch.float32).pin_memory(),
                       torch.zeros(bs, dtype=torch.float64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        self._ring_i = 0
        self.data = dict(projection_mat=data["projection_mat"], image_wh=data["image_wh"], timestamp=self.ts,
                         T_temp2cur=self.T, img_metas=data["img_metas"], gt_ego_fut_cmd=data["gt_ego_fut_cmd"],
                         target_point=data["target_point"])
        # ground truth of the loss path: static buffers refreshed by _feed (a handful of small device copies)
        self._gt_keys = [k for k in data if k == "gt_padded" or k == "gt_depth" or k.startswith("ego_status")
                         or (k.startswith("gt_ego_") and k != "gt_ego_fut_cmd")]
        for k in self._gt_keys:
            self.data[k] = _clone_tree(data[k])
        self._prev_T = None
        self._feed(img, data)
        # cold frames (no temporal cache yet) run eagerly; they also size every workspace / cache
        if model.use_grid_mask:
            model.grid_mask.external_randomize = False
            model.grid_mask.train()
        for i in range(warm_frames):
            if i:
                self._feed(*frames.next())
            self._eager_body()
        if model.use_grid_mask:
            model.grid_mask.external_randomize = True
        torch.cuda.synchronize()
        self._feed(*frames.next())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # one more eager pass on the capture-side stream (library workspaces)
            self._eager_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._feed(*frames.next())
        if self.debug_hook is not None:
            self.debug_hook("before_capture")
        # one rank:      graph 1 = forward + losses + backward;                                   graph 2 = clip + AdamW
        # several ranks: graph 1 = forward + target assignment + positive counts;  [counts all-reduce];
        #                graph 2 = losses + backward;  [gradient all-reduce];                      graph 3 = clip + AdamW
        # The two-part backward (decoder, then encoder, so that the decoder segment's all-reduce could travel meanwhile)
        # is used by the eagerly launched step (TrainStep.__call__) only: captured, the encoder's backward as a separate
        # autograd.backward call made ROCm 7.2 die in capture_end in every arrangement tried (tools/diag_graph_split.py,
        # DESIGN.md section 4), and there is no 2-GPU box in this build loop to debug an in-graph overlap on.
        self.graph_f = torch.cuda.CUDAGraph()
        self.graph_l = self.graph_e = None
        if self.split_forward:
            with torch.cuda.graph(self.graph_f):
                self.inner.part_forward(self.img, self.data, keep_levels=self.split_backward)
            self.graph_l = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_l, pool=self.graph_f.pool()):
                self.loss = self.inner.part_loss_backward(whole=not self.split_backward)
                if not self.split_backward:
                    self.inner.part_backward_encoder()
            if self.split_backward:
                self.graph_e = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_e, pool=self.graph_f.pool()):
                    self.inner.part_backward_encoder()
        else:
            with torch.cuda.graph(self.graph_f):
                self.inner.part_forward(self.img, self.data)
                self.loss = self.inner.part_loss_backward(whole=True)
                self.inner.part_backward_encoder()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_f.pool()):
            self._update()
        if self.debug_hook is not None:
            self.debug_hook("after_capture")
        # the capture itself did not execute the work: replay once so the banks hold this frame's state
        self._replay()

    def _replay(self):
        self.graph_f.replay()
        if self.graph_l is not None:
            self.inner.exchange_counts()
            self.graph_l.replay()
        if self.graph_e is not None:
            # [F | counts | L: losses + decoder backward | early all-reduce (side stream) || E: encoder backward | late | B]
            self.inner.reduce_early()
            self.graph_e.replay()
            self.inner.reduce_late()
        elif self.world > 1:
            self.inner.grads.all_reduce_mean()
        self.graph_b.replay()

    def _feed(self, img, data):
        import numpy as np
        self.img.copy_(img, non_blocking=True)
        Ts = [m["T_global"] for m in data["img_metas"]]
        Tinv = [m["T_global_inv"] for m in data["img_metas"]]
        prev = self._prev_T if self._prev_T is not None else Ts
        T_host, ts_host, done = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        done.synchronize()  # the copy that last used this slot has finished (8 frames ago: never blocks in practice)
        T_host.copy_(torch.from_numpy(np.stack([ti @ tp for ti, tp in zip(Tinv, prev)]).astype(np.float32)))
        self._prev_T = Ts
        ts_host.copy_(self._ts_from(data))
        self.T.copy_(T_host, non_blocking=True)
        self.ts.copy_(ts_host, non_blocking=True)
        done.record()
        for k in self._gt_keys:
            _copy_tree(self.data[k], data[k])
        for k in ("projection_mat", "image_wh"):    # per-frame camera geometry (image augmentation): into the static buffers
            if data[k].data_ptr() != self.data[k].data_ptr():
                self.data[k].copy_(data[k], non_blocking=True)
        if self.model.use_grid_mask and getattr(self.model.grid_mask, "_last_h", None) is not None:
            self.model.grid_mask.randomize(self.img.device)

    def _ts_from(self, data):
        host = data.get("timestamp_host")           # a frame source with its own clock (hipad_amd.dataflow.SequenceFrames)
        if host is not None:
            return host
        return torch.full((self.frames.bs,), 0.5 * (self.frames.step - 1), dtype=torch.float64)

    def _fwd_bwd(self):
        # gradients start at zero: FlatGrads allocates them so and every update() clears them again
        self.inner.part_forward(self.img, self.data, keep_levels=self.split_backward)
        self.inner.exchange_counts()
        loss = self.inner.part_loss_backward(whole=not self.split_backward)
        self.inner.part_backward_encoder()
        return loss

    def _update(self):
        from . import functional as HF
        self.inner.update()
        HF.advance_dropout_clock(self.img.device)

    def _eager_body(self):
        self._fwd_bwd()
        self.inner.grads.check_views()
        self.inner.grads.all_reduce_mean()
        self._update()

    def __call__(self):
        self._feed(*self.frames.next())
        self._replay()
        return self.loss


class GraphedInference:
    """Closed-loop / test-time step (reference SparseDetector.simple_test, models/sparse_detector.py:153-167) with
    the network part replayed from ONE hipGraph: encoder + decoder in eval mode write into static output tensors;
    the data-dependent tail -- track-id bookkeeping (InstanceBank.get_instance_id) and the result decoders with their
    device->host copies -- runs eagerly on those outputs.  The closed-loop agent calls the model at 20 Hz with
    batch 1 (bench2drive/leaderboard/team_code/hipad_b2d_agent.py:456-615): this is that call."""

    def __init__(self, model, frames, warm_frames=3):
        from . import runtime_env
        if not runtime_env.graph_replay_is_safe():
            raise RuntimeError("captured steps refused: %s (see hipad_amd/runtime_env.py)" % runtime_env.why_unsafe())
        self.model, self.frames = model.eval(), frames
        self.dec = model.head.onedecoder_head
        dev, bs = frames.device, frames.bs
        img, data = frames.next()
        self.img = torch.empty_like(img)
        self.ts = torch.zeros(bs, dtype=torch.float64, device=dev)
        self.T = torch.zeros(bs, 4, 4, dtype=torch.float32, device=dev)
        self.data = dict(projection_mat=data["projection_mat"], image_wh=data["image_wh"], timestamp=self.ts,
                         T_temp2cur=self.T, img_metas=data["img_metas"], gt_ego_fut_cmd=data["gt_ego_fut_cmd"],
                         target_point=data["target_point"])
        self._prev_T = None
        self._feed(img, data)
        self.dec.with_instance_id = False  # ids are assigned outside the graph (data-dependent shapes)
        with torch.no_grad():
            for i in range(warm_frames):  # cold frames run eagerly and size the temporal caches
                if i:
                    self._feed(*frames.next())
                self._network()
            torch.cuda.synchronize()
            self._feed(*frames.next())
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._network()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._feed(*frames.next())
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outs = self._network()
            self.graph.replay()

    def _network(self):
        feature_maps = self.model.extract_feat(self.img, False, self.data)
        return self.model.head(self.img, feature_maps, self.data)

    def _feed(self, img, data):
        import numpy as np
        self.img.copy_(img, non_blocking=True)
        Ts = [m["T_global"] for m in data["img_metas"]]
        Tinv = [m["T_global_inv"] for m in data["img_metas"]]
        prev = self._prev_T if self._prev_T is not None else Ts
        T = torch.from_numpy(np.stack([ti @ tp for ti, tp in zip(Tinv, prev)]).astype(np.float32))
        self._prev_T = Ts
        self.T.copy_(T)  # pageable -> device: synchronous, so the host buffer may go out of scope
        self.ts.copy_(torch.full((self.frames.bs,), 0.5 * (self.frames.step - 1), dtype=torch.float64))

    def __call__(self):
        """One frame: feed, replay, assign track ids, decode results -> list of {"img_bbox": result}."""
        self._feed(*self.frames.next())
        self.graph.replay()
        with torch.no_grad():
            det = self.outs[0]
            det["instance_id"] = self.dec.det_instance_bank_list[0].get_instance_id(
                det["classification"][-1], det["prediction"][-1], self.dec.det_decoder.score_threshold)
            results = self.model.head.post_process(self.outs, self.data)
        return [dict(img_bbox=r) for r in results]


def _clone_tree(obj):
    if isinstance(obj, torch.Tensor):
        return obj.clone()
    if isinstance(obj, dict):
        return {k: _clone_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_clone_tree(v) for v in obj]
    return obj


def _copy_tree(dst, src):
    if isinstance(dst, torch.Tensor):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_tree(dst[k], src[k])
    elif isinstance(dst, list):
        for d, s_ in zip(dst, src):
            _copy_tree(d, s_)


DECODER_DTYPE = torch.float32  # no autocast in the decoder: its Linear layers are the bf16-operand MFMA kernel


def frame_losses(det, img, data):
    """The reference's forward_train (models/sparse_detector.py:141-151): encoder, decoder, the decoder's losses
    (target assignment included) and the dense-depth loss -> dict of scalar losses."""
    feature_maps, depths = det.extract_feat(img, True, data)
    with torch.autocast("cuda", dtype=DECODER_DTYPE, enabled=DECODER_DTYPE != torch.float32):
        outs = det.head(img, feature_maps, data)
    losses = det.head.loss(outs, data)
    if depths is not None and "gt_depth" in data:
        losses["loss_dense_depth"] = det.depth_branch.loss(depths, data["gt_depth"])   # (LossDict.total stays the decoder's)
    return losses


def _sum_losses(losses, depth=None):
    """Sum of the loss terms; a producer that already holds the sum of its terms (criterion.LossDict.total) saves the
    chain of scalar additions."""
    total = getattr(losses, "total", None)
    if total is None:
        for v in losses.values():
            total = v if total is None else total + v
    return total if depth is None else total + depth


def _frame_loss(det, img, data):
    if OBJECTIVE == "surrogate":
        feature_maps, depths = det.extract_feat(img, True, data)
        outs = det.head(img, feature_maps, data)
        return surrogate_objective(outs, depths)
    losses = frame_losses(det, img, data)
    depth = losses.pop("loss_dense_depth", None)
    return _sum_losses(losses, depth)


OBJECTIVE = "losses"  # "surrogate": mean square of every head output (debugging aid; skips target assignment)
