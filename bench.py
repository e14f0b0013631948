#!/usr/bin/env python
"""bench.py -- frames/s of the HiP-AD hot path on MI355X (contract: see the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one synthetic frame (6 cameras, 704x256).
Workloads
  stage2_full  (default) one TRAINING step of the whole hipad_b2d_stage2 model on one frame per GPU:
               ResNet50 + FPN (bf16, channels-last) -> flat pyramid -> unified decoder (det 900 + map 100 +
               plan 480 + ego 1 queries, 6 layers, motion head; hand-written aggregation / projection /
               softmax-weight / attention kernels, bf16 GEMMs) -> the reference's losses with the Hungarian
               target assignment on the device (criterion.py, hipad_linear_assignment) -> backward -> gradient
               all-reduce (RCCL) -> clip -> AdamW.  --plan-queries 48 gives BASELINE.json's 6x8 wording.
  stage2_full_frames   the same training step fed through the data pipeline: uint8 1600x900 camera
               frames resident in HBM -> GroupInBatchSampler -> per-sequence augmentation draw -> device image pipeline
               (hipad_amd.imgpipe) -> GridMask -> step (hipad_amd.dataflow.SequenceFrames).
  daf_stage2   the aggregation path of one stage-2 frame: for each of the 6 decoder layers the
               four deformable-aggregation calls (det 900x13, map 100x300, plan 480x90, ego 1x13
               key points; 6 cams x 4 levels x 8 groups; C=256, bf16 pyramid rows, fp32 arithmetic) forward AND backward on the
               89 760-position pyramid.  This is the hand-written-kernel part of the frame; the
               rest of the model is not in this number (config.workload says so).
Inputs are resident in HBM before the timed region.  `roofline` is for the dominant kernel, timed
live with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (oracle/) on a
bounded sample of the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import hipad_amd  # noqa: E402,F401  sets the HIP runtime flags graph replay needs -- before torch is imported

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stage2_full",
                    choices=("stage2_full", "stage2_full_frames", "daf_stage2", "stage2_infer", "stage2_r101_1600",
                             "stage1_fp32"))
    ap.add_argument("--plan-queries", type=int, default=480, choices=(48, 480))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch the step kernel by kernel instead of replaying hipGraphs")
    ap.add_argument("--bs", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--comm-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="wire dtype of the gradient all-reduce at N > 1 (fp32 = the reference's DDP behaviour)")
    return ap.parse_args()


def dist_setup(n):
    from hipad_amd import dist as D
    # RCCL ("nccl") over xGMI.  HIPAD_DIST_BACKEND=gloo + HIPAD_SHARE_GPU=1 rehearse the N > 1 control flow with
    # several ranks on ONE card (RCCL refuses two ranks per device); never the measured configuration.
    rank, world, local = D.init_from_env(os.environ.get("HIPAD_DIST_BACKEND", "nccl"))
    if os.environ.get("HIPAD_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    return rank, world, local


# ------------------------------------------------------------------------------------------
class DafStage2:
    """The deformable-aggregation calls of one stage-2 frame (forward + backward)."""

    LAYERS = 6

    def __init__(self, device, seed, plan_queries=480, hw=(256, 704)):
        from hipad_amd import lib, synthetic as syn
        self.lib = lib
        lib.load()
        g = torch.Generator().manual_seed(seed)
        ss, st, F = syn.pyramid_tables(hw)
        self.F = F
        self.ss = torch.from_numpy(ss).to(device)
        self.st = torch.from_numpy(st).to(device)
        # bf16 rows, as the training step hands them over (the encoder's output dtype; hipad_daf_*_bf16); grad_feat is fp32
        self.feat = torch.randn(1, F, 256, generator=g).to(device).to(torch.bfloat16)
        self.gfeat = torch.zeros(1, F, 256, dtype=torch.float32, device=device)
        pm, wh = syn.projection_mats(hw)
        names = ["det", "map", "plan" if plan_queries == 480 else "plan48", "ego"]
        self.calls = []
        self.host = {}
        for i, n in enumerate(names):
            kp = syn.synthetic_key_points(n, seed=seed + i)
            loc = syn.project(kp, pm, wh)  # (1,A,P,6,2)
            A, P = loc.shape[1:3]
            w = torch.softmax(torch.randn(1, A, P * 6 * 4, 8, generator=g), 2).reshape(1, A, P, 6, 4, 8).contiguous()
            gout = torch.randn(1, A, 256, generator=g)
            loc_t = torch.from_numpy(loc).contiguous()
            self.host[n] = (loc_t, w, gout)
            d = dict(name=n, A=A, P=P, loc=loc_t.to(device), w=w.to(device), gout=gout.to(device))
            d["out"] = torch.empty(1, A, 256, device=device)
            d["gloc"] = torch.empty_like(d["loc"])
            d["gw"] = torch.empty_like(d["w"])
            v = (loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1)
            d["kept_pairs"] = int(v.sum())
            d["rows_touched"] = self.unique_rows(d["loc"])
            self.calls.append(d)
        self.rows_touched_frame = self.unique_rows([d["loc"] for d in self.calls])

    def unique_rows(self, loc):
        """Distinct pyramid rows the in-bounds bilinear corners of ``loc`` (one location tensor, or a list of them:
        the union) touch (the kernels' own index work, hipad_daf_taps): the compulsory pyramid traffic of a call is
        rows x 512 B (bf16 rows)."""
        rows = []
        for one in (loc if isinstance(loc, (list, tuple)) else [loc]):
            valid, taps = self.lib.daf_taps(self.ss, self.st, one, self.F)
            h_low, w_low, mask, base = taps[..., 0].long(), taps[..., 1].long(), taps[..., 2], taps[..., 3].long()
            W = self.ss[:, :, 1].long()[None, None, None]                       # (1,1,1,cams,L)
            ok = valid.bool()[..., None]
            for bit, (dh, dw) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                sel = ok & ((mask >> bit) & 1).bool()
                rows.append((base + (h_low + dh) * W + (w_low + dw))[sel])
        return int(torch.unique(torch.cat(rows)).numel())

    # -- algorithmic bytes -------------------------------------------
    def alg_bytes(self, d, kind):
        """weights + locations + out/grad_out + compulsory pyramid traffic (every pyramid row an
        in-bounds corner touches, once per launch -- counted from the kernels' own index work, not the looser
        min(F*C, taps*C) cap), plus what the kernel writes."""
        A, P = d["A"], d["P"]
        wbytes = 4 * A * P * 6 * 4 * 8
        lbytes = 4 * A * P * 6 * 2
        obytes = 4 * A * 256
        fbytes = 2 * 256 * d["rows_touched"]       # compulsory pyramid traffic: every touched bf16 row (512 B) once
        if kind == "fwd":
            return wbytes + lbytes + obytes + fbytes
        if kind == "bwd_lw":   # reads w, loc, grad_out, feat; writes grad_w, grad_loc
            return (wbytes + lbytes + obytes + fbytes) + (wbytes + lbytes)
        return wbytes + lbytes + obytes + 2 * (4 * 256 * d["rows_touched"])  # bwd_feat_single_call: read-modify-write of fp32 grad_feat rows

    def fwd(self, d):
        self.lib.daf_forward(self.feat, self.ss, self.st, d["loc"], d["w"], out=d["out"])

    def bwd(self, d):  # one call's backward the reference's way: all three gradients by one call of the op
        self.lib.daf_backward(self.feat, self.ss, self.st, d["loc"], d["w"], d["gout"], self.gfeat, d["gloc"], d["gw"],
                              overwrite_loc_w=True)

    def bwd_lw(self, d):  # grad_loc + grad_weights kernel alone (one launch)
        self.lib.daf_backward(self.feat, self.ss, self.st, d["loc"], d["w"], d["gout"], None, d["gloc"], d["gw"],
                              overwrite_loc_w=True)

    def bwd_feat(self, d):  # grad_feat pipeline of ONE call (fill, count, alloc, place, accumulate)
        self.lib.daf_backward(self.feat, self.ss, self.st, d["loc"], d["w"], d["gout"], self.gfeat, None, None)

    def bwd_feat_frame(self):
        """The feature gradient of the frame's 24 calls the way the training step computes it: ONE counting sort + ONE
        accumulation pass over the taps of all of them (hipad_daf_backward_feat_multi)."""
        calls = [(d["loc"], d["w"], d["gout"]) for _ in range(self.LAYERS) for d in self.calls]
        self.lib.daf_backward_feat_multi(calls, self.gfeat, self.ss, self.st)

    def frame_feat_alg_bytes(self):
        """Algorithmic bytes of bwd_feat_frame: every call's weights, locations and grad_out read once, and a
        read-modify-write of every fp32 grad_feat row (1 KiB) that ANY of the calls touches -- once, not once per call."""
        total = 0
        for d in self.calls:
            total += self.LAYERS * (4 * d["A"] * d["P"] * 6 * 4 * 8 + 4 * d["A"] * d["P"] * 6 * 2 + 4 * d["A"] * 256)
        return total + 2 * 4 * 256 * self.rows_touched_frame

    def step(self):
        self.gfeat.zero_()  # one shared feature-gradient buffer per frame
        for _ in range(self.LAYERS):
            for d in self.calls:
                self.fwd(d)
        for _ in range(self.LAYERS):
            for d in reversed(self.calls):
                self.bwd_lw(d)
        self.bwd_feat_frame()

    def kernel_times(self, reps=20):
        """Average launch duration (ms) of every (call, direction), HIP events on the launch stream; plus the frame's
        merged feature-gradient pass (key ("frame", "bwd_feat"))."""
        res = {}

        def timed(fn, *a):
            fn(*a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(*a)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        for d in self.calls:
            for tag, fn in (("fwd", self.fwd), ("bwd_lw", self.bwd_lw), ("bwd_feat_single_call", self.bwd_feat)):
                res[(d["name"], tag)] = timed(fn, d)
        res[("frame", "bwd_feat")] = timed(self.bwd_feat_frame)
        return res

    def cpu_baseline(self, seconds):
        """CPU oracle (scalar C restatement of the CUDA kernels, 1 core): forward+backward of the four
        calls of ONE decoder layer, repeated until `seconds` have passed; a frame is 6 such layers."""
        from oracle import daf as O
        feat = self.feat.float().cpu().numpy()
        ss, st = self.ss.cpu().numpy(), self.st.cpu().numpy()
        spent, reps = 0.0, 0
        while spent < seconds:
            for d in self.calls:
                loc, w, gout = (x.numpy() for x in self.host[d["name"]])
                t = time.perf_counter()
                O.daf_forward(feat, ss, st, loc, w)
                O.daf_backward(feat, ss, st, loc, w, gout)
                spent += time.perf_counter() - t
            reps += 1
        sec_per_frame = spent / reps * self.LAYERS
        return dict(value=round(1.0 / sec_per_frame, 4), unit="frames/s", cores=1, kind="port",
                    sample=f"oracle/daf_oracle.c fwd+bwd of all four calls of one decoder layer x {reps} "
                           f"repetitions ({spent:.1f} s); frame = 6 layers")


class Stage2Full:
    """One training step of the whole model per frame (see module docstring)."""

    def __init__(self, device, seed, plan_queries=480, bs=1, eager=False, hw=(256, 704), stage=2, comm_dtype=None,
                 pipeline_frames=False, **build):
        import warnings
        warnings.filterwarnings("ignore", category=DeprecationWarning)
        from hipad_amd.frame import GraphedTrainStep, SyntheticFrames, TrainStep, build_detector
        torch.manual_seed(1234)  # identical initial weights on every rank (then broadcast anyway)
        self.model, self.cfg = build_detector(stage=stage, input_hw=hw, plan_queries=plan_queries, device=device, **build)
        self.model.train()
        if pipeline_frames:
            # stored uint8 camera frames (HBM resident) -> sequence-grouped sampler -> per-sequence image augmentation ->
            # device image pipeline (resize / crop / flip / rotate / normalise, two launches per sample) -> the step
            from hipad_amd.dataflow import SequenceFrames
            self.frames = SequenceFrames(bs=bs, input_hw=hw, device=device, seed=seed)
        else:
            self.frames = SyntheticFrames(bs=bs, input_hw=hw, device=device, seed=seed)
        self.hw, self.stage = hw, stage
        self.bs, self.plan_queries, self.eager = bs, plan_queries, eager
        if eager:
            self.train_step = TrainStep(self.model, self.cfg, comm_dtype=comm_dtype)
            self.graphed = None
        else:
            self.graphed = GraphedTrainStep(self.model, self.cfg, self.frames, comm_dtype=comm_dtype)  # warm-up + capture
            self.train_step = self.graphed.inner
        self.daf = DafStage2(device, seed, plan_queries, hw=hw)  # op-level harness for the roofline / cpu legs

    def step(self):
        if self.graphed is not None:
            self.last_loss = self.graphed()
        else:
            self.last_loss = self.train_step(*self.frames.next())

    def sanity(self):
        """Loss and pre-clip gradient norm of the last step (a step that computes garbage still runs fast)."""
        loss, gn = float(self.last_loss), float(self.train_step.grad_norm)
        return dict(loss=round(loss, 4), grad_norm_before_clip=round(gn, 3), finite=bool(np.isfinite(loss) and np.isfinite(gn)))

    def loss_terms(self):
        """Every loss term of one extra eager frame (names and values, for the record in the JSON line)."""
        from hipad_amd.frame import frame_losses
        with torch.no_grad():
            img, data = self.frames.next()
            was = self.model.head.onedecoder_head.run_step
            losses = frame_losses(self.model, img, data)
            self.model.head.onedecoder_head.run_step = was
        return {k: round(float(v), 4) for k, v in losses.items()}


    def frame_roofline(self, measured_ms_per_frame):
        """roofline.frame of the JSON line (hipad_amd.roofline): one extra eager forward with per-operator accounting."""
        from hipad_amd import roofline as RL
        daf = self.daf

        def daf_bytes(loc, kind):
            A, P = loc.shape[1:3]
            v = (loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1)
            d = dict(A=A, P=P, kept_pairs=int(v.sum()) // loc.shape[0], rows_touched=daf.unique_rows(loc[:1].contiguous()))
            return loc.shape[0] * daf.alg_bytes(d, kind)

        img, data = self.frames.next()
        cats = RL.census(self.model, img, data, daf_bytes)
        return RL.frame_roofline(cats, measured_ms_per_frame)

    def cpu_frame_baseline(self, seconds):
        """The whole training frame (encoder + decoder + losses, forward + backward, fp32) on the host cores: the mirrored
        modules on CPU tensors with oracle/ supplying the operators that exist only as HIP kernels (oracle/cpu_frame.py);
        all host cores, count stated."""
        from oracle import cpu_frame
        if self.stage != 2 or tuple(self.hw) != (256, 704):
            return None   # the CPU baseline is quoted on the headline configuration only
        r = cpu_frame.time_frames(seconds=seconds, plan_queries=self.plan_queries)
        return dict(value=round(r["frames"] / r["seconds"], 4), unit="frames/s", cores=r["cores"], kind="port",
                    sample=f"{r['frames']} whole stage-2 training frame(s) (ResNet50+FPN + decoder + losses, forward + "
                           f"backward, fp32, batch 1, plan {self.plan_queries}) after one untimed warm-up frame, "
                           f"{r['seconds']:.1f} s on {r['cores']} host threads: torch CPU ops for the mirrored modules, "
                           "oracle/daf_oracle.c (anchors dealt to the threads) for the aggregation, fp32 softmax attention "
                           "in place of the HIP kernel; optimiser step not included")

    def breakdown(self, reps=5):
        """ms per frame of the encoder forward, decoder forward and the rest, by HIP events, launched
        EAGERLY (so it includes launch gaps the graph replay does not have; shares, not the headline)."""
        from hipad_amd.frame import _frame_loss
        if self.model.use_grid_mask:
            self.model.grid_mask.external_randomize = False
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        acc = dict(encoder_fwd=0.0, decoder_fwd=0.0, backward_opt=0.0)  # encoder_fwd is folded into decoder_fwd
        for _ in range(reps):
            img, data = self.frames.next()
            self.train_step.grads.zero()
            e = [ev() for _ in range(4)]
            e[0].record()
            e[1].record()
            loss = _frame_loss(self.model, img, data)
            e[2].record()
            loss.backward()
            self.train_step.grads.check_views()
            self.train_step.update()
            e[3].record()
            e[3].synchronize()
            acc["encoder_fwd"] += e[0].elapsed_time(e[1]) / reps
            acc["decoder_fwd"] += e[1].elapsed_time(e[2]) / reps
            acc["backward_opt"] += e[2].elapsed_time(e[3]) / reps
        return {k: round(v, 3) for k, v in acc.items()}


class Stage2Infer:
    """Closed-loop style inference: one frame per step through the replayed network graph + track ids + result
    decoders (detections, map vectors, agent trajectories, plan way-points on the host)."""

    def __init__(self, device, seed, plan_queries=480):
        import warnings
        warnings.filterwarnings("ignore", category=DeprecationWarning)
        from hipad_amd.frame import GraphedInference, SyntheticFrames, build_detector
        torch.manual_seed(1234)
        self.model, _ = build_detector(stage=2, input_hw=(256, 704), plan_queries=plan_queries, device=device)
        self.frames = SyntheticFrames(bs=1, input_hw=(256, 704), device=device, seed=seed)
        self.graphed = GraphedInference(self.model, self.frames)
        self.daf = DafStage2(device, seed, plan_queries)

    def step(self):
        self.last = self.graphed()


def _daf_grid_threads(A, P, cams=6, bs=1):
    """Threads of the wave-per-item aggregation launches for (A, P): the host-side work split of
    hip-ad_amd/csrc/daf.hip make_plan(), restated to find this launch in the rocprofv3 counter files."""
    n_anchor = bs * A
    want_chunks = (4096 + n_anchor - 1) // n_anchor
    target = max(24, ((P + want_chunks - 1) // want_chunks) * cams)
    target = min(target, 128)
    ppc = min(P, max(1, target // cams))
    nchunks = (P + ppc - 1) // ppc
    ppc = (P + nchunks - 1) // nchunks
    nchunks = (P + ppc - 1) // ppc
    return ((n_anchor * nchunks + 3) // 4) * 256


PMC_FILE = "r03_daf_pmc_traffic.json"
MFMA_PMC_FILE = "r03_linear_path_mfma_pmc.json"


def pmc_traffic(kernel, A, P):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE,
    tools/pmc_traffic.py; MI355X_MICROARCH.md HBM section), or None when the file has no such launch."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)["kernels"]
    if kernel == "frame_pass":            # the frame's merged pass: every kernel of it, each launched once per pass
        parts = [v for k, v in table.items()
                 if k.startswith(("hipad::daf_tap_pass_kernel", "hipad::daf_alloc_kernel", "hipad::fill_zero_kernel"))]
        # the accumulation kernel of the frame pass = its launch with the largest grid (single-call pipelines launch fewer
        # workgroups; tools/pmc_daf_frame.py runs only the frame pass)
        feat = [(int(k.rsplit("=", 1)[1]), v) for k, v in table.items() if k.startswith("hipad::daf_bwd_feat_kernel grid=")]
        if feat:
            parts.append(max(feat, key=lambda kv: kv[0])[1])
        return sum(v["hbm_bytes_per_launch"] for v in parts) if parts else None
    else:
        hit = table.get("hipad::%s grid=%d" % (kernel, _daf_grid_threads(A, P)))
    return None if hit is None else hit["hbm_bytes_per_launch"]


def roofline_of(daf, layers=6):
    """Dominant hand-written kernel of the step (largest per-frame time = launches x average duration), measured live
    with HIP events on the launch stream, against its algorithmic bytes; plus the same numbers for every other
    aggregation launch: forward and grad loc+weights kernel of the four query sets (6 launches per frame each), the
    frame's merged feature-gradient pass (1 per frame), and -- for comparison, not part of the step any more -- the
    feature-gradient pipeline of a single call."""
    kt = daf.kernel_times()
    table, per_frame = {}, {}
    for (name, tag), ms in kt.items():
        if name == "frame":
            alg, launches = daf.frame_feat_alg_bytes(), 1
        else:
            dcall = next(d for d in daf.calls if d["name"] == name)
            alg, launches = daf.alg_bytes(dcall, tag.replace("_single_call", "")), layers
        gbs = alg / (ms * 1e-3) / 1e9
        in_step = not tag.endswith("_single_call")
        table[f"{name}_{tag}"] = dict(ms=round(ms, 4), alg_mbytes=round(alg / 1e6, 1), GBs=round(gbs, 1),
                                      frac=round(gbs / HBM_PEAK_GBS, 4),
                                      ms_per_frame=round(launches * ms, 3) if in_step else None,
                                      launches_per_frame=launches if in_step else 0)
        if in_step:
            per_frame[(name, tag)] = (launches * ms, alg, ms)
    dom = max(per_frame, key=lambda k: per_frame[k][0])
    _, alg, ms = per_frame[dom]
    achieved = alg / (ms * 1e-3) / 1e9
    if dom[0] == "frame":
        kernel = ("feature-gradient pass of the frame's 24 aggregation calls (hipad_daf_backward_feat_multi: fill, "
                  "daf_tap_pass x2, daf_alloc, daf_bwd_feat_kernel; one pass per frame)")
        extra = dict(rows_touched=daf.rows_touched_frame, kept_pairs=layers * sum(d["kept_pairs"] for d in daf.calls))
        traffic = pmc_traffic("frame_pass", None, None)
        note_ = "; sum over the pass's kernels: fill, count pass, alloc, place pass, accumulate)"
    else:
        dcall = next(d for d in daf.calls if d["name"] == dom[0])
        kname = {"fwd": "daf_fwd_c256_kernel<4, true> (+ combine)", "bwd_lw": "daf_bwd_lw_kernel<4, true, true>"}[dom[1]]
        pmc_name = {"fwd": "daf_fwd_c256_kernel<4, true, unsigned short>",
                    "bwd_lw": "daf_bwd_lw_kernel<4, true, true, unsigned short>"}[dom[1]]
        kernel = f"{kname} [{dom[0]}: A={dcall['A']} P={dcall['P']}]"
        extra = dict(rows_touched=dcall["rows_touched"], kept_pairs=dcall["kept_pairs"])
        traffic = pmc_traffic(pmc_name, dcall["A"], dcall["P"])
        note_ = ")"
    return dict(bound="hbm", kernel=kernel, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source="profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "tools/pmc_daf_frame.py -- the aggregation path as the step runs it --, bytes per launch" + note_,
                alg_bytes_per_launch=alg, avg_launch_ms=round(ms, 4), launches_per_frame=1 if dom[0] == "frame" else layers,
                aggregation_launches=table, **extra)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves as a CHILD
    `python -m torch.distributed.run` (before this process has touched the GPU; never an exec), relay its
    output (rank 0's JSON line) and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


_T0 = time.perf_counter()


def note(msg):
    """Progress line on stderr (a long bench run must show signs of life; stdout carries only the JSON line)."""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))
    rank, world, local = dist_setup(a.gpus)
    if world != a.gpus:
        raise SystemExit(f"bench: --gpus {a.gpus} but the launcher started {world} rank(s); refusing to report "
                         f"a {world}-rank number as the {a.gpus}-GPU point")
    dev = torch.device("cuda", local)
    full = a.workload in ("stage2_full", "stage2_full_frames", "stage2_r101_1600", "stage1_fp32")
    infer = a.workload == "stage2_infer"
    extra = {}
    if a.workload == "stage2_r101_1600":   # BASELINE.json config 5: the pyramid (522 MB fp32) leaves the Infinity Cache
        extra = dict(hw=(640, 1600), backbone_depth=101)
    elif a.workload == "stage1_fp32":      # BASELINE.json config 2
        extra = dict(stage=1, encoder_dtype=torch.float32)
    elif a.workload == "stage2_full_frames":   # the same step fed through the data pipeline
        extra = dict(pipeline_frames=True)
    comm = torch.bfloat16 if (a.comm_dtype == "bf16" and world > 1) else None
    wl = (Stage2Full(dev, seed=rank, plan_queries=a.plan_queries, bs=a.bs, eager=a.eager, comm_dtype=comm, **extra) if full
          else Stage2Infer(dev, seed=rank, plan_queries=a.plan_queries) if infer
          else DafStage2(dev, seed=rank, plan_queries=a.plan_queries))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    note("workload built (model, warm-up frames, graph capture)")
    for _ in range(a.warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    barrier()
    from hipad_amd.dist import max_over_ranks
    dt = max_over_ranks(time.perf_counter() - t0, dev)

    note("timed region done: %.3f ms per step" % (dt / a.steps * 1e3))
    sanity = wl.sanity() if full else None
    # eager extras run on EVERY rank: the loss path all-reduces its positive counts (reduce_mean), so a rank-0-only
    # call would leave the other ranks out of a collective
    loss_terms = wl.loss_terms() if full else None
    eager_breakdown = wl.breakdown() if (full and a.eager) else None
    if sanity is not None and not sanity["finite"]:
        raise SystemExit(f"bench: the training step went non-finite ({sanity}); refusing to report a throughput")
    daf = wl.daf if (full or infer) else wl
    roof = roofline_of(daf)
    note("aggregation launches timed")
    if full:
        roof["frame"] = wl.frame_roofline(dt / a.steps * 1e3 / a.bs)
        note("frame roofline census done")
    if full:
        hw_txt = "%dx%d" % (wl.hw[1], wl.hw[0])
        enc_txt = {"stage2_full": "ResNet50+FPN bf16 channels-last",
                   "stage2_full_frames": "ResNet50+FPN bf16 channels-last, INPUT THROUGH THE DATA PIPELINE (six uint8 1600x900 "
                   "frames per sample resident in HBM -> sequence-grouped sampler -> per-sequence resize/crop/flip/rotate draw "
                   "-> device image pipeline, Pillow-exact, + normalisation -> GridMask -> step; augmented projection matrices "
                   "composed per frame)", "stage2_r101_1600": "ResNet101+FPN bf16 channels-last "
                   "(BASELINE config 5: 510 000 pyramid positions, 522 MB fp32, outside the Infinity Cache)",
                   "stage1_fp32": "ResNet50+FPN fp32 (BASELINE config 2: hipad_b2d_stage1, no motion head)"}[a.workload]
        workload = (f"{a.workload}: one training step (forward + the reference's losses with device-side Hungarian target "
                    "assignment [det/map/motion/ego/plan/depth terms] + backward + grad all-reduce + clip + AdamW) of "
                    f"hipad_b2d_stage{wl.stage} on one 6-cam {hw_txt} frame per GPU: {enc_txt}, decoder det 900 "
                    f"+ map 100 + plan {a.plan_queries} + ego 1 queries x 6 layers + motion head, bf16 GEMMs / bf16-operand "
                    "attention, fp32 aggregation arithmetic on the encoder's bf16 pyramid rows; synthetic ground truth (~20 boxes, ~10 map lines per frame)")
        dtype = "f32" if a.workload == "stage1_fp32" else "bf16"
        cfg = dict(workload=workload, frames_per_gpu_per_step=a.bs, plan_queries=a.plan_queries, parallelism=f"dp{world}",
                   grad_allreduce_dtype=("none (1 rank)" if world == 1 else a.comm_dtype),
                   launch="eager (decoder-segment all-reduce overlapped with the encoder's backward)" if a.eager else
                   ("hipGraph replay: forward+losses+backward graph, clip+AdamW graph" if world == 1 else
                    "hipGraph replay: forward graph | positive-count all-reduce | losses + decoder-backward graph | "
                    "all-reduce of the decoder gradient segment (RCCL, side stream) overlapped with the encoder-backward "
                    "graph | all-reduce of the encoder segment | clip+AdamW graph"),
                   with_cp=False, with_cp_note="activation checkpointing of the backbone (reference config: with_cp=True, "
                   "projects/configs/hipad_b2d_stage2.py:119) is OFF: 288 GB of HBM hold the activations, so the "
                   "reference's backbone re-computation in the backward is not part of this step",
                   eager_frame_breakdown_ms=eager_breakdown, last_step=sanity, loss_terms=loss_terms,
                   roofline_scope="`roofline`: the aggregation launch with the largest per-frame time (live HIP-event timing, "
                                  "algorithmic bytes from the kernels' own tap indices); `roofline.frame`: sum over ALL "
                                  "operators of the step of max(bytes / 8 TB/s, flops / 2.5 PF) against the measured step; "
                                  "MFMA utilisation of the linear path: profiles/" + MFMA_PMC_FILE)
    elif infer:
        workload = ("stage2_infer: closed-loop style inference of hipad_b2d_stage2, batch 1, one 6-cam 704x256 frame per step: "
                    "encoder + decoder replayed from a hipGraph, then track ids and the result decoders (boxes, map vectors, "
                    "agent trajectories, plan way-points incl. collision rescoring) with their device->host copies; ms_per_step "
                    "is the per-frame latency")
        dtype = "bf16"
        cfg = dict(workload=workload, frames_per_gpu_per_step=1, plan_queries=a.plan_queries, parallelism=f"dp{world}")
    else:
        workload = ("daf_stage2: the 24 deformable-aggregation calls (6 layers x det 900x13, map 100x300, "
                    f"plan {a.plan_queries}x90, ego 1x13) fwd+bwd of one stage-2 frame, 6 cams 704x256, "
                    "89760-position pyramid of bf16 rows (the encoder's output dtype) read in place, fp32 arithmetic, fp32 "
                    "feature gradient; aggregation path only (image encoder, attention/FFN and losses not included)")
        dtype = "f32"   # the arithmetic type of the path (rows are widened bf16 -> fp32 in registers)
        cfg = dict(workload=workload, frames_per_gpu_per_step=1, plan_queries=a.plan_queries, parallelism=f"dp{world}")
    out = dict(metric="frames/sec (6-cam 704x256, 900+100+6+48 queries) fwd+bwd at 1/2/4/8 GPUs",
               value=round(world * a.steps * (a.bs if full else 1) / dt, 3), unit="frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=round(dt / a.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype=dtype, data="synthetic", config=cfg, roofline=roof)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the measurement is complete here; the CPU leg below only adds `cpu_baseline`.  stdout carries exactly ONE JSON
        # line (the contract), so the line as it stands goes to stderr and to gpurun_out/ first: a kill during the CPU
        # leg (two round-2 runs ended there with an empty stdout) can no longer lose the GPU numbers.
        partial = json.dumps(dict(out, cpu_baseline=None, partial="cpu_baseline leg still to run"))
        print("[bench partial] " + partial, file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_partial.json"), "w") as f:
                f.write(partial + "\n")
        except OSError:
            pass
        note("cpu baseline ...")
        out["cpu_baseline"] = wl.cpu_frame_baseline(a.cpu_seconds) if full else daf.cpu_baseline(a.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    note("done")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
