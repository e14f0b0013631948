"""Encoder only (GridMask off): gradients from a captured fwd+bwd vs the same thing run eagerly."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
if os.environ.get("IMM") == "1":
    torch.backends.miopen.immediate = True
if os.environ.get("BENCH") == "1":
    torch.backends.cudnn.benchmark = True
from hipad_amd.frame import build_detector
torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
model.use_grid_mask = False
if os.environ.get("NOCUDNN") == "1":
    torch.backends.cudnn.enabled = False
if os.environ.get("FP32") == "1":
    model.encoder_dtype = torch.float32
print("IMM", os.environ.get("IMM"), "NOCUDNN", os.environ.get("NOCUDNN"), "FP32", os.environ.get("FP32"), flush=True)
enc = [p for n, p in model.named_parameters() if n.startswith(("img_backbone", "img_neck", "depth_branch")) and p.requires_grad]
names = [n for n, p in model.named_parameters() if n.startswith(("img_backbone", "img_neck", "depth_branch")) and p.requires_grad]
img = torch.randn(1, 6, 3, 256, 704, device="cuda")
R = None


def fwd_bwd():
    global R
    for p in enc:
        p.grad = None
    fm, depths = model.extract_feat(img, True, {})
    if R is None:
        R = torch.randn_like(fm[0])
    loss = (fm[0] * R).sum() * 1e-3 + sum(d.float().mean() for d in depths)
    loss.backward()
    return loss


def bn_state():
    return [b.clone() for b in model.buffers()]


def restore(state):
    for b, s in zip(model.buffers(), state):
        b.copy_(s)


state = bn_state()
for _ in range(3):
    restore(state)
    fwd_bwd()
ref = [p.grad.clone() for p in enc]
restore(state); fwd_bwd()
print("eager vs eager max rel diff", max(float((p.grad - r).abs().max() / r.abs().max().clamp(min=1e-20)) for p, r in zip(enc, ref)), flush=True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    restore(state); fwd_bwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
restore(state)
static = [torch.zeros_like(p) for p in enc]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd()
    for s, p in zip(static, enc):
        s.copy_(p.grad)
for i in range(3):
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 3 << 20, 1 << 24)]
    torch.cuda.synchronize(); del junk
    restore(state)
    g.replay(); torch.cuda.synchronize()
    bad = []
    for n, s, r, p in zip(names, static, ref, enc):
        err = float((s - r).abs().nan_to_num(1e30).max() / r.abs().max().clamp(min=1e-20))
        if not (err < 0.1):
            bad.append((n, tuple(p.shape), "%.2e" % err))
    print("replay", i, "params off by >10%% of max|grad|: %d of %d" % (len(bad), len(enc)), "last bad:", bad[-3:], flush=True)
