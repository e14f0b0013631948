"""How well-conditioned is the gradient of the objective at random initialisation?  CPU, fp32, plain torch ops + the C
oracle (oracle/cpu_frame.py) -- none of the HIP kernels involved.  One cold frame twice: as it is, and with the feature
pyramid multiplied by (1 + EPS * N(0,1)) (EPS = 1e-4: the size of the run-to-run forward noise of the bf16 encoder on the
GPU), every discrete choice of the second run replayed from the first.  Prints the relative L2 change of the decoder's
gradient, per parameter group.  (Answers whether the 30-80 % run-to-run differences of the GPU step's gradient are the
kernels' doing or the objective's: tools/graph_vs_eager_pinned.py.)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import torch
from oracle import cpu_frame
from hipad_amd import compat as CR
from hipad_amd.frame import SyntheticFrames, build_detector, frame_losses
from test_graph_step_gpu import Choices, quiet

EPS = float(os.environ.get("EPS", "1e-4"))
with cpu_frame.cpu_path(int(os.environ.get("THREADS", "8"))):
    torch.manual_seed(5)
    model, _ = build_detector(stage=2, plan_queries=480, device="cpu")
    model.encoder_dtype = torch.float32
    model.train(); quiet(model)
    model.head.onedecoder_head.with_instance_id = False
    img, data = SyntheticFrames(bs=1, device="cpu", seed=3).next()
    plain = model.extract_feat
    grads, recorded = [], None
    for run in range(2):
        if run == 1:
            gen = torch.Generator().manual_seed(99)

            def noisy(*a, **k):
                fm, depths = plain(*a, **k)
                fm[0] = fm[0] * (1 + EPS * torch.randn(fm[0].shape, generator=gen))
                return fm, depths
            model.extract_feat = noisy
        for bank in ("det", "map", "plan", "ego"):
            getattr(model.head.onedecoder_head, bank + "_instance_bank").reset()
        model.head.onedecoder_head.run_step = 0
        ch = Choices(recorded, None)
        ch.begin_frame(0)
        identity, scope = CR.discrete_choice[0], CR.discrete_scope[0]
        CR.discrete_choice[0], CR.discrete_scope[0] = ch, "all"
        try:
            losses = frame_losses(model, img, data)
            total = sum(losses.values())
            model.zero_grad(set_to_none=True)
            total.backward()
        finally:
            CR.discrete_choice[0], CR.discrete_scope[0] = identity, scope
        recorded = ch.frames
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        print("run %d: loss %.6f" % (run, float(total)), flush=True)
a, b = grads
dec = [n for n in a if n.startswith("head.")]
num = sum(float((a[n] - b[n]).double().pow(2).sum()) for n in dec)
den = sum(float(a[n].double().pow(2).sum()) for n in dec)
print("decoder gradient: relative L2 change %.3e for a %.0e relative perturbation of the pyramid (|g| %.2f)" % ((num / den) ** 0.5, EPS, den ** 0.5))
rows = sorted(((float((a[n] - b[n]).double().pow(2).sum()) / num, float(a[n].norm()), float(b[n].norm()), n) for n in dec), reverse=True)
for share, na, nb, n in rows[:12]:
    print("  %6.2f%%  |a| %9.3f |b| %9.3f  %s" % (100 * share, na, nb, n))
