import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hipad_amd, torch
torch.backends.cudnn.benchmark = True
from projects.mmdet3d_plugin.models.plan.instance_bank import front_view_encoder
torch.manual_seed(0)
enc = front_view_encoder(256, (8, 22)).cuda().train()
x = torch.randn(1, 256, 8, 22, device="cuda")
outs = []
for r in range(6):
    h = x
    per = []
    for m in enc:
        h = m(h)
        per.append(h.clone())
    outs.append(per)
for i, m in enumerate(enc):
    d = max(float((outs[0][i] - outs[r][i]).abs().max()) for r in range(1, 6))
    print(i, type(m).__name__, "max abs diff over 5 repeats:", d)
