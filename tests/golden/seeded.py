"""Seeded tensors shared by tests/golden/make_golden.py (generator) and the tests (consumer).

Large random inputs / parameters are NOT stored in the fixtures; both sides re-draw them
from torch's CPU generator (deterministic for a given torch build -- the GPU box runs this
same image) and the fixture carries a float64 checksum so a drifted RNG stream fails loudly
instead of producing a confusing numerical mismatch.
"""
import torch


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g) * scale


def checksum(t):
    t = t.detach().double().flatten()
    idx = torch.arange(1, t.numel() + 1, dtype=torch.float64)
    return torch.stack([t.sum(), t.abs().sum(), (t * (idx % 97)).sum()])


def fill_parameters(module, seed_base, scale=0.08):
    """Overwrite every trainable parameter of `module` with seeded normals, in named_parameters order."""
    sums = []
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            if not p.requires_grad:
                continue
            p.copy_(seeded(tuple(p.shape), seed_base + i, scale))
            sums.append(checksum(p))
    return torch.stack(sums).sum(0)
