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


def fill_parameters_by_name(module, seed_base, scale=0.05):
    """Seed every trainable parameter from a hash of its NAME (independent of construction order):
    matrices ~ N(0, scale), vectors (biases, norm weights) ~ N(0, scale) + (1 for *norm* weights)."""
    import zlib
    total = torch.zeros(3, dtype=torch.float64)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue
            seed = (zlib.crc32(name.encode()) + seed_base) % (2 ** 31)
            val = seeded(tuple(p.shape), seed, scale)
            if p.dim() == 1 and name.endswith("weight"):
                val = val + 1.0  # LayerNorm / BatchNorm / Scale gains stay near one
            p.copy_(val)
            total += checksum(p)
    return total
