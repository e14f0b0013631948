"""Modality routing of the grouped attention modules (reference models/separate_attn.py:25-721): the group-granular
split (one piece per group of neighbouring modalities) computes exactly what the general per-modality gather computes --
outputs and gradients -- for the three module kinds with the stage-2 group lists, with and without a temporal cache."""
import copy

import pytest
import torch

from hipad_amd.compat import ATTENTION, build_from_cfg

SELECT = ["det", "map", "plan", "ego"]
ATTN = dict(type="MultiheadFlashAttention", embed_dims=64, num_heads=2, batch_first=True, dropout=0.0)
ATTN2 = dict(ATTN, embed_dims=128)
KINDS = {
    "separate": dict(type="SeparateAttention", attn=[ATTN2, ATTN], query_select=SELECT, separate_list=[["det"], ["map"]],
                     decouple_list=[True, False]),
    "interactive": dict(type="InteractiveAttention", attn=ATTN, query_select=SELECT, query_list=[["plan", "ego"]],
                        key_list=[["det", "map"]], decouple_list=[False]),
    "temporal": dict(type="TemporalSeparateAttention", attn=[ATTN2, ATTN, ATTN], query_select=SELECT,
                     query_list=[["det"], ["map"], ["plan", "ego"]], key_list=[["det"], ["map"], ["det", "map"]],
                     decouple_list=[True, False, False]),
}


@pytest.mark.parametrize("kind", sorted(KINDS))
@pytest.mark.parametrize("cached", [False, True])
def test_group_split_equals_per_modality_gather(kind, cached, monkeypatch):
    from hipad_amd import functional as HF
    from oracle import cpu_frame
    monkeypatch.setattr(HF, "attention", cpu_frame.attention)      # the attention core has no CPU path of its own
    import projects.mmdet3d_plugin.models.attention  # noqa: F401  (registers the attention class)
    import projects.mmdet3d_plugin.models.separate_attn  # noqa: F401
    torch.manual_seed(0)
    mod = build_from_cfg(copy.deepcopy(KINDS[kind]), ATTENTION).eval()
    sizes, tsizes = [9, 4, 6, 1], [5, 0, 6, 1]
    cum = torch.tensor([0] + sizes).cumsum(0)
    tcum = torch.tensor([0] + tsizes).cumsum(0)
    fc_before, fc_after = torch.nn.Linear(64, 128), torch.nn.Linear(128, 64)

    def run(general):
        leaves = [torch.randn(2, sum(sizes), 64, generator=torch.Generator().manual_seed(i)).requires_grad_(True) for i in (1, 2)]
        query, qpos = leaves
        kw = dict(num_anchor_cumsum=cum, fc_before=fc_before, fc_after=fc_after)
        if cached and kind == "temporal":
            key, kpos = [torch.randn(2, sum(tsizes), 64, generator=torch.Generator().manual_seed(i)).requires_grad_(True) for i in (3, 4)]
            leaves += [key, kpos]
            kw.update(key=key, value=key, key_pos=kpos, num_temp_anchor_cumsum=tcum)
        if general:
            saved = type(mod)._route


            def general_route(self, q, k, v, qp, kp, qc, kc, fb, fa, read_updated=False, attn_mask=None):
                return self._route_pieces(q, k, v, qp, kp, qc, kc, fb, fa, read_updated)

            type(mod)._route = general_route
        try:
            out = mod(leaves[0], query_pos=leaves[1], **kw)
        finally:
            if general:
                type(mod)._route = saved
        for p in list(mod.parameters()) + list(fc_before.parameters()) + list(fc_after.parameters()):
            p.grad = None
        out.square().sum().backward()
        return out.detach(), [t.grad.clone() for t in leaves], [p.grad.clone() for p in mod.parameters()]

    a, b = run(False), run(True)
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-6)
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-5)
