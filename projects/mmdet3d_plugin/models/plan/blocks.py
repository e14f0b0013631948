"""Planning refinement heads (registered names / keywords / parameter names of the reference's
models/plan/blocks.py:16-157)."""
import torch
import torch.nn as nn

from hipad_amd.compat import MLPStack, PLUGIN_LAYERS, BaseModule, Linear, Scale, bias_init_with_prob

from ..blocks import linear_relu_ln

__all__ = ["SparsePlanRefinementModule", "SparsePlanAlignRefinementModule"]


def _cls_head(embed_dims):
    return MLPStack(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, 1))


def _reg_head(embed_dims, out_dim):
    return MLPStack(*linear_relu_ln(embed_dims, 2, 2), Linear(embed_dims, out_dim), Scale([1.0] * out_dim))


@PLUGIN_LAYERS.register_module()
class SparsePlanRefinementModule(BaseModule):
    def __init__(self, embed_dims=256, ego_fut_ts=6, ego_fut_cmd=3, ego_fut_mode=3, add_anchor=False):
        super().__init__()
        self.embed_dims, self.ego_fut_ts, self.ego_fut_cmd, self.ego_fut_mode = embed_dims, ego_fut_ts, ego_fut_cmd, ego_fut_mode
        self.add_anchor = add_anchor
        self.plan_cls_branch = _cls_head(embed_dims)
        self.plan_reg_branch = _reg_head(embed_dims, ego_fut_ts * 2)

    def init_weight(self):
        nn.init.constant_(self.plan_cls_branch[-1].bias, bias_init_with_prob(0.01))

    def forward(self, instance_feature, anchor, anchor_embed, use_plan_anchor_embed=True):
        from hipad_amd import chain as CH
        if CH.usable(instance_feature):
            specs = (CH.spec_of(self.plan_reg_branch), CH.spec_of(self.plan_cls_branch))
            if all(sp is not None for sp in specs):  # both stacks as ONE chain launch, input sum + residual anchor inside
                return CH.run([CH.Call(specs[0], instance_feature, anchor_embed if use_plan_anchor_embed else None,
                                       residual=anchor), CH.Call(specs[1], instance_feature)])
        src = instance_feature + anchor_embed if use_plan_anchor_embed else instance_feature
        return self.plan_reg_branch(src) + anchor, self.plan_cls_branch(instance_feature)


@PLUGIN_LAYERS.register_module()
class SparsePlanAlignRefinementModule(BaseModule):
    """Way-point regression for several anchor groups that share their mode scores.

    The query tensor holds ``len(anchor_types)`` equal chunks (one per anchor group).  All
    "temp"/"spat" groups are summed into one aligned query that feeds their regression branches and
    one shared score branch; each "speed" group adds, on top of that, the sum of the speed groups of
    its speed interval over the frequencies, and is scored by a second branch (reference
    models/plan/blocks.py:53-157).
    """

    def __init__(self, embed_dims=256, ego_fut_ts=6, ego_fut_cmd=3, ego_fut_mode=3, anchor_types=None):
        super().__init__()
        self.embed_dims, self.ego_fut_ts, self.ego_fut_cmd, self.ego_fut_mode = embed_dims, ego_fut_ts, ego_fut_cmd, ego_fut_mode
        self.anchor_types = anchor_types
        self.anchor_group = len(anchor_types)
        self.plan_cls_branch = _cls_head(embed_dims)
        by_freq = {}
        for t in anchor_types:
            if t[0] == "speed":
                by_freq.setdefault(t[1], []).append(t[2])
        if by_freq:
            areas = list(by_freq.values())
            self.speed_areas = areas[0]
            if any(a != self.speed_areas for a in areas[1:]):
                raise AssertionError("every speed frequency must list the same intervals")
            self.plan_cls_branch_speed = _cls_head(embed_dims)
        for t in anchor_types:
            name = f"plan_reg_branch_{t[0]}_{t[1]}"
            setattr(self, name, _reg_head(embed_dims, ego_fut_ts * 2))  # same-named groups share (last one wins)

    def init_weight(self):
        prior = bias_init_with_prob(0.01)
        nn.init.constant_(self.plan_cls_branch[-1].bias, prior)
        if hasattr(self, "plan_cls_branch_speed"):
            nn.init.constant_(self.plan_cls_branch_speed[-1].bias, prior)

    def _forward_chains(self, aligned, speed_in, areas, anchor):
        """Every distinct (branch, input) pair of the module as one chain of ONE grouped launch (eight chains for the ten
        anchor types of the HiP-AD configs) instead of ~50 Linear / LayerNorm launches on 48..144 rows.  With batch 1
        the regression chains write their rows of the stacked (1, groups * modes, 2 * ts) result directly."""
        from hipad_amd import chain as CH
        bs, rows = aligned.shape[:2]
        n_area = max(1, len(areas))
        stacked = bs == 1   # rows of one group are contiguous in the stacked tensor only for a single sample
        plan, reg_calls, cls_calls = [], {}, {}
        for i, t in enumerate(self.anchor_types):
            branch = getattr(self, f"plan_reg_branch_{t[0]}_{t[1]}")
            if t[0] in ("temp", "spat"):
                rk, ck, piece, x, cls_mod = (id(branch), "aligned"), (id(self.plan_cls_branch), "aligned"), None, aligned, self.plan_cls_branch
            elif t[0] == "speed":
                rk, ck, piece, x, cls_mod = (id(branch), "speed"), (id(self.plan_cls_branch_speed), "speed"), areas.index(t[2]), speed_in, self.plan_cls_branch_speed
            else:
                raise NotImplementedError(t[0])
            reg_calls.setdefault(rk, dict(module=branch, x=x, first=i, members=[]))["members"].append((i, piece))
            cls_calls.setdefault(ck, dict(module=cls_mod, x=x))
            plan.append((rk, ck, piece))
        # a regression chain may write straight into the stacked result when its pieces are consecutive anchor groups
        for c in reg_calls.values():
            idx = [i for i, _ in c["members"]]
            pcs = [p for _, p in c["members"]]
            c["direct"] = stacked and (pcs == [None] or (pcs == list(range(n_area)) and idx == list(range(idx[0], idx[0] + n_area))))
        all_direct = all(c["direct"] for c in reg_calls.values())
        if not all_direct:  # mixed placement is not produced by any config: all or nothing keeps the code simple
            for c in reg_calls.values():
                c["direct"] = False
        calls, order = [], []
        for key, c in list(reg_calls.items()) + list(cls_calls.items()):
            spec = CH.spec_of(c["module"])
            if spec is None:
                return None
            slot = None
            if c.get("direct"):
                slot = CH.OutSlot("reg", (1, rows * len(self.anchor_types), spec.N_out), row0=c["first"] * rows)
            calls.append(CH.Call(spec, c["x"], out_slot=slot))
            order.append((key, slot is not None))
        outs = list(CH.run(calls))
        private = {key: None for key, direct in order if not direct}
        it = iter(outs)
        for key in private:
            private[key] = next(it)
        pieces_of = {key: (t.split(rows, dim=1) if t.shape[1] != rows else (t,)) for key, t in private.items()}
        if all_direct:
            regs = outs[-1]
        else:
            regs = torch.cat([pieces_of[rk][0 if piece is None else piece] for rk, _, piece in plan], dim=1)
        scores = torch.cat([pieces_of[ck][0 if piece is None else piece] for _, ck, piece in plan], dim=1)
        return regs + anchor, scores

    def _mix_table(self):
        """Rows = [aligned] + one speed query per interval, columns = anchor groups: which group queries each sums."""
        areas = list(getattr(self, "speed_areas", []))
        aligned = [1.0 if t[0] in ("temp", "spat") else 0.0 for t in self.anchor_types]
        table = [aligned]
        for area in areas:
            table.append([a + (1.0 if (t[0] == "speed" and t[2] == area) else 0.0) for a, t in zip(aligned, self.anchor_types)])
        return table, areas

    def forward(self, instance_feature, anchor, anchor_embed, use_plan_anchor_embed=True):
        from hipad_amd import chain as CH
        from hipad_amd import functional as HF
        if CH.usable(instance_feature) and instance_feature.shape[1] % self.anchor_group == 0 \
                and instance_feature.shape[-1] % 4 == 0:
            # GPU path: the group mixing in one launch, then every distinct (branch, input) pair as one chain of ONE
            # grouped launch
            table, areas = self._mix_table()
            rows = instance_feature.shape[1] // self.anchor_group
            mixed = HF.chunk_mix(instance_feature, anchor_embed if use_plan_anchor_embed else None, table, rows)
            if areas:
                aligned, speed_in = mixed.split([rows, rows * len(areas)], dim=1)
            else:
                aligned, speed_in = mixed, None
            out = self._forward_chains(aligned, speed_in, areas, anchor)
            if out is not None:
                return out
        return self._forward_layers(instance_feature, anchor, anchor_embed, use_plan_anchor_embed)

    def _forward_layers(self, instance_feature, anchor, anchor_embed, use_plan_anchor_embed=True):
        if use_plan_anchor_embed:
            instance_feature = instance_feature + anchor_embed
        chunks = instance_feature.chunk(self.anchor_group, dim=1)
        aligned = sum(c for c, t in zip(chunks, self.anchor_types) if t[0] in ("temp", "spat"))
        speed_query = {}
        if hasattr(self, "speed_areas"):
            for area in self.speed_areas:
                same_area = [c for c, t in zip(chunks, self.anchor_types) if t[0] == "speed" and t[2] == area]
                speed_query[area] = aligned + sum(same_area)
        # The reference evaluates one (regression, score) pair per anchor type; many of those calls
        # repeat the same module on the same input (the score head of all temp/spat groups, both heads
        # of the speed groups that share an interval).  Evaluate every distinct (module, input) once and
        # stack the inputs of a shared module into one call.
        areas = list(speed_query)
        cache = {}

        def run(module, key, sources):
            if (id(module), key) not in cache:
                out = module(torch.cat(sources, dim=1) if len(sources) > 1 else sources[0])
                cache[(id(module), key)] = out.chunk(len(sources), dim=1)
            return cache[(id(module), key)]

        regs, scores = [], []
        for t in self.anchor_types:
            branch = getattr(self, f"plan_reg_branch_{t[0]}_{t[1]}")
            if t[0] in ("temp", "spat"):
                regs.append(run(branch, "aligned", [aligned])[0])
                scores.append(run(self.plan_cls_branch, "aligned", [aligned])[0])
            elif t[0] == "speed":
                i = areas.index(t[2])
                regs.append(run(branch, "speed", [speed_query[a] for a in areas])[i])
                scores.append(run(self.plan_cls_branch_speed, "speed", [speed_query[a] for a in areas])[i])
            else:
                raise NotImplementedError(t[0])
        return torch.cat(regs, dim=1) + anchor, torch.cat(scores, dim=1)
