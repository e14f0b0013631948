"""Steady-state kernel time of a rocprofv3 kernel trace by category (last `window_ms`), plus the top kernels
of the elementwise / copy categories with their grid sizes."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
window = float(sys.argv[2]) * 1e6
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
CATS = [("encoder (MIOpen conv/BN, bf16 elementwise)", r"igemm|ck::|_ZN2ck|MIOpen|SubTensorOp|BFloat16|bfloat16|max_pool|batch_norm|upsample|Bf16|threshold"),
        ("linear (hipad gemm, fused backward)", r"hipad::gemm|hipad::linear_"),
        ("aggregation (daf, weights softmax, projection)", r"hipad::daf|hipad::weights_softmax|hipad::project|hipad::fill_zero|hipad::proj"),
        ("attention (hipad)", r"hipad::attn"),
        ("optimizer", r"hipad::adamw|hipad::grad_sqnorm|multi_tensor"),
        ("layer norm", r"layer_norm|GammaBeta|LayerNorm|hipad::layernorm"),
        ("copies / fills / cat", r"copyBuffer|fillBuffer|FillFunctor|direct_copy|CatArray|copy_kernel"),
        ("reductions", r"reduce_kernel"),
        ("elementwise", r"elementwise|masked_scale|dropout"),
        ("sort / topk / gather / index", r"sort|topk|gather|index|scatter|radix|bitonic")]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= end - window]
agg = collections.OrderedDict((c, [0, 0.0]) for c, _ in CATS)
agg["other"] = [0, 0.0]
other = collections.Counter()
for s, e, n in sel:
    for c, pat in CATS:
        if re.search(pat, n):
            agg[c][0] += 1; agg[c][1] += (e - s) / 1e6
            break
    else:
        agg["other"][0] += 1; agg["other"][1] += (e - s) / 1e6
        other[n[:90]] += (e - s) / 1e6
busy = sum(v[1] for v in agg.values())
print(f"window {window/1e6:.0f} ms = {steps:g} steps; per step: {len(sel)/steps:.0f} dispatches, {busy/steps:.1f} ms of kernel time")
for c, (cnt, t) in agg.items():
    print(f"  {t/steps:7.2f} ms {100*t/busy:5.1f}% {cnt/steps:7.0f} launches  {c}")
print("largest 'other':", [(k, round(v / steps, 2)) for k, v in other.most_common(8)])
