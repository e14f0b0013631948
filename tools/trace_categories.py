"""Steady-state kernel time of a rocprofv3 kernel trace by category (last `window_ms`), plus the top kernels
of the elementwise / copy categories with their grid sizes."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _trace
# usage: trace_categories.py <dir> <window_ms> <steps>   (last window_ms of the trace, divided by `steps`)
#    or: trace_categories.py <dir> steps <n> [skip_last]  (EXACTLY the last n optimiser-delimited steps)
EXACT = sys.argv[2] == "steps"
CATS = [("encoder (MIOpen / CK conv, hipad BatchNorm, bf16 elementwise)", r"igemm|ck::|_ZN2ck|MIOpen|SubTensorOp|BFloat16|bfloat16|max_pool|batch_norm|upsample|Bf16|threshold|hipad::bn_|hipad::grid_mask"),
        ("linear (hipad gemm, fused backward)", r"hipad::gemm|hipad::linear_"),
        ("MLP chains (hipad chain fwd / bwd / dW, operand pack)", r"hipad::chain_|hipad::pack_weights"),
        ("aggregation (daf, weights softmax, projection)", r"hipad::daf|hipad::weights_softmax|hipad::project|hipad::fill_zero|hipad::proj"),
        ("attention (hipad)", r"hipad::attn"),
        ("optimizer", r"hipad::adamw|hipad::grad_sqnorm|multi_tensor"),
        ("layer norm", r"layer_norm|GammaBeta|LayerNorm|hipad::layernorm"),
        ("copies / fills / cat", r"copyBuffer|fillBuffer|FillFunctor|direct_copy|CatArray|copy_kernel"),
        ("reductions", r"reduce_kernel"),
        ("elementwise", r"elementwise|masked_scale|dropout"),
        ("sort / topk / gather / index", r"sort|topk|gather|index|scatter|radix|bitonic")]
allrows = _trace.load(sys.argv[1])
if EXACT:
    sel4, steps, span_ms = _trace.steady_steps(allrows, int(sys.argv[3]), skip_last=int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    window = span_ms * steps * 1e6
else:
    window = float(sys.argv[2]) * 1e6
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    end = max(r[1] for r in allrows)
    sel4 = [r for r in allrows if r[0] >= end - window]
sel = [(s_, e_, n_) for s_, e_, n_, g_ in sel4]
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
print(f"window {window/1e6:.1f} ms = {steps:g} steps{' (optimiser-delimited, exact)' if EXACT else ''}; per step: {len(sel)/steps:.0f} dispatches, {busy/steps:.2f} ms of kernel time, {window/1e6/steps:.2f} ms wall under tracing")
for c, (cnt, t) in agg.items():
    print(f"  {t/steps:7.2f} ms {100*t/busy:5.1f}% {cnt/steps:7.0f} launches  {c}")
print("largest 'other':", [(k, round(v / steps, 2)) for k, v in other.most_common(8)])
