"""Top kernels of the last `window_ms` of a rocprofv3 kernel trace by LAUNCH COUNT, with typical grid sizes."""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _trace
# usage: trace_counts.py <dir> <window_ms> <steps> [top]   or   trace_counts.py <dir> steps <n> [top]
rows = _trace.load(sys.argv[1])
if sys.argv[2] == "steps":
    sel, steps, _ = _trace.steady_steps(rows, int(sys.argv[3]))
else:
    window = float(sys.argv[2]) * 1e6
    steps = float(sys.argv[3])
    end = max(r[1] for r in rows)
    sel = [r for r in rows if r[0] >= end - window]
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for s, e, n, g in sel:
    k = n.replace("(anonymous namespace)::", "").replace("at::native::", "").replace("void ", "")
    k = k.split("(")[0][:110]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e6; agg[k][2][g] += 1
print(f"per step: {len(sel)/steps:.0f} launches")
for k, (c, t, grids) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print(f"{c/steps:7.0f} /step {t/steps:6.2f} ms  {k}   grids {[(g, round(n/steps)) for g, n in grids.most_common(4)]}")
