"""Summarise the LAST `window_ms` of a rocprofv3 kernel trace (steady state only): top kernels by time."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
window = float(sys.argv[2]) * 1e6
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= end - window]
span = (end - min(r[0] for r in sel)) / 1e6
busy = sum(r[1] - r[0] for r in sel) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in sel:
    k = n.replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:150]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
print(f"window {span:.1f} ms, {len(sel)} dispatches, GPU busy {busy:.1f} ms ({100*busy/span:.0f}%)")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t:8.2f} ms {100*t/busy:5.1f}% {c:6d} calls {1e3*t/c:8.1f} us  {k}")
