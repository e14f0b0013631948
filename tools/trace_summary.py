"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid size): count, avg/min/max us."""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = max(glob.glob(path + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)   # the traced program, not a helper process
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hipad::", "")
    agg[(name[:40], "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"]), r["VGPR_Count"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(f"{'kernel':42s} {'grid':>12s} {'vgpr':>5s} {'n':>5s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k[0]:42s} {k[1]:>12s} {k[2]:>5s} {len(v):5d} {sum(v)/len(v):9.1f} {min(v):9.1f} {max(v):9.1f}")
