"""Top kernels of a rocprofv3 --stats kernel_stats.csv."""
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".csv") else glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", round(tot / 1e6, 1), "dispatches", sum(int(r["Calls"]) for r in rows))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):7d} calls "
          f"{float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:120]}")
