"""Shared helpers for the rocprofv3 kernel-trace summaries: load the CSV, cut an EXACT number of steady-state training
steps out of it (the span between the ends of two optimiser launches, `adamw_flat_kernel` runs once per step)."""
import csv
import glob


def load(path):
    f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    return rows


def steady_steps(rows, nsteps=3, marker="adamw_flat_kernel", skip_last=0):
    """Rows of the last ``nsteps`` whole steps (ending ``skip_last`` steps before the final one) and that count."""
    ends = [e for s, e, n, g in rows if marker in n]
    if len(ends) < nsteps + 1 + skip_last:
        raise SystemExit(f"only {len(ends)} optimiser launches in the trace")
    hi = ends[-1 - skip_last]
    lo = ends[-1 - skip_last - nsteps]
    return [r for r in rows if lo < r[1] <= hi], nsteps, (hi - lo) / 1e6 / nsteps
