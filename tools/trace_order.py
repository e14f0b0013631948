"""Print the last N kernel dispatches of a rocprofv3 kernel trace in start order (name, grid, workgroup, LDS)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[-n:]:
    print(r.get("Queue_Id"), r.get("Stream_Id", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us grid",
          r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), "wg", r.get("Workgroup_Size_X"), "lds", r.get("LDS_Block_Size"),
          r["Kernel_Name"][:110])
