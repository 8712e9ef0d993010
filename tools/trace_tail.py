"""Print the last N kernel launches of a rocprofv3 --kernel-trace CSV whose name contains a substring (duration, grid, name):
    python tools/trace_tail.py <dir> <substring> <N>"""
import csv
import glob
import sys

d, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_prev = None
for r in rows[-n:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if t_prev is None else f"gap {(s - t_prev) / 1e3:6.1f}"
    t_prev = e
    grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    print(f"{(e - s) / 1e3:8.1f} us  {gap:12s} grid={grid:>8s}/{wg:>4s}  {name[:90]}")
