#!/usr/bin/env python3
"""Print the tail of a rocprofv3 --kernel-trace CSV as a timeline (start us, duration us, queue, kernel, grid):
tools/trace_timeline.py <dir-or-csv> [n_last]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1000:11.1f} {(e - s) / 1000:8.1f} q{r.get('Queue_Id', '?'):>2} {r['Kernel_Name'][:44]:44} grid={r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')}")
