#!/usr/bin/env python3
"""Print the last n kernel launches of a rocprofv3 kernel trace: start offset, duration, name.
usage: show_trace_tail.py <dir with *kernel_trace.csv> [n]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  wg {r.get('Workgroup_Size', '?'):>5} grid {r.get('Grid_Size', '?'):>8}  {r['Kernel_Name'][:80]}")
