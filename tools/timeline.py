#!/usr/bin/env python3
"""Per-kernel timeline from a rocprofv3 kernel trace (csv): for one step of the benchmark, print
each dispatch's start, duration, grid and stream, so that gaps and overlaps can be read off.
usage: timeline.py <trace_kernel_trace.csv> [first_dispatch_name_substring] [count]"""
import csv
import sys

from kname import is_orbx, kernel_name  # noqa: F401

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if is_orbx(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# skip to the last quarter of the run (steady state)
first = max(0, len(rows) - count)
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = kernel_name(r["Kernel_Name"]).replace("orbx::", "")
    grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r)
    print("%9.1f us  +%8.1f us  %-18s grid %-18s queue %s" % ((s - t0) / 1e3, (e - s) / 1e3, name, grid, r.get("Queue_Id", "?")))
