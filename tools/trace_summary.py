#!/usr/bin/env python3
"""Per-dispatch durations of the orbx kernels of the LAST benchmark step in a rocprofv3 kernel trace."""
import csv
import sys

from kname import is_orbx, kernel_name  # noqa: F401

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if is_orbx(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kernel_name(r["Kernel_Name"]),
                         r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
# last step = last 26 dispatches (7 resize + 1 fast + 1 octree + 8 blur + 1 describe)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
last = rows[-n:]
t0 = last[0][0]
for s, e, name, gx, gy, gz in last:
    print("%-26s start %8.1f us  dur %8.1f us  grid %s x %s x %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, gx, gy, gz))
print("span %.1f us" % ((last[-1][1] - t0) / 1e3))
