#!/usr/bin/env python3
"""Sum rocprofv3 counter rows per (kernel, counter); print per-dispatch averages."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(float)
calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].split("(")[0]
        if not name.startswith("orbx::"):
            continue
        acc[(name, row["Counter_Name"])] += float(row["Counter_Value"])
        calls[name].add(row["Dispatch_Id"])
for (name, ctr), v in sorted(acc.items()):
    n = len(calls[name])
    print("%-28s %-26s total %.4g  per-dispatch %.4g  (%d dispatches)" % (name, ctr, v, v / n, n))
